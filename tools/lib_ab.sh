#!/bin/bash
# A/B over library builds (tools/build_variants.sh): tools/lib_ab.sh base maxilp ...
for rep in 1 2 3; do
  for v in "$@"; do
    echo "[$rep] $v: $(PMT_LIB_PATH=$PWD/parametron.jl_amd/lib_variants/$v.so python tools/gram_probe.py 4096x4096 2>&1 | grep rows | tr '\n' ' ')"
  done
done
