#!/bin/bash
# A/B over library builds of gram_tall.hip (tools/build_variants_file.sh): tools/tall_ab.sh shapes -- variants
shapes=(); while [ "$1" != "--" ]; do shapes+=("$1"); shift; done; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "[$rep] $v"; PMT_LIB_PATH=$PWD/parametron.jl_amd/lib_variants/$v.so python tools/tall_probe.py "${shapes[@]}" 2>&1 | grep "^r="
  done
done
