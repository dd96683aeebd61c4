#!/bin/bash
# A/B of library variants (parametron.jl_amd/lib_variants/<name>.so, tools/build_variants_file.sh) on the objective node's shapes, order alternated:
#   tools/stream_ab.sh "1048576x16,1048576x64" base abl1 abl2 ...      (base = the shipped library)
SHAPES=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/parametron.jl_amd/lib_variants/$v.so; [ "$v" = base ] && lib=$PWD/parametron.jl_amd/lib/libparametron_hip.so
    echo "[$rep] $v"; PMT_LIB_PATH=$lib python tools/bench_study.py tall --shapes $SHAPES --out /tmp/ab.json 2>/dev/null | grep " us "
  done
done
