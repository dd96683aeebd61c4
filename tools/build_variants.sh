#!/bin/bash
# Builds A/B variants of libparametron_hip.so that differ only in how gram_sk.hip is compiled:
#   tools/build_variants.sh name1 "extra flags" name2 "extra flags" ...
# -> parametron.jl_amd/lib_variants/<name>.so   (select with PMT_LIB_PATH)
set -e
cd "$(dirname "$0")/../parametron.jl_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../lib_variants ../build/variants
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function"
OTHERS=$(ls ../build/*.o | grep -v gram_sk.o)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $BASE $flags -c gram_sk.hip -o ../build/variants/gram_sk_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib_variants/$name.so $OTHERS ../build/variants/gram_sk_$name.o
  echo "built $name ($flags)"
done
