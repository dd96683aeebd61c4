#!/bin/bash
# like build_variants.sh but for an arbitrary source file: tools/build_variants_file.sh batch_small name1 "flags" ...
set -e
cd "$(dirname "$0")/../parametron.jl_amd/csrc"
SRC=$1; shift
make -s -j8 >/dev/null
mkdir -p ../lib_variants ../build/variants
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function"
OTHERS=$(ls ../build/*.o | grep -v "/$SRC.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $BASE $flags -c $SRC.hip -o ../build/variants/${SRC}_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib_variants/$name.so $OTHERS ../build/variants/${SRC}_$name.o
  echo "built $name ($flags)"
done
