#!/bin/bash
# A tuning build of the whole library (-DPMT_TUNING: environment switches and the copy-engine trace compiled in):
#   tools/build_tuning.sh [name] ["extra flags"]  -> parametron.jl_amd/lib_variants/<name>.so   (select with PMT_LIB_PATH)
set -e
cd "$(dirname "$0")/../parametron.jl_amd/csrc"
name=${1:-tuning}; extra=$2
mkdir -p ../lib_variants ../build/tuning_$name
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function -DPMT_TUNING $extra"
objs=""
for f in *.hip; do
  o=../build/tuning_$name/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ gram_common.h -nt $o ] || [ dma.h -nt $o ] || [ common.h -nt $o ]; then
    /opt/rocm/bin/hipcc $BASE -c $f -o $o 2>/dev/null &
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib_variants/$name.so $objs -ldl
echo "built lib_variants/$name.so"
