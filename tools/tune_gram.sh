#!/bin/bash
# Builds the codegen-knob variants of gram_sk.hip (stagger x load-issue k-step x MFMA order), reports VGPRs / scratch of the shipped
# instantiation, and leaves parametron.jl_amd/lib_variants/t_<s>_<l>_<o>.so for tools/lib_ab.sh on the GPU box.
cd "$(dirname "$0")/.."
for s in 0 1; do for l in 0 1 2; do for o in 0 1; do
  name=t_${s}_${l}_${o}
  flags="-DPMT_GRAM_SK_STAGGER=$s -DPMT_SK_LOADKS=$l -DPMT_SK_ORDER=$o"
  tools/build_variants.sh $name "$flags" > /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Iparametron.jl_amd/csrc $flags -S --cuda-device-only -o /tmp/tune.s parametron.jl_amd/csrc/gram_sk.hip 2>/dev/null
  echo "$name $(awk '/^_ZN3pmt14gram_sk_kernelILi2ELi16ELi2ELi0EEEvNS_6SKArgsE:/{f=1} f&&/; NumVgprs|ScratchSize/{printf "%s ", $0} /; Occupancy/{if(f){exit}}' /tmp/tune.s)"
done; done; done
