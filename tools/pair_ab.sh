#!/bin/bash
# A/B of the pair-fold handshake (gram_sk.hip): round 4's relaxed form against the release/acquire form, alternating, host_csc per solve
for rep in 1 2 3; do
  for v in pair_relaxed pair_formal; do
    echo "[$rep] $v: $(PMT_LIB_PATH=$PWD/parametron.jl_amd/lib_variants/$v.so python tools/host_api_bench.py 30 2>/dev/null | grep -E '"handoff_host_csc"|"c3_host_csc"|"handoff_device"' | tr -d '\n')"
  done
done
