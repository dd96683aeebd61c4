#!/bin/bash
# A/B of the pair-fold handshake (gram_sk.hip), alternating, host_csc per solve:
#   pair_relaxed   round 4: relaxed agent-scope atomic stores / loads of the partial and of the flag + s_waitcnt
#   pair_formal    plain stores -> barrier -> RELEASE flag store; relaxed spin -> ACQUIRE fence -> barrier -> plain loads
#   pair_rel_only  release on the producer, no acquire fence: the consumer reads the partial with agent-scope atomic loads
#   pair_acq_only  relaxed producer, acquire fence on the consumer
for rep in 1 2 3; do
  for v in "$@"; do
    echo "[$rep] $v: $(PMT_LIB_PATH=$PWD/parametron.jl_amd/lib_variants/$v.so python tools/host_api_bench.py 30 2>/dev/null | grep -E '"handoff_host_csc"|"c3_host_csc"|"handoff_device"' | tr -d '\n')"
  done
done
