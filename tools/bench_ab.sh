#!/bin/bash
# A/B of library variants (parametron.jl_amd/lib_variants/<name>.so) on the HEADLINE step (bench.py, config 2, 200 steps), alternated three times:
#   tools/bench_ab.sh base nolin      -> value, ms_per_step, gram_sk avg_ms, roofline.frac per run
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$PWD/parametron.jl_amd/lib_variants/$v.so; [ "$v" = base ] && lib=$PWD/parametron.jl_amd/lib/libparametron_hip.so
    echo "[$rep] $v"; PMT_LIB_PATH=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(d['value'], d['ms_per_step'], r['avg_ms'], r['frac'])"
  done
done
