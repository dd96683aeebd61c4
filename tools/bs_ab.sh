#!/bin/bash
# A/B over library builds of batch_small.hip (tools/build_variants_file.sh batch_small <name> "<flags>" ...):
#   tools/bs_ab.sh base p34 ...   -> compute-only ms per 8192-instance step, order alternated over 3 repetitions
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "shipped" ]; then lib=$PWD/parametron.jl_amd/lib/libparametron_hip.so; else lib=$PWD/parametron.jl_amd/lib_variants/$v.so; fi
    echo "[$rep] $v: $(PMT_LIB_PATH=$lib python bench.py --workload batch --steps 30 --warmup 5 2>&1 | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("compute %.4f ms  step %.4f ms" % (d["compute_only"]["ms_per_step"], d["ms_per_step"]))')"
  done
done
