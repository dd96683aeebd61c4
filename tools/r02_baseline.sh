#!/bin/bash
# Round-2 baseline on the GPU box: sanity tests, fp64 co-issue microbenchmark, bench lines and the profiles round 1 did not
# commit (rocprofv3 stats + PMC passes of batch_small_kernel and sparse_slab_kernel).
#   gpurun --timeout 1500 -- 'bash tools/r02_baseline.sh r02a'
set -u
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; tail -3 "$OUT/pytest_gpu.txt"
./tools/f64_coissue > "$OUT/f64_coissue.txt" 2>&1; cat "$OUT/f64_coissue.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --workload batch > "$OUT/bench_batch.json" 2> "$OUT/bench_batch.err"
prof() {  # prof <name> <cmd...>: stats + FETCH/WRITE + SQ passes
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_stats" -- "$@" > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/${name}_pmc/$c" -- "$@" > /dev/null 2>&1
  done
}
prof batch python bench.py --workload batch --steps 5 --warmup 2
prof sparse python tools/sparse_probe.py
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("batch", "sparse"):
    lines = []
    st = glob.glob("%s/%s_stats/*/*kernel_stats.csv" % (out, name))
    if st:
        lines.append(open(st[0]).read()[:2500])
    vals = collections.defaultdict(dict)
    for d in sorted(glob.glob("%s/%s_pmc/*/" % (out, name))):
        c = d.rstrip("/").split("/")[-1]
        f = glob.glob(d + "*/*counter_collection.csv")
        if not f: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "pmt::" in k: vals[k][c] = sum(v) / len(v)
    for k, cs in vals.items():
        lines.append(k)
        for c, v in cs.items(): lines.append("   %-32s %.1f" % (c, v))
    open("%s/%s_profile_summary.txt" % (out, name), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
PY
# keep only the small artefacts (traces are large)
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
cat "$OUT/bench.json" | head -c 1500; echo
cat "$OUT/bench_batch.json" | head -c 1500; echo
