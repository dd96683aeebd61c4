#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/<tag>/ (copy the summaries into profiles/ afterwards):
#   the bench line + bench_detail.json (default run and the driver's command), rocprofv3 --kernel-trace --stats of the same command,
#   per-launch durations of the timed region, and PMC FETCH_SIZE / WRITE_SIZE (separate passes) keyed by (kernel, SHAPE): one
#   tools/shape_once.py process per shape, so that a kernel launched at several sizes is never averaged over them.
# usage: gpurun -- 'bash tools/collect_profiles.sh r06'
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --detail "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_driver_detail.json" > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu-baseline --no-pmc --detail "$OUT/bench_under_rocprof_detail.json" > "$OUT/bench_under_rocprof.json" 2> /dev/null
timeout 300 python bench.py --workload batch > "$OUT/bench_batch.json" 2> /dev/null
# PMC traffic per (kernel, shape): the step's kernels from a short timed loop of bench.py; every other kernel from its own shape process
pmc() {  # pmc <name> <cmd...>
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc/$name/$c" -- "$@" > /dev/null 2>&1
  done
}
pmc c2_step python bench.py --steps 5 --warmup 1 --timed-loop-only
for shape in 1048576x16 1048576x32 1048576x64 1048576x128 262144x512 65536x1024 4096x512 300x300 8192x512 4096x1024 65536x512; do pmc gram_$shape python tools/shape_once.py $shape 5; done
pmc c4_batch python bench.py --workload batch --steps 5 --warmup 2
pmc c5_sparse python tools/c5_once.py 5
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
res = {"_note": "per-launch HBM-side bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950 correction) x1024, "
                "WRITE_SIZE x1024; keyed by workload (one process per shape: tools/collect_profiles.sh), then by kernel; averages over the launches "
                "of that process whose bytes are within a factor two of its largest launch (warm-up fills and setup kernels aside); tag " + out}
for d in sorted(glob.glob(out + "/pmc/*/")):
    name = d.rstrip("/").split("/")[-1]
    per = {}
    for counter, scale, key in (("FETCH_SIZE", 2 * 1024.0, "read_bytes"), ("WRITE_SIZE", 1024.0, "write_bytes")):
        f = glob.glob(d + counter + "/*/*counter_collection.csv")
        if not f:
            continue
        rows = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "pmt::" in k:
                rows[k].append(scale * float(r["Counter_Value"]))
        for k, v in rows.items():
            keep = [x for x in v if x >= 0.5 * max(v)] or v
            per.setdefault(k, {})[key] = sum(keep) / len(keep)
            per[k]["launches"] = len(keep)
    res[name] = per
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
# the replay file bench.py falls back to when rocprofv3 is absent: the step's kernels, flat
flat = {k: v for k, v in res.get("c2_step", {}).items() if "read_bytes" in v and "write_bytes" in v}
flat["_note"] = res["_note"]
json.dump(flat, open(out + "/pmc_traffic_c2_step.json", "w"), indent=1)
# per-launch durations of the timed region only (the stats CSV averages over spin-up and warm-up launches too)
tr = sorted(csv.DictReader(open(glob.glob(out + "/stats/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
bu = json.load(open(out + "/bench_under_rocprof.json"))
det = json.load(open(out + "/bench_under_rocprof_detail.json"))
first = det["config_detail"]["setup_spinup_steps"] + bu["warmup"]
lines = ["rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-pmc`, per-launch durations of the TIMED REGION (launches %d..%d of each kernel):"
         % (first, first + bu["steps"] - 1)]
for name in ("gram_mid_kernel", "gram_sk_kernel", "gram_sk_fixup_kernel", "affine_tile_kernel<1"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if name in r["Kernel_Name"]]
    reg = d[first:first + bu["steps"]]
    if reg:
        lines.append("%-28s %3d launches in the process; timed region: avg %.1f us, min %.1f, max %.1f; spin-up + warm-up before it: avg %.1f us" %
                     (name, len(d), sum(reg) / len(reg), min(reg), max(reg), sum(d[:first]) / max(1, first)))
lines.append("bench.py's HIP-event average for the dominant kernel in the same run: %.1f us" % (bu["roofline"]["avg_ms"] * 1e3))
open(out + "/rocprofv3_timed_region.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for nm in ("bench.json", "bench_driver.json"):
    b = json.load(open(out + "/" + nm))
    print(nm, len(open(out + "/" + nm).read()), "bytes:", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"].get("traffic"), b["roofline"].get("measured_in_this_run"))
print(open(glob.glob(out + "/stats/*/*kernel_stats.csv")[0]).read()[:1500])
for name, per in res.items():
    if name == "_note": continue
    for k, v in per.items():
        if "read_bytes" in v: print("%-18s %-60s read %8.1f MB  write %8.1f MB  (%d launches)" % (name, k[:60], v["read_bytes"] / 1e6, v.get("write_bytes", 0) / 1e6, v["launches"]))
PY
find "$OUT" -name "*.db" -delete; find "$OUT/pmc" -name "*.csv" -size +200k -delete
