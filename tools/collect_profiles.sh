#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/<tag>/ (copy the summaries into profiles/ afterwards):
#   bench JSON (default run), rocprofv3 --kernel-trace --stats of the same command, PMC FETCH_SIZE / WRITE_SIZE in separate passes.
# usage: gpurun -- 'bash tools/collect_profiles.sh r01c'
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu-baseline --no-rocprof-child > "$OUT/bench_under_rocprof.json" 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof-child > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof-child > /dev/null 2>&1
timeout 300 python bench.py --workload batch > "$OUT/bench_batch.json" 2> /dev/null
python - "$OUT" <<'PY'
import csv, collections, glob, json, sys
out = sys.argv[1]
f = collections.defaultdict(list); w = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(out + "/pmc_fetch/*/*counter_collection.csv")[0])):
    f[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(glob.glob(out + "/pmc_write/*/*counter_collection.csv")[0])):
    w[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
res = {"_note": "per-launch HBM-side bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950 correction) x1024, "
                "WRITE_SIZE x1024; `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`; per kernel the launches of its largest workload; tag " + out}
# a kernel is launched at several sizes in one bench process (the configurations, the small model of config 1, the shapes of `tall`): the
# record is the average over the launches of the LARGEST workload (per-launch read + write within a factor two of the maximum; both
# passes run the same launches in the same order)
for k in f:
    if "pmt::" in k and k in w:
        rd, wr = [2 * 1024 * v for v in f[k]], [1024 * v for v in w[k]]
        if len(rd) == len(wr):
            tot = [a + b for a, b in zip(rd, wr)]
            keep = [i for i, t in enumerate(tot) if t >= 0.5 * max(tot)] or list(range(len(tot)))
        else:
            keep = None
        sel = (lambda x: [x[i] for i in keep]) if keep is not None else (lambda x: x)
        res[k] = {"read_bytes": sum(sel(rd)) / len(sel(rd)), "write_bytes": sum(sel(wr)) / len(sel(wr)), "launches": len(sel(rd)), "launches_in_process": len(rd)}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
# per-launch durations of the timed region only (the stats CSV averages over spin-up and warm-up launches too)
tr = sorted(csv.DictReader(open(glob.glob(out + "/stats/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
bu = json.load(open(out + "/bench_under_rocprof.json"))
# launches of a step's kernel in start order: setup spin-up, W warm-up steps, then the K timed steps (everything later in the process —
# the separate all-kernel pass, the refresh variant, the other configurations — comes after them)
first = bu["config"]["setup_spinup_steps"] + bu["warmup"]
lines = ["rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline`, per-launch durations of the TIMED REGION (launches %d..%d of each kernel):"
         % (first, first + bu["steps"] - 1)]
for name in ("gram_sk_kernel", "gram_sk_fixup_kernel", "affine_tile_kernel<1"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if name in r["Kernel_Name"]]
    reg = d[first:first + bu["steps"]]
    lines.append("%-28s %3d launches in the process; timed region: avg %.1f us, min %.1f, max %.1f; spin-up + warm-up before it: avg %.1f us" %
                 (name, len(d), sum(reg) / len(reg), min(reg), max(reg), sum(d[:first]) / max(1, first)))
json.dump({name: {"avg_us": sum(d[first:first + bu["steps"]]) / bu["steps"], "launches": bu["steps"],
                  "source": "rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline`, launches of the timed region; tag " + out}
           for name, key in (("gram_sk_kernel", "gram_sk_kernel<"), ("gram_sk_fixup_kernel", "gram_sk_fixup_kernel"), ("affine_tile_kernel<VAT>", "affine_tile_kernel<1"))
           for d in [[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if key in r["Kernel_Name"]]] if len(d) >= first + bu["steps"]},
          open(out + "/rocprof_in_step.json", "w"), indent=1)
lines.append("bench.py's HIP-event average for the dominant kernel in the same run: %.1f us" % (bu["roofline"]["avg_ms"] * 1e3))
open(out + "/rocprofv3_timed_region.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
b = json.load(open(out + "/bench.json"))
print("bench:", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline_affine"]["achieved"], b["roofline_affine"]["frac"])
print(open(glob.glob(out + "/stats/*/*kernel_stats.csv")[0]).read()[:1200])
for k, v in res.items():
    if k != "_note": print(k, round(v["read_bytes"] / 1e6, 1), round(v["write_bytes"] / 1e6, 1))
PY
