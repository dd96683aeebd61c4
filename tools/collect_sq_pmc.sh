#!/bin/bash
# MFMA-utilisation counters of the Gram kernel (separate rocprofv3 --pmc passes, --kernel-trace only):
#   gpurun -- 'bash tools/collect_sq_pmc.sh r01e'   ->  gpurun_out/<tag>_gram_sk_pmc_sq.txt
set -u
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY; do
  GRAM_PROBE_WARM=5 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python tools/gram_probe.py 4096x4096 > /dev/null 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
vals, dur = {}, []
for d in sorted(glob.glob(out + "/*/")):
    name = d.rstrip("/").split("/")[-1]
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gram_sk_kernel" in r["Kernel_Name"]]
    if v: vals[name] = sum(v) / len(v)
    t = glob.glob(d + "*/*kernel_trace.csv")
    if t:
        dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "gram_sk_kernel" in r["Kernel_Name"]]
lines = ["rocprofv3 --kernel-trace --pmc <counter> (one pass per counter), python tools/gram_probe.py 4096x4096; averages per gram_sk_kernel launch; tag " + tag]
for k, v in vals.items(): lines.append("%-28s %.1f" % (k, v))
ns = sum(dur) / max(len(dur), 1)
lines.append("kernel duration under the profiler  %.1f us" % (ns / 1e3))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    clk = vals["GRBM_GUI_ACTIVE"] / 8 / ns                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024               # per SIMD (256 CUs x 4)
    lines.append("shader clock %.3f GHz; MFMA pipe busy %.0f cycles per SIMD = %.1f %% of the %.0f cycles the kernel ran"
                 % (clk, busy, 100 * busy / (clk * ns), clk * ns))
open("gpurun_out/%s_gram_sk_pmc_sq.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
