#!/bin/bash
# Per-kernel PMC summary (one rocprofv3 pass per counter, --kernel-trace only):
#   tools/pmc_kernel.sh <outdir> <kernel-substring> <cmd...>
set -u
OUT=$1; KERN=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in ${PMC_LIST:-GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR FETCH_SIZE WRITE_SIZE}; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- "$@" > /dev/null 2>&1
done
python - "$OUT" "$KERN" <<'PY'
import csv, glob, sys
out, kern = sys.argv[1], sys.argv[2]
vals, dur = {}, []
for d in sorted(glob.glob(out + "/*/")):
    name = d.rstrip("/").split("/")[-1]
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
    if v: vals[name] = sum(v) / len(v)
    t = glob.glob(d + "*/*kernel_trace.csv")
    if t:
        dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]]
lines = ["rocprofv3 --kernel-trace --pmc <counter> (one pass per counter); per-launch averages for kernels matching '%s'" % kern]
for k, v in vals.items(): lines.append("%-28s %.1f" % (k, v))
ns = sum(dur) / max(len(dur), 1)
lines.append("kernel duration under the profiler  %.1f us (%d launches)" % (ns / 1e3, len(dur)))
if "GRBM_GUI_ACTIVE" in vals and ns:
    clk = vals["GRBM_GUI_ACTIVE"] / 8 / ns
    lines.append("shader clock %.3f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % clk)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        lines.append("MFMA pipe busy %.0f cycles per SIMD = %.1f %% of %.0f kernel cycles" % (busy, 100 * busy / (clk * ns), clk * ns))
    if "SQ_LDS_IDX_ACTIVE" in vals:
        lines.append("LDS busy %.0f cycles per CU = %.1f %% (bank conflicts %.1f %% of them)" % (vals["SQ_LDS_IDX_ACTIVE"] / 256, 100 * vals["SQ_LDS_IDX_ACTIVE"] / 256 / (clk * ns), 100 * vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(vals["SQ_LDS_IDX_ACTIVE"], 1)))
if "SQ_WAVE_CYCLES" in vals:
    w = vals["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
        if k in vals: lines.append("%-22s %.1f %% of wave cycles" % (k, 100 * vals[k] / w))
if "FETCH_SIZE" in vals: lines.append("HBM read  %.1f MB per launch (FETCH_SIZE x 2 gfx950 correction x 1024)" % (vals["FETCH_SIZE"] * 2 * 1024 / 1e6))
if "WRITE_SIZE" in vals: lines.append("HBM write %.1f MB per launch (WRITE_SIZE x 1024)" % (vals["WRITE_SIZE"] * 1024 / 1e6))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
