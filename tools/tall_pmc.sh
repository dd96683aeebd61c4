#!/bin/bash
# SQ counters of gram_tall_kernel (separate rocprofv3 --pmc passes, --kernel-trace only):
#   gpurun -- 'bash tools/tall_pmc.sh TAG [shape] [lib variant]'   ->  gpurun_out/<tag>_gram_tall_pmc_sq.txt
set -u
TAG=${1:-rXX}; SHAPE=${2:-1048576x128}; VAR=${3:-}
OUT=gpurun_out/${TAG}_tallsq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$VAR" ] && export PMT_LIB_PATH=$PWD/parametron.jl_amd/lib_variants/$VAR.so
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python tools/tall_probe.py $SHAPE > /dev/null 2>&1
done
python - "$OUT" "$TAG" "$SHAPE" "$VAR" <<'PY'
import csv, glob, sys
out, tag, shape, var = sys.argv[1:5]
vals, dur = {}, []
for d in sorted(glob.glob(out + "/*/")):
    name = d.rstrip("/").split("/")[-1]
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gram_tall_kernel" in r["Kernel_Name"]]
    if v: vals[name] = sum(v) / len(v)
    t = glob.glob(d + "*/*kernel_trace.csv")
    if t:
        dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "gram_tall_kernel" in r["Kernel_Name"]]
lines = ["rocprofv3 --kernel-trace --pmc <counter> (one pass per counter), python tools/tall_probe.py %s, library %s; averages per gram_tall_kernel launch; tag %s" % (shape, var or "shipped", tag)]
for k, v in vals.items(): lines.append("%-28s %.1f" % (k, v))
ns = sum(dur) / max(len(dur), 1)
lines.append("kernel duration under the profiler  %.1f us" % (ns / 1e3))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    clk = vals["GRBM_GUI_ACTIVE"] / 8 / ns
    busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
    lines.append("shader clock %.3f GHz; MFMA pipe busy %.0f cycles per SIMD = %.1f %% of the %.0f cycles the kernel ran"
                 % (clk, busy, 100 * busy / (clk * ns), clk * ns))
open("gpurun_out/%s_gram_tall_pmc_sq.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
