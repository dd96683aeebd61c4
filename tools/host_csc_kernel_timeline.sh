# rocprofv3 --kernel-trace of tools/model_e2e.py --host-csc: the kernels of the last solve with start times relative to the first Parameter callback
# usage (GPU box): bash tools/host_csc_kernel_timeline.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_hc_trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_hc_trace -- python tools/model_e2e.py --host-csc > /dev/null 2>&1
python - <<'PY'
import csv, glob
tr = sorted(csv.DictReader(open(glob.glob("gpurun_out/r04_hc_trace/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
# last complete solve: find the last 5 ranged gram launches
idx = [i for i, r in enumerate(tr) if "gram_sk_kernel<2, 16, 2, 0, true>" in r["Kernel_Name"]]
last = idx[-5:]
i0 = last[0]
# go back to the fill of A before it
j = i0
while j > 0 and "fill_uniform_matrix" not in tr[j]["Kernel_Name"]: j -= 1
while j > 0 and "fill_uniform" in tr[j - 1]["Kernel_Name"]: j -= 1
t0 = int(tr[j]["Start_Timestamp"])
for r in tr[j:last[-1] + 6]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pmt::", "")[:48]
    print("%-50s start +%8.1f us  dur %7.1f us  stream/queue %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "")))
PY
