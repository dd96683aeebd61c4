"""Timeline of one steady-state solve! from a rocprofv3 --kernel-trace --memory-copy-trace run (CSV): kernels and copies of the LAST
complete re-evaluation relative to the start of its contraction.  usage: python tools/timeline.py <dir with *_kernel_trace.csv, *_memory_copy_trace.csv>"""
import csv
import glob
import sys

d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = []
for r in kt:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void pmt::", "")[:60]))
for r in mc:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
grams = [i for i, e in enumerate(ev) if "gram_sk_kernel" in e[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
g0 = grams[which]
t0 = ev[g0][0]
g1 = grams[which + 1] if which + 1 < 0 or which + 1 < len(grams) else len(ev)
lo = g0
while lo > 0 and ev[lo - 1][0] > t0 - 300000: lo -= 1
for s, e, n in ev[lo:g1]:
    print("%9.1f %9.1f  %8.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
