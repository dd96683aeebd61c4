"""kernels + device->host copies of the second-to-last re-evaluation, merged by time: python tools/timeline2.py <rocprofv3 output dir>"""
import csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void pmt::", "")[:40]) for r in kt]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY d2h") for r in mc if "DEVICE_TO_HOST" in r["Direction"]]
ev.sort()
grams = [i for i, e in enumerate(ev) if "gram_sk_kernel" in e[2]]
g0 = grams[-3]; t0 = ev[g0][0]
lo = g0
while lo > 0 and ev[lo - 1][0] > t0 - 700000: lo -= 1
for s, e, n in ev[lo:grams[-2] + 1]:
    if "fill_uniform" in n: continue
    print("%9.1f %9.1f %7.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
