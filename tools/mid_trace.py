#!/usr/bin/env python
"""Phase trace of the one-launch mid-size Gram node (gram_mid.hip built with -DPMT_MID_TRACE; PMT_LIB_PATH points at that build):
100 MHz wall-clock stamps per workgroup — 0 start, 1 main loop done, 2 waves summed, 3 partial out + counted, 4 partials folded,
5 tile in LDS, 6 terms stored.  usage: PMT_LIB_PATH=.../trace.so python tools/mid_trace.py 4096 512"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parametron_jl_amd  # noqa: F401,E402
from parametron_jl_amd import _lib  # noqa: E402

r, n = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.rand(r * n, dtype=torch.float64, device=dev)
b = torch.rand(r, dtype=torch.float64, device=dev)
x = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
nq = n * (n + 1) // 2
oq = torch.empty(nq * 3, dtype=torch.int64, device=dev)
ol = torch.empty(n * 2, dtype=torch.int64, device=dev)
oc = torch.empty(1, dtype=torch.float64, device=dev)
ws = torch.empty(max(1, lib.pmt_quad_gram_workspace_bytes(r, n) // 8), dtype=torch.float64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(20):
    _lib.call("pmt_quad_gram_f64", p(A), r, r, n, p(x), p(b), -1, 1, None, p(oq), p(ol), p(oc), p(ws), stream)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 8))()
fn = C.CDLL(os.environ["PMT_LIB_PATH"]).pmt_mid_trace_read
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
live = t[:, 0] > 0
t = t[live]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
us[t == 0] = np.nan
print("workgroups with a stamp:", len(t))
names = ["start", "main loop done", "waves summed", "partial out + counted", "folded (last arrivers)", "tile in LDS", "terms stored"]
for k, nm in enumerate(names):
    col = us[:, k]
    col = col[~np.isnan(col)]
    if len(col):
        print("%-26s n=%4d  min %6.2f  median %6.2f  max %6.2f us" % (nm, len(col), col.min(), np.median(col), col.max()))
d = us[:, 1] - us[:, 0]
print("main loop duration: median %.2f max %.2f" % (np.nanmedian(d), np.nanmax(d)))
la = ~np.isnan(us[:, 4])
for a, bq, nm in ((2, 3, "partial out + count"), (3, 4, "fold"), (4, 5, "tile -> LDS"), (5, 6, "epilogue")):
    dd = us[la, bq] - us[la, a]
    if len(dd):
        print("last arrivers: %-20s median %.2f max %.2f us" % (nm, np.nanmedian(dd), np.nanmax(dd)))
# dispatch gap: the i-th workgroup to START beyond the first 256 takes the CU the i-th workgroup to END has left
st, en = np.sort(us[:, 0]), np.sort(np.nanmax(us, axis=1))
if len(st) > 256:
    gap = st[256:] - en[:len(st) - 256]
    print("workgroup end -> next workgroup's start on the freed CU: median %.2f us, min %.2f, max %.2f (n = %d)" % (np.median(gap), gap.min(), gap.max(), len(gap)))
