"""What could an ASYNCHRONOUS tile epilogue of the contraction win?  (park the finished tile's accumulators in AGPRs and
drain them through LDS inside the next tile's stage loop.)  Before building it, bound it with the pieces that exist:
   (a) the shipped kernel                                         (PMT_LIB_PATH = shipped library)
   (b) the kernel without its epilogue's global stores           (build with -DPMT_SK_EPI_ABL=2: the staging and the word() arithmetic stay)
       = the most a perfectly hidden write-out could save
   (c) (b) + the same ~270 MB of traffic issued by OTHER waves of the same CUs while the contraction runs: N launches of the <= 16-VGPR
       co-resident pack kernel on a second stream (each reads 16.8 MB and writes 50 MB).  An async epilogue's stores come from the
       contraction's own waves and share the CU's vector-memory path with its panel loads exactly as these do.
If (c) is not clearly below (a), spreading the stores over the stage loop cannot win: they cost the panel loads what the burst costs now.
    python tools/gram_async_epilogue_bound.py [N background launches, default 4]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402,F401
from parametron_jl_amd import _lib  # noqa: E402

nbg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
r = n = 4096
m = 512
dev = torch.device("cuda:0")
dptr = lambda t: C.c_void_p(t.data_ptr())
main_s = torch.cuda.Stream()
side_s = torch.cuda.Stream(priority=0)
sm, ss = C.c_void_p(main_s.cuda_stream), C.c_void_p(side_s.cuda_stream)
A = torch.empty(r * n, dtype=torch.float64, device=dev)
b = torch.empty(r, dtype=torch.float64, device=dev)
Cm = torch.empty(m * n, dtype=torch.float64, device=dev)
_lib.call("pmt_fill_uniform_f64", dptr(A), r * n, 1, 1.0, None)
_lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, None)
_lib.call("pmt_fill_uniform_f64", dptr(Cm), m * n, 3, 1.0, None)
xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
nq = n * (n + 1) // 2
Q = torch.empty(nq * 3, dtype=torch.int64, device=dev)
q = torch.empty(n * 2, dtype=torch.int64, device=dev)
c = torch.empty(1, dtype=torch.float64, device=dev)
ws = torch.empty(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)
outs = [torch.empty(m * n * 3, dtype=torch.int64, device=dev) for _ in range(max(nbg, 1))]
cc = torch.empty(m, dtype=torch.float64, device=dev)
torch.cuda.synchronize()


def step(background):
    _lib.call("pmt_quad_gram_f64", dptr(A), r, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), sm)
    for k in range(background):
        _lib.call("pmt_affine_pack_vector_background_f64", dptr(Cm), m, m, n, dptr(xvar), None, 0, dptr(xvar), 0, dptr(outs[k]), dptr(cc), ss)


for bg in (0, nbg, 0, nbg):
    for _ in range(25):
        step(bg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step(bg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("background launches %d (%.0f MB of co-resident traffic per contraction): %.4f ms per contraction" % (bg, bg * 67.1, dt * 1e3), flush=True)
