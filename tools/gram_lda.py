"""Does the power-of-two column stride (lda = 4096 doubles = 32 KiB) hot-spot memory channels?  Time pmt_quad_gram_f64 with padded lda."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402


def dptr(t):
    return C.c_void_p(t.data_ptr())


r, n = 4096, 4096
dev = torch.device("cuda:0")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for pad in (64, 0, 512, 0, 64, 0, 32):
    lda = r + pad
    A = torch.empty(lda * n, dtype=torch.float64, device=dev)
    b = torch.empty(r, dtype=torch.float64, device=dev)
    _lib.call("pmt_fill_uniform_f64", dptr(A), lda * n, 1, 1.0, stream)
    _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    nq = n * (n + 1) // 2
    Q = torch.empty(nq * 3, dtype=torch.int64, device=dev)
    q = torch.empty(n * 2, dtype=torch.int64, device=dev)
    c = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)

    def run():
        _lib.call("pmt_quad_gram_f64", dptr(A), lda, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    P.profile_enable(True)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    rep = P.profile_report()
    P.profile_enable(False)
    ms = rep["gram_sk_kernel"]["avg_ms"]
    print("lda = rows + %3d : gram %.4f ms  (%.1f TFLOP/s algorithmic)" % (pad, ms, float(r) * n * (n + 1) / ms / 1e9), flush=True)
