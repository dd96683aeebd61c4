"""Per-phase cycle breakdown of the stream-K Gram kernel (workgroup 0, every wave): run with PMT_GRAM_SK_ABLATE=4."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMT_GRAM_SK_ABLATE", "4")
import torch  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402


def dptr(t):
    return C.c_void_p(t.data_ptr())


r, n = (int(v) for v in (sys.argv[1].split("x") if len(sys.argv) > 1 else ("4096", "4096")))
dev = torch.device("cuda:0")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.empty(r * n, dtype=torch.float64, device=dev)
b = torch.empty(r, dtype=torch.float64, device=dev)
_lib.call("pmt_fill_uniform_f64", dptr(A), r * n, 1, 1.0, stream)
_lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
nq = n * (n + 1) // 2
Q = torch.empty(nq * 3, dtype=torch.int64, device=dev)
q = torch.empty(n * 2, dtype=torch.int64, device=dev)
c = torch.empty(1, dtype=torch.float64, device=dev)
ws = torch.zeros(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)
for _ in range(3):
    _lib.call("pmt_quad_gram_f64", dptr(A), r, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream)
torch.cuda.synchronize()
dbg = ws[1000 * 16384:1000 * 16384 + 64].cpu().numpy().reshape(8, 8)
print("wave  load-issue  mfma-block  lds-store  barrier   stages   (cycles per stage)")
for w in range(8):
    st = max(dbg[w, 4], 1)
    print("%4d  %10.0f  %10.0f  %9.0f  %7.0f   %6.0f" % (w, dbg[w, 0] / st, dbg[w, 1] / st, dbg[w, 2] / st, dbg[w, 3] / st, st))
