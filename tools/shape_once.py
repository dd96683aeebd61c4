"""one shape of the objective node, a few launches (the workload of a PMC pass): python tools/shape_once.py ROWSxCOLS [launches]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parametron_jl_amd import _lib
from parametron_jl_amd.device import padded_lda
r, n = (int(v) for v in sys.argv[1].split("x"))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0"); s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
lda = padded_lda(r)
A = torch.empty(lda * n, dtype=torch.float64, device=dev); b = torch.empty(r, dtype=torch.float64, device=dev)
_lib.call("pmt_fill_uniform_matrix_f64", P(A), r, n, lda, 1, 1.0, s); _lib.call("pmt_fill_uniform_f64", P(b), r, 2, 1.0, s)
x = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
Q = torch.empty(n * (n + 1) // 2 * 3, dtype=torch.int64, device=dev); q = torch.empty(2 * n, dtype=torch.int64, device=dev); c = torch.empty(1, dtype=torch.float64, device=dev)
ws = torch.empty(max(1, _lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8), dtype=torch.float64, device=dev)
for _ in range(k):
    _lib.call("pmt_quad_gram_f64", P(A), lda, r, n, P(x), P(b), -1, 1, P(x), P(Q), P(q), P(c), P(ws), s)
torch.cuda.synchronize()
