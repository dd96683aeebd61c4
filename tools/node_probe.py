import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import parametron_jl_amd as P
from parametron_jl_amd import _lib
def dptr(t): return C.c_void_p(t.data_ptr())
dev = torch.device("cuda:0")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for r, n in [(4096, 4096), (16384, 4096), (65536, 1024), (262144, 512), (1048576, 128)]:
    A = torch.empty(r * n, dtype=torch.float64, device=dev); b = torch.empty(r, dtype=torch.float64, device=dev)
    _lib.call("pmt_fill_uniform_f64", dptr(A), r * n, 1, 1.0, stream); _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    nq = n * (n + 1) // 2
    Q = torch.empty(nq * 3, dtype=torch.int64, device=dev); q = torch.empty(n * 2, dtype=torch.int64, device=dev); c = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)
    def run(): _lib.call("pmt_quad_gram_f64", dptr(A), r, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream)
    for _ in range(20): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); node = (time.perf_counter() - t0) / 20
    P.profile_enable(True)
    for _ in range(5): run()
    torch.cuda.synchronize(); rep = P.profile_report(); P.profile_enable(False)
    print("r=%d n=%d node %.3f ms | %s" % (r, n, node * 1e3, {k.replace("_kernel", ""): round(v["avg_ms"], 3) for k, v in rep.items()}), flush=True)
