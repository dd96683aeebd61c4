"""Tall Gram node (gram_tall.hip) on the GPU: time per shape, per-kernel split, and a check of Q, q, const against float64 torch
(a development probe — the parity tests proper compare with the oracle: tests/test_gpu_fullsize_parity.py)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parametron_jl_amd as P
from parametron_jl_amd import _lib
def dptr(t): return C.c_void_p(t.data_ptr())
dev = torch.device("cuda:0")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(1 << 20, 128), (262144, 128), (8192, 128), (1024, 128), (100003, 100), (5000, 17), (65536, 64), (4096, 4096), (262144, 512), (65536, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for r, n in shapes:
    lda = r + (64 if (r * 8) % 4096 == 0 and os.environ.get("PAD", "1") == "1" else 0)
    A = torch.empty(lda * n, dtype=torch.float64, device=dev); b = torch.empty(r, dtype=torch.float64, device=dev)
    _lib.call("pmt_fill_uniform_matrix_f64", dptr(A), r, n, lda, 1, 1.0, stream); _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    nq = n * (n + 1) // 2
    Q = torch.zeros(nq * 3, dtype=torch.int64, device=dev); q = torch.zeros(n * 2, dtype=torch.int64, device=dev); c = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)
    nob = os.environ.get("NOB") == "1"          # no b: the node without its affine part and constant (what the contraction alone costs)
    def run(): _lib.call("pmt_quad_gram_f64", dptr(A), lda, r, n, dptr(xvar), None if nob else dptr(b), 0 if nob else -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream)
    run(); torch.cuda.synchronize()
    ok = "unchecked"
    if r * n <= (1 << 27) + 1 and not nob:
        Am = A.view(n, lda)[:, :r]                       # column-major r x n == row-major n x lda
        G = 2.0 * (Am @ Am.T)
        iu = torch.triu_indices(n, n, device=dev)
        want = G[iu[0], iu[1]]
        got = Q.view(torch.float64)[0::3]
        eq = (got - want).abs().max().item() / want.abs().max().item()
        rows_ok = bool(torch.equal(Q[1::3], iu[0] + 1) and torch.equal(Q[2::3], iu[1] + 1))
        wq = -2.0 * (Am @ b)
        el = ((q.view(torch.float64)[0::2] - wq).abs().max() / wq.abs().max()).item()
        ec = abs(c.item() - (b * b).sum().item()) / (b * b).sum().item()
        ok = "Q relerr %.1e idx %s q relerr %.1e const relerr %.1e" % (eq, rows_ok, el, ec)
    t0 = time.perf_counter(); k = 0
    while k < 10 or time.perf_counter() - t0 < 0.04:            # (at least 40 ms: the clock settles)
        run(); k += 1
        if k % 10 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); node = (time.perf_counter() - t0) / 20
    P.profile_enable(True)
    for _ in range(5): run()
    torch.cuda.synchronize(); rep = P.profile_report(); P.profile_enable(False)
    fl = r * n * (n + 1.0)
    print("r=%d n=%d lda=%d node %.4f ms  %.1f TFLOP/s (%.2f of 78.6)  A %.2f TB/s | %s | %s" % (
        r, n, lda, node * 1e3, fl / node / 1e12, fl / node / 78.6e12, 8.0 * r * n / node / 1e12,
        {k.replace("_kernel", ""): round(v["avg_ms"], 4) for k, v in rep.items()}, ok), flush=True)
    del A, b, Q, q, ws
