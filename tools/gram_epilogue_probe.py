"""Fixed cost of the Gram node (prologue + epilogue + launch) = its time at a contraction length of a few rows, with the MOI
term output (24 B per entry) and with the CSC value output (8 B per entry): python tools/gram_epilogue_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402


def dptr(t):
    return C.c_void_p(t.data_ptr())


def main():
    dev = torch.device("cuda:0")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 4096
    nq = n * (n + 1) // 2
    for r in (16, 64, 256, 1024, 4096):
        A = torch.empty(r * n, dtype=torch.float64, device=dev)
        b = torch.empty(r, dtype=torch.float64, device=dev)
        _lib.call("pmt_fill_uniform_f64", dptr(A), r * n, 1, 1.0, stream)
        _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
        xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
        Q = torch.empty(nq * 3, dtype=torch.int64, device=dev)
        Px = torch.empty(nq, dtype=torch.float64, device=dev)
        q = torch.empty(n * 2, dtype=torch.int64, device=dev)
        c = torch.empty(1, dtype=torch.float64, device=dev)
        ws = torch.empty(_lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8, dtype=torch.float64, device=dev)
        runs = {
            "terms": lambda: _lib.call("pmt_quad_gram_f64", dptr(A), r, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream),
            "csc": lambda: _lib.call("pmt_quad_gram_csc_f64", dptr(A), r, r, n, dptr(xvar), dptr(b), -1, dptr(xvar), 1.0, dptr(Px), None, dptr(q), dptr(c), dptr(ws), stream),
        }
        out = []
        for name, run in runs.items():
            for _ in range(30):
                run()
            torch.cuda.synchronize()
            P.profile_enable(True)
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            rep = P.profile_report()
            P.profile_enable(False)
            out.append("%s %.4f ms (+fixup %.4f)" % (name, rep["gram_sk_kernel"]["avg_ms"], rep.get("gram_sk_fixup_kernel", {"avg_ms": 0.0})["avg_ms"]))
        print("rows=%d cols=%d  " % (r, n) + "   ".join(out), flush=True)


if __name__ == "__main__":
    main()
