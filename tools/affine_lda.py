"""Effect of a padded leading dimension on the affine kernels (column stride 32 KiB vs 32.5 KiB)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402


def dptr(t):
    return C.c_void_p(t.data_ptr())


dev = torch.device("cuda:0")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (r, n) in ((4096, 4096), (512, 4096)):
    for pad in (0, 16, 64):
        lda = r + pad
        A = torch.empty(lda * n, dtype=torch.float64, device=dev)
        b = torch.empty(r, dtype=torch.float64, device=dev)
        _lib.call("pmt_fill_uniform_f64", dptr(A), lda * n, 1, 1.0, stream)
        _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
        xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
        lt = torch.empty(r * n * 2, dtype=torch.int64, device=dev)
        vat = torch.empty(r * n * 3, dtype=torch.int64, device=dev)
        cc = torch.empty(r, dtype=torch.float64, device=dev)

        def run():
            _lib.call("pmt_affine_assemble_f64", dptr(A), lda, r, n, dptr(xvar), dptr(b), -1, dptr(lt), dptr(cc), stream)
            _lib.call("pmt_affine_pack_vector_f64", dptr(A), lda, r, n, dptr(xvar), dptr(b), -1, dptr(xvar), 0, dptr(vat), dptr(cc), stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        P.profile_enable(True)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        rep = P.profile_report()
        P.profile_enable(False)
        a, v = rep["affine_tile_kernel<LT>"]["avg_ms"], rep["affine_tile_kernel<VAT>"]["avg_ms"]
        print("%dx%d lda=rows+%2d : LT %.4f ms (%.0f GB/s)   VAT %.4f ms (%.0f GB/s)" % (r, n, pad, a, 24.0 * r * n / a / 1e6, v, 32.0 * r * n / v / 1e6), flush=True)
