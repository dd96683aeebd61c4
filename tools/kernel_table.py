"""Per-kernel achieved bandwidth of the HBM-bound entry points at representative sizes (algorithmic bytes / HIP-event time):
python tools/kernel_table.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402
from parametron_jl_amd.device import padded_lda  # noqa: E402

DEV = torch.device("cuda:0")


def d(t):
    return C.c_void_p(t.data_ptr())


def f64(n):
    return torch.empty(int(n), dtype=torch.float64, device=DEV)


def i64(n):
    return torch.empty(int(n), dtype=torch.int64, device=DEV)


def timed(label, kernel, algbytes, fn, reps=20):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    P.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    rep = P.profile_report()
    P.profile_enable(False)
    ms = sum(v["avg_ms"] for k, v in rep.items() if k.startswith(kernel))
    print("%-58s %8.1f MB  %8.4f ms  %7.0f GB/s  %5.1f %% of 8 TB/s" % (label, algbytes / 1e6, ms, algbytes / ms / 1e6, algbytes / ms / 1e6 / 80), flush=True)


def main():
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = n = 4096
    m = 512
    lda, ldc = padded_lda(r), padded_lda(m)
    A, Cm, b, dd = f64(lda * n), f64(ldc * n), f64(r), f64(m)
    _lib.call("pmt_fill_uniform_matrix_f64", d(A), r, n, lda, C.c_uint64(1), 1.0, s)
    _lib.call("pmt_fill_uniform_matrix_f64", d(Cm), m, n, ldc, C.c_uint64(3), 1.0, s)
    _lib.call("pmt_fill_uniform_f64", d(b), r, C.c_uint64(2), 1.0, s)
    _lib.call("pmt_fill_uniform_f64", d(dd), m, C.c_uint64(4), 2.0, s)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=DEV)
    lt, consts = i64(2 * r * n), f64(r)
    timed("affine_assemble  A 4096x4096 -> LinearTerms", "affine_tile_kernel<LT>", 24.0 * r * n,
          lambda: _lib.call("pmt_affine_assemble_f64", d(A), lda, r, n, d(xvar), d(b), -1, d(lt), d(consts), s))
    vat, vc = i64(3 * m * n), f64(m)
    timed("affine_pack_vector  C 512x4096 -> VectorAffineTerms", "affine_tile_kernel<VAT>", 32.0 * m * n,
          lambda: _lib.call("pmt_affine_pack_vector_f64", d(Cm), ldc, m, n, d(xvar), d(dd), -1, d(xvar), 0, d(vat), d(vc), s))
    vat2 = i64(3 * r * n)
    timed("affine_pack_vector  A 4096x4096 -> VectorAffineTerms", "affine_tile_kernel<VAT>", 32.0 * r * n,
          lambda: _lib.call("pmt_affine_pack_vector_f64", d(A), lda, r, n, d(xvar), d(b), -1, d(xvar), 0, d(vat2), d(consts), s))
    # literal expansion: 16 rows of 1024-term affine functions -> 16.8 M quadratic terms (the 'auto' mode limit)
    rows, nx = 16, 1024
    xt, xc = i64(2 * rows * nx), f64(rows)
    _lib.call("pmt_affine_assemble_f64", d(A), lda, rows, nx, d(xvar), d(b), -1, d(xt), d(xc), s)
    oq, ol, oc = i64(3 * rows * nx * nx), i64(2 * 2 * rows * nx), f64(1)
    timed("quad_expand  16 x (1024 . 1024) -> 16.8 M QuadraticTerms (MOI)", "quad_expand_kernel", 24.0 * rows * nx * nx,
          lambda: _lib.call("pmt_quad_expand_f64", rows, d(xt), nx, d(xc), d(xt), nx, d(xc), 1, d(xvar), d(oq), d(ol), d(oc), s))
    # MOI copies of materialised functions
    out_lt = i64(2 * r * n)
    timed("pack_scalar_affine  16.8 M LinearTerms", "pack_scalar_affine", 32.0 * r * n,
          lambda: _lib.call("pmt_pack_scalar_affine_f64", d(lt), r * n, d(xvar), d(out_lt), s))
    out_vat = i64(3 * r * n)
    timed("pack_vector_affine  4096 rows x 4096 LinearTerms", "pack_vector_affine", 40.0 * r * n,
          lambda: _lib.call("pmt_pack_vector_affine_f64", d(lt), None, r, n, d(xvar), 0, d(out_vat), s))
    nq = rows * nx * nx
    out_q = i64(3 * nq)
    timed("pack_scalar_quadratic  16.8 M QuadraticTerms", "pack_scalar_quadratic", 48.0 * nq,
          lambda: _lib.call("pmt_pack_scalar_quadratic_f64", d(oq), nq, d(xvar), d(out_q), s))
    # bilinear x'Qx, n = 4096
    timed("bilinear  x'Qx, Q 4096x4096 -> 16.8 M QuadraticTerms", "bilinear_kernel", 32.0 * n * n,
          lambda: _lib.call("pmt_bilinear_f64", d(A), lda, n, n, d(xvar), d(xvar), 1, d(xvar), d(oq), s))
    # generator
    timed("fill_uniform_matrix  4096x4096", "fill_uniform_matrix_kernel", 8.0 * r * n,
          lambda: _lib.call("pmt_fill_uniform_matrix_f64", d(A), r, n, lda, C.c_uint64(7), 1.0, s))


if __name__ == "__main__":
    main()
