"""CPU: the host half of the sparse constraint node (BASELINE config 5) — row-major order of a CSC pattern and the slab
boundaries (include/parametron_hip.h: pmt_sparse_rowmajor_order, pmt_sparse_slab_ptr).  The
row-major order is matvecmul!'s (src/functions.jl:790-796: row, then ascending column) restricted to the structural non-zeros."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import parametron_jl_amd as P
from parametron_jl_amd import _lib


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def order(csc):
    m, n = csc.shape
    nnz = csc.nnz
    colptr, rowval = csc.indptr.astype(np.int64) + 1, csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.zeros(max(nnz, 1), dtype=np.int64) for _ in range(3))
    rptr = np.zeros(m + 1, dtype=np.int64)
    _lib.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    return colptr, perm, trow, tcol, rptr


@pytest.mark.parametrize("m,n,density,nslab", [(37, 61, 0.08, 8), (5, 3, 0.9, 8), (200, 1000, 0.3, 8), (64, 500, 0.02, 3), (9, 40, 0.0, 8), (1, 700, 0.5, 5)])
def test_rowmajor_order_and_slab_ptr(m, n, density, nslab):
    rng = np.random.default_rng(m * n + nslab)
    csc = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=rng.random)
    csc.sort_indices()
    nnz = csc.nnz
    colptr, perm, trow, tcol, rptr = order(csc)
    csr = csc.tocsr()
    csr.sort_indices()
    assert np.array_equal(rptr, csr.indptr)
    if nnz:
        assert np.array_equal(tcol[:nnz] - 1, csr.indices) and np.array_equal(csc.data[perm[:nnz]], csr.data)
        assert np.array_equal(trow[:nnz] - 1, np.repeat(np.arange(m), np.diff(csr.indptr)))
    slab = np.zeros(m * (nslab + 1), dtype=np.int64)
    _lib.call("pmt_sparse_slab_ptr", m, n, nslab, vp(rptr), vp(tcol), vp(slab))
    sl = slab.reshape(m, nslab + 1)
    assert np.array_equal(sl[:, 0], rptr[:-1]) and np.array_equal(sl[:, -1], rptr[1:]) and np.all(np.diff(sl, axis=1) >= 0)


def test_sparse_host_helpers_reject_bad_input():
    colptr = np.array([0, 1], dtype=np.int64)                              # 0-based colptr
    one = np.ones(1, dtype=np.int64)
    out = [np.zeros(1, dtype=np.int64) for _ in range(3)]
    rptr = np.zeros(2, dtype=np.int64)
    with pytest.raises(P.ArgumentError):
        _lib.call("pmt_sparse_rowmajor_order", 1, 1, vp(colptr), vp(one), vp(out[0]), vp(out[1]), vp(out[2]), vp(rptr))
    colptr = np.array([1, 2], dtype=np.int64)
    bad_row = np.array([3], dtype=np.int64)
    with pytest.raises(P.DimensionMismatch):
        _lib.call("pmt_sparse_rowmajor_order", 1, 1, vp(colptr), vp(bad_row), vp(out[0]), vp(out[1]), vp(out[2]), vp(rptr))
