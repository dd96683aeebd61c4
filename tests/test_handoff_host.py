"""CPU: the host half of the solver hand-off — pmt_csc_order (structure of the solver's CSC matrices) against scipy's COO -> CSC
conversion on random index sets with duplicates, empty columns and the upper-triangular folding of quadratic terms."""
import numpy as np
import pytest
import scipy.sparse as sp

import parametron_jl_amd as P
from parametron_jl_amd.handoff import _csc_order


@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_csc_order_matches_scipy(seed, upper):
    rng = np.random.default_rng(seed)
    nrows, ncols = (17, 17) if upper else (11, 23)
    nnz = 300
    rows = rng.integers(1, nrows + 1, nnz)
    cols = rng.integers(1, ncols + 1, nnz)
    cols[cols == 5] = 6                                                    # an empty column
    vals = rng.random(nnz)
    perm, seg, col_ptr, row_idx = _csc_order(rows, cols, nrows, ncols, upper)
    r0, c0 = rows - 1, cols - 1
    if upper:
        r0, c0 = np.minimum(r0, c0), np.maximum(r0, c0)
    ref = sp.coo_matrix((vals, (r0, c0)), shape=(nrows, ncols)).tocsc()
    ref.sum_duplicates(); ref.sort_indices()
    assert np.array_equal(col_ptr, ref.indptr) and np.array_equal(row_idx, ref.indices)
    assert sorted(perm.tolist()) == list(range(nnz)) and seg[0] == 0 and seg[-1] == nnz
    sums = np.add.reduceat(vals[perm], seg[:-1])
    np.testing.assert_allclose(sums, ref.data, rtol=1e-14)
    for s in range(len(row_idx)):                                          # runs are homogeneous and keep the original order (stable)
        run = perm[seg[s]:seg[s + 1]]
        assert np.all(r0[run] == row_idx[s]) and np.all(np.diff(run) > 0)


def test_csc_order_empty_and_errors():
    perm, seg, col_ptr, row_idx = _csc_order(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 4, 3, False)
    assert len(perm) == 0 and seg.tolist() == [0] and col_ptr.tolist() == [0, 0, 0, 0] and len(row_idx) == 0
    with pytest.raises(P.DimensionMismatch):
        _csc_order(np.array([0], dtype=np.int64), np.array([1], dtype=np.int64), 2, 2, False)    # indices are 1-based
    with pytest.raises(P.DimensionMismatch):
        _csc_order(np.array([1], dtype=np.int64), np.array([3], dtype=np.int64), 2, 2, False)


def test_host_copy_2d_matches_numpy():
    """pmt_host_copy_2d (host -> host, pitched, worker threads): the dense blocks of a host solver's A whose Parameter the host itself updates
    are copied on the host instead of coming back over PCIe.  No device involved: checked here against numpy for every pitch combination,
    thread counts from 1 to 64, sizes from one word to beyond the per-thread minimum."""
    import ctypes as C
    from parametron_jl_amd import _lib
    rng = np.random.default_rng(3)
    for (rows, cols, spad, dpad, top) in [(1, 1, 0, 0, 0), (512, 4096, 64, 8704 - 512, 4096), (40, 33, 8, 24, 7), (1000, 300, 0, 0, 0), (7, 5000, 1, 2, 1)]:
        for threads in (0, 1, 3, 64):
            src = rng.random((cols, rows + spad))
            dst = np.full((cols, top + rows + dpad), -1.0)
            want = dst.copy()
            want[:, top:top + rows] = src[:, :rows]
            _lib.call("pmt_host_copy_2d", C.c_void_p(dst.ctypes.data + 8 * top), 8 * dst.shape[1], C.c_void_p(src.ctypes.data), 8 * src.shape[1], 8 * rows, cols, threads)
            assert np.array_equal(dst, want), (rows, cols, threads)
    with pytest.raises(_lib.ArgumentError):
        _lib.call("pmt_host_copy_2d", C.c_void_p(dst.ctypes.data), 8, C.c_void_p(src.ctypes.data), 8, 16, 2, 0)      # pitch smaller than a row
    with pytest.raises(_lib.ArgumentError):
        _lib.call("pmt_host_copy_2d", C.c_void_p(dst.ctypes.data), 16, C.c_void_p(src.ctypes.data), 16, 16, 2, 65)
