"""not gpu: the host-side structure of the block form of the sparse node (pmt_sparse_blocks_width / pmt_sparse_blocks_build, sparse.hip)
against a numpy restatement of its definition (include/parametron_hip.h)."""
import ctypes as C

import numpy as np
import pytest

sp = pytest.importorskip("scipy.sparse")

from parametron_jl_amd import _lib  # noqa: E402

vp = lambda a: a.ctypes.data_as(C.c_void_p)


def _order(csc):
    m, n = csc.shape
    nnz = csc.nnz
    colptr, rowval = csc.indptr.astype(np.int64) + 1, csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.empty(nnz, dtype=np.int64) for _ in range(3))
    rptr = np.empty(m + 1, dtype=np.int64)
    _lib.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    return colptr, rowval, perm, trow, tcol, rptr


@pytest.mark.parametrize("m,n,density", [(300, 3000, 0.05), (129, 1025, 0.1), (50, 40, 0.6), (1000, 700, 0.35)])
def test_block_structure_matches_its_definition(m, n, density):
    rng = np.random.default_rng(1)
    csc = sp.random(m, n, density=density, format="csc", random_state=rng)
    csc.sort_indices()
    colptr, rowval, perm, trow, tcol, rptr = _order(csc)
    nnz = csc.nnz
    cw = C.c_int(0)
    _lib.call("pmt_sparse_blocks_width", m, n, vp(colptr), vp(rowval), C.byref(cw))
    cw = cw.value
    assert cw in (32, 64, 128, 256, 512, 1024)
    nrb, ncb = -(-m // 128), -(-n // cw)
    counts = np.zeros((nrb, ncb), dtype=np.int64)
    np.add.at(counts, ((trow - 1) // 128, (tcol - 1) // cw), 1)
    assert counts.max() <= 7168
    if cw < 1024:                                                    # the next wider band would not have fitted
        wide = np.zeros((nrb, -(-n // (2 * cw))), dtype=np.int64)
        np.add.at(wide, ((trow - 1) // 128, (tcol - 1) // (2 * cw)), 1)
        assert wide.max() > 7168
    desc, idx, band = np.zeros(nrb * n, dtype=np.uint64), np.zeros(nnz, dtype=np.uint32), np.zeros(m * (ncb + 1), dtype=np.int64)
    _lib.call("pmt_sparse_blocks_build", m, n, vp(colptr), vp(rowval), vp(perm), vp(tcol), vp(rptr), cw, vp(desc), vp(idx), vp(band))
    p0 = (desc & 0xffffffff).astype(np.int64).reshape(nrb, n)
    ln = ((desc >> 32) & 0xffff).astype(np.int64).reshape(nrb, n)
    lb = (desc >> 48).astype(np.int64).reshape(nrb, n)
    # a column's part of a row block is the run [p0, p0 + len) of the CSC arrays
    for rb in range(nrb):
        for c in range(0, n, max(1, n // 97)):
            rows = csc.indices[csc.indptr[c]:csc.indptr[c + 1]]
            sel = np.nonzero((rows >= rb * 128) & (rows < (rb + 1) * 128))[0]
            assert ln[rb, c] == sel.size
            if sel.size:
                assert p0[rb, c] == csc.indptr[c] + sel[0]
    # LDS slots: the runs of a block, column after column
    for rb in range(nrb):
        for cb in range(ncb):
            l = ln[rb, cb * cw:(cb + 1) * cw]
            assert np.array_equal(lb[rb, cb * cw:(cb + 1) * cw], np.concatenate(([0], np.cumsum(l)[:-1])))
    # the index word of term t: slot of its coefficient | column inside the band << 16
    col = tcol - 1
    row = trow - 1
    slot = lb[row // 128, col] + (perm - p0[row // 128, col])
    assert np.array_equal(idx & 0xffff, slot.astype(np.uint32)) and np.array_equal(idx >> 16, (col % cw).astype(np.uint32))
    # band_ptr: first term of a row at or beyond the band's first column
    for r in range(0, m, max(1, m // 53)):
        cols_r = tcol[rptr[r]:rptr[r + 1]] - 1
        for cb in range(ncb + 1):
            assert band[r * (ncb + 1) + cb] == rptr[r] + np.searchsorted(cols_r, cb * cw)


def test_block_form_is_declined_where_it_does_not_apply():
    cw = C.c_int(7)
    # rows not ascending within a column
    colptr = np.array([1, 3, 4], dtype=np.int64)
    rowval = np.array([2, 1, 1], dtype=np.int64)
    _lib.call("pmt_sparse_blocks_width", 2, 2, vp(colptr), vp(rowval), C.byref(cw))
    assert cw.value == 0
    # no non-zeros
    colptr = np.array([1, 1, 1], dtype=np.int64)
    cw = C.c_int(7)
    _lib.call("pmt_sparse_blocks_width", 2, 2, vp(colptr), vp(rowval), C.byref(cw))
    assert cw.value == 0
    from parametron_jl_amd import ErrorException
    with pytest.raises(Exception):
        colptr = np.array([1, 2, 3], dtype=np.int64)
        rowval = np.array([1, 9], dtype=np.int64)
        _lib.call("pmt_sparse_blocks_width", 2, 2, vp(colptr), vp(rowval), C.byref(cw))


def test_hypersparse_pattern_is_declined_before_its_table_is_sized():
    """a 3e6 x 3e6 pattern with two entries per column: the (row block, 32-column strip) table alone would be ~2.2e9 counters (9 GB); the width
    function must answer 0 (slab form) without allocating it — and without throwing across the C boundary"""
    n = 3_000_000
    colptr = np.arange(1, 2 * n + 2, 2, dtype=np.int64)
    rowval = np.empty(2 * n, dtype=np.int64)
    rowval[0::2] = np.arange(1, n + 1)
    rowval[1::2] = np.minimum(np.arange(1, n + 1) + 7, n)
    rowval[-1] = n; rowval[-2] = n - 1
    cw = C.c_int(7)
    _lib.call("pmt_sparse_blocks_width", n, n, vp(colptr), vp(rowval), C.byref(cw))
    assert cw.value == 0
