"""-m gpu: sparse constraint matrix (BASELINE config 5) through the host API: the constraint C*x - d with a sparse C yields the
reference's dense matvecmul! output minus the structural zeros, in the reference's row-major order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sp = pytest.importorskip("scipy.sparse")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import Variable  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _model(m, n, density, seed):
    rng = np.random.default_rng(seed)
    Cs = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=lambda k: rng.random(k) + 0.1)
    model = P.Model(P.MockOptimizer(variable_offset=5))
    x = [Variable(model) for _ in range(n)]
    def upd(C):
        C.data[:] = rng.random(C.nnz) + 0.1                      # values change, pattern fixed
    Cp = P.Parameter(upd, Cs, model)
    d = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(m)), np.zeros(m), model)
    return model, x, Cp, d


def test_sparse_constraint_matches_dense_reference_without_structural_zeros():
    m, n = 37, 61
    model, x, Cp, d = _model(m, n, 0.08, 1)
    expr = Cp * x - d
    P.constraint(model, expr, "<=", np.zeros(m)) if False else P.constraint(model, Cp * x == d)
    for _ in range(3):
        P.solve(model)
        f = list(model.constraints)[0].f
        dense = O.AffVec(m).vecsubtract(O.AffVec(m).matvecmul_vars(Cp().toarray(), [v.index for v in x]), d())
        terms, consts = dense.moi(model.model_var_to_optimizer)
        want = terms[terms["coeff"] != 0.0]
        assert np.array_equal(f.terms.view(np.int64), want.view(np.int64))
        assert np.array_equal(f.constants, consts)
    # native (LinearTerm) form of the same node
    got = expr()
    rows = O.AffVec(m).vecsubtract(O.AffVec(m).matvecmul_vars(Cp().toarray(), [v.index for v in x]), d()).as_tuples()
    for g_, (t, c) in zip(got, rows):
        assert [(tt.coeff, tt.var.index) for tt in g_.linear] == [p for p in t if p[0] != 0.0] and g_.constant == c


def test_sparse_pattern_must_stay_fixed():
    model, x, Cp, d = _model(8, 9, 0.3, 2)
    P.constraint(model, Cp * x == d)
    P.solve(model)
    Cp.val = sp.random(8, 9, density=0.5, format="csc", random_state=np.random.default_rng(9))
    Cp.f = lambda C: None
    with pytest.raises(P.DimensionMismatch):
        P.solve(model)


def test_config5_size_properties():
    # n = 16384, m = 4096, 5 % non-zeros: sortedness of (row, col) and a checksum instead of an oracle run
    m, n = 4096, 16384
    rng = np.random.default_rng(3)
    nnz_per_col = int(0.05 * m)
    indptr = np.arange(0, (n + 1) * nnz_per_col, nnz_per_col, dtype=np.int64)
    indices = np.concatenate([np.sort(rng.choice(m, nnz_per_col, replace=False)) for _ in range(n)]).astype(np.int64)
    data = rng.random(indices.size) + 0.1
    Cs = sp.csc_matrix((data, indices, indptr), shape=(m, n))
    model = P.Model(P.MockOptimizer())
    x = [Variable(model) for _ in range(n)]
    Cp = P.Parameter(model, val=Cs)
    d = P.Parameter(model, val=rng.random(m))
    P.constraint(model, Cp * x == d)
    P.solve(model)
    t = list(model.constraints)[0].f.terms
    assert len(t) == Cs.nnz
    key = t["out"].astype(np.int64) * (n + 1) + t["var"]
    assert np.all(np.diff(key) > 0)                                  # strictly row-major, columns ascending within a row
    assert t["coeff"].sum() == pytest.approx(data.sum(), rel=1e-12)
    csr = Cs.tocsr()
    assert np.array_equal(t["coeff"], csr.data) and np.array_equal(t["var"], csr.indices + 1)
    # a ROW BAND of the full-size block against the ORACLE, byte for byte: the reference's dense matvecmul! + vecsubtract!
    # on the densified band and update!(::MOI.VectorAffineFunction) (src/moi_interop.jl:64-81), minus the structural zeros; the band straddles
    # the kernel's 128-row block boundary
    f = list(model.constraints)[0].f
    r0, r1 = 1000, 1160
    band = csr[r0:r1].toarray()
    xi = np.arange(1, n + 1, dtype=np.int64)
    wt, wc = O.AffVec(r1 - r0).vecsubtract(O.AffVec(r1 - r0).matvecmul_vars(band, xi), d()[r0:r1]).moi(model.model_var_to_optimizer)
    wt = wt[wt["coeff"] != 0.0]
    wt["out"] += r0                                                  # the band's rows are rows r0+1 .. r1 of the function
    lo_, hi_ = int(csr.indptr[r0]), int(csr.indptr[r1])
    assert hi_ - lo_ == len(wt) and np.array_equal(f.terms[lo_:hi_].view(np.int64), wt.view(np.int64))
    assert np.array_equal(f.constants[r0:r1].view(np.int64), wc.view(np.int64))


@pytest.mark.parametrize("m,n,density,nslab", [(37, 61, 0.08, 8), (5, 3, 0.9, 8), (200, 1000, 0.3, 8), (64, 500, 0.02, 3), (9, 40, 0.0, 8), (300, 2100, 0.2, 8), (1, 700, 0.5, 8)])
def test_xcd_aware_slab_kernels_equal_the_flat_scatter(m, n, density, nslab):
    """pmt_sparse_pack_vector_slabs_f64 / pmt_sparse_assemble_slabs_f64 (one column slab per XCD, 16-byte chunk writes) write the same
    bytes as the flat gather kernels — rows with no entry in a slab, slabs wider than the matrix, segments longer than one wave
    (> 64 terms) and an empty matrix included."""
    import ctypes as C
    import gpu_util as g
    rng = np.random.default_rng(m * n)
    csc = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=lambda k: rng.random(k) - 0.5)
    nnz = csc.nnz
    colptr, rowval = csc.indptr.astype(np.int64) + 1, csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.zeros(max(nnz, 1), dtype=np.int64) for _ in range(3))
    rptr = np.zeros(m + 1, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    g.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    slab = np.zeros(m * (nslab + 1), dtype=np.int64)
    g.call("pmt_sparse_slab_ptr", m, n, nslab, vp(rptr), vp(tcol), vp(slab))
    sl = slab.reshape(m, nslab + 1)
    assert np.array_equal(sl[:, 0], rptr[:-1]) and np.array_equal(sl[:, -1], rptr[1:]) and np.all(np.diff(sl, axis=1) >= 0)
    xvar = np.arange(n, 0, -1, dtype=np.int64)                             # any variables
    tvar = xvar[tcol[:nnz] - 1] if nnz else np.zeros(1, dtype=np.int64)
    varmap = np.random.default_rng(1).permutation(n).astype(np.int64) + 1
    d_nz, d_perm, d_row, d_var = g.to_dev(csc.data if nnz else np.zeros(1)), g.to_dev(perm), g.to_dev(trow), g.to_dev(tvar)
    d_slab, d_vm = g.to_dev(slab), g.to_dev(varmap)
    flat_v, slab_v = g.empty_terms(max(nnz, 1), g.VAT), g.empty_terms(max(nnz, 1), g.VAT)
    flat_l, slab_l = g.empty_terms(max(nnz, 1), g.LT), g.empty_terms(max(nnz, 1), g.LT)
    g.call("pmt_sparse_pack_vector_f64", g.ptr(d_nz), g.ptr(d_perm), g.ptr(d_row), g.ptr(d_var), nnz, g.ptr(d_vm), 7, g.ptr(flat_v), g.stream())
    g.call("pmt_sparse_pack_vector_slabs_f64", g.ptr(d_nz), g.ptr(d_perm), g.ptr(d_var), g.ptr(d_slab), m, nslab, g.ptr(d_vm), 7, g.ptr(slab_v), g.stream())
    g.call("pmt_sparse_assemble_f64", g.ptr(d_nz), g.ptr(d_perm), g.ptr(d_var), nnz, g.ptr(flat_l), g.stream())
    g.call("pmt_sparse_assemble_slabs_f64", g.ptr(d_nz), g.ptr(d_perm), g.ptr(d_var), g.ptr(d_slab), m, nslab, g.ptr(slab_l), g.stream())
    # 32-bit index streams, once with the varmap gather and once with varmap folded into the variable stream
    d_perm32, d_var32, d_mapped32 = g.to_dev(perm.astype(np.uint32)), g.to_dev(tvar.astype(np.uint32)), g.to_dev(varmap[tvar - 1].astype(np.uint32))
    d_mapped = g.to_dev(varmap[tvar - 1])
    n32_v, f32_v, f64_v, n32_l = g.empty_terms(max(nnz, 1), g.VAT), g.empty_terms(max(nnz, 1), g.VAT), g.empty_terms(max(nnz, 1), g.VAT), g.empty_terms(max(nnz, 1), g.LT)
    g.call("pmt_sparse_pack_vector_slabs_u32_f64", g.ptr(d_nz), g.ptr(d_perm32), g.ptr(d_var32), g.ptr(d_slab), m, nslab, g.ptr(d_vm), 7, g.ptr(n32_v), g.stream())
    g.call("pmt_sparse_pack_vector_slabs_u32_f64", g.ptr(d_nz), g.ptr(d_perm32), g.ptr(d_mapped32), g.ptr(d_slab), m, nslab, None, 7, g.ptr(f32_v), g.stream())
    g.call("pmt_sparse_pack_vector_slabs_f64", g.ptr(d_nz), g.ptr(d_perm), g.ptr(d_mapped), g.ptr(d_slab), m, nslab, None, 7, g.ptr(f64_v), g.stream())
    g.call("pmt_sparse_assemble_slabs_u32_f64", g.ptr(d_nz), g.ptr(d_perm32), g.ptr(d_var32), g.ptr(d_slab), m, nslab, g.ptr(n32_l), g.stream())
    if nnz:
        g.assert_terms_equal(g.terms_to_host(slab_v, nnz, g.VAT), g.terms_to_host(flat_v, nnz, g.VAT))
        g.assert_terms_equal(g.terms_to_host(slab_l, nnz, g.LT), g.terms_to_host(flat_l, nnz, g.LT))
        for other in (n32_v, f32_v, f64_v):
            g.assert_terms_equal(g.terms_to_host(other, nnz, g.VAT), g.terms_to_host(flat_v, nnz, g.VAT))
        g.assert_terms_equal(g.terms_to_host(n32_l, nnz, g.LT), g.terms_to_host(flat_l, nnz, g.LT))
        t = g.terms_to_host(slab_v, nnz, g.VAT)
        model_var = np.argsort(varmap)[t["var"] - 1] + 1                   # undo varmap; column c carries Variable n - c
        dense = np.zeros((m, n))
        dense[t["out"] - 8, n - model_var] = t["coeff"]                     # row_offset 7, 1-based rows
        assert np.array_equal(dense, csc.toarray())


def test_device_resident_sparse_parameter_matches_the_host_updated_one():
    """DeviceUniformSparseParameter: nzval regenerated on the device (config 5 at the headline's boundary) — the MOI triplets equal those
    of a host-updated sparse Parameter holding the same values, and they change from solve to solve"""
    import scipy.sparse as sp
    import parametron_jl_amd as P
    from oracle import oracle as O
    m, n, k = 96, 200, 7
    rng = np.random.default_rng(2)
    indptr = np.arange(0, (n + 1) * k, k, dtype=np.int64)
    indices = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for _ in range(n)]).astype(np.int64)
    pattern = sp.csc_matrix((np.ones(indices.size), indices, indptr), shape=(m, n))
    model = P.Model(P.MockOptimizer())
    x = [P.Variable(model) for _ in range(n)]
    Cp = P.DeviceUniformSparseParameter(pattern, 3, model)
    d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    P.constraint(model, Cp * x == d)
    prev = None
    for epoch in range(3):
        P.solve(model)
        f = list(model.constraints)[0].f
        vals = O.fill_uniform(indices.size, Cp.current_seed())                 # the same counter-based stream on the CPU
        dense = sp.csc_matrix((vals, indices, indptr), shape=(m, n)).toarray()
        got = np.zeros((m, n)); got[f.terms["out"] - 1, f.terms["var"] - 1] = f.terms["coeff"]
        assert np.array_equal(got, dense)
        assert np.array_equal(f.constants, 0.0 - O.fill_uniform(m, d.current_seed(), 2.0))
        assert np.array_equal(Cp().toarray(), dense)                            # host copy on demand
        assert prev is None or not np.array_equal(prev, dense)
        prev = dense
    model.close()


def _pattern(m, n, density, seed, dense_cols=(), empty_rows=()):
    rng = np.random.default_rng(seed)
    C = sp.random(m, n, density=density, format="lil", random_state=rng, data_rvs=lambda k: rng.random(k) + 0.1)
    for c in dense_cols:                                             # a column with every row set: runs of 128 inside one row block
        C[:, c] = (rng.random(m) + 0.1).reshape(-1, 1)
    for r in empty_rows:
        C[r, :] = 0.0
    C = C.tocsc()
    C.eliminate_zeros()
    C.sort_indices()
    return C


@pytest.mark.parametrize("m,n,density,kw", [
    (300, 3000, 0.05, {}),                                           # three row blocks (the last one ragged), bands of 1024 with a ragged last one
    (128, 1024, 0.04, {}),                                           # exactly one block
    (513, 2049, 0.02, {"dense_cols": (0, 700, 2048), "empty_rows": (0, 17, 512)}),
    (40, 70, 0.5, {}),                                               # smaller than a block
    (1000, 900, 0.3, {}),                                            # dense enough to force narrower bands than 1024
    (260, 5000, 0.06, {"dense_cols": tuple(range(100, 140))}),      # a cluster of full columns: per-block counts far above the average
])
def test_block_kernels_equal_the_flat_scatter(m, n, density, kw):
    """pmt_sparse_pack_vector_blocks_f64 / pmt_sparse_assemble_blocks_f64 (CSC -> row-major through LDS) against the flat gather kernels,
    bit for bit, with a permuting varmap and a row offset."""
    import ctypes as C
    from parametron_jl_amd import _lib
    csc = _pattern(m, n, density, 11, **kw)
    nnz = csc.nnz
    colptr, rowval = csc.indptr.astype(np.int64) + 1, csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.empty(nnz, dtype=np.int64) for _ in range(3))
    rptr = np.empty(m + 1, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    cw = C.c_int(0)
    _lib.call("pmt_sparse_blocks_width", m, n, vp(colptr), vp(rowval), C.byref(cw))
    cw = cw.value
    assert cw >= 32 and cw & (cw - 1) == 0
    nrb, ncb = -(-m // 128), -(-n // cw)
    desc, idx, band = np.zeros(nrb * n, dtype=np.uint64), np.zeros(nnz, dtype=np.uint32), np.zeros(m * (ncb + 1), dtype=np.int64)
    _lib.call("pmt_sparse_blocks_build", m, n, vp(colptr), vp(rowval), vp(perm), vp(tcol), vp(rptr), cw, vp(desc), vp(idx), vp(band))
    # every block fits the kernel's LDS buffer, and the widest admissible band was chosen
    counts = np.zeros((nrb, ncb), dtype=np.int64)
    np.add.at(counts, ((trow - 1) // 128, (tcol - 1) // cw), 1)
    assert counts.max() <= 7168
    dev = torch.device("cuda:0")
    dp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(5)
    xvar = rng.permutation(np.arange(1, n + 1)).astype(np.int64) + 3                 # x[col]
    varmap = rng.permutation(np.arange(1, n + 4)).astype(np.int64) + 100            # optimizer index of model variable v at varmap[v - 1]
    nz = torch.from_numpy(csc.data.copy()).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else np.ascontiguousarray(a)).to(dev)
    dperm, drow, dtvar = t(perm), t(trow), t(xvar[tcol - 1])
    ddesc, dband, dxvar, dvarmap = t(desc), t(band), t(xvar), t(varmap)
    didx = torch.from_numpy(idx.view(np.int32)).to(dev)
    want = torch.zeros(nnz * 3, dtype=torch.int64, device=dev)
    got = torch.zeros(nnz * 3, dtype=torch.int64, device=dev)
    want_lt = torch.zeros(nnz * 2, dtype=torch.int64, device=dev)
    got_lt = torch.zeros(nnz * 2, dtype=torch.int64, device=dev)
    dvec = torch.from_numpy(rng.random(m) - 0.5).to(dev)                        # d of C*x - d: the constants come out of the same launch
    gconst = torch.full((m,), 7.0, dtype=torch.float64, device=dev)
    gconst_lt = torch.full((m,), 7.0, dtype=torch.float64, device=dev)
    for _ in range(2):                                               # second pass with new coefficients: same structure, new values
        _lib.call("pmt_sparse_pack_vector_f64", dp(nz), dp(dperm), dp(drow), dp(dtvar), nnz, dp(dvarmap), 7, dp(want), stream)
        _lib.call("pmt_sparse_pack_vector_blocks_f64", dp(nz), dp(ddesc), dp(didx), dp(dband), dp(dxvar), m, n, nnz, cw, dp(dvarmap), 7, dp(dvec), -1, dp(got), dp(gconst), stream)
        _lib.call("pmt_sparse_assemble_f64", dp(nz), dp(dperm), dp(dtvar), nnz, dp(want_lt), stream)
        _lib.call("pmt_sparse_assemble_blocks_f64", dp(nz), dp(ddesc), dp(didx), dp(dband), dp(dxvar), m, n, nnz, cw, None, 0, dp(got_lt), dp(gconst_lt), stream)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got_lt, want_lt)
        assert torch.equal(gconst, 0.0 - dvec) and torch.equal(gconst_lt, torch.zeros_like(gconst_lt))      # 0 - d (src/functions.jl:751-764); no d: zeros
        nz.copy_(torch.from_numpy(rng.random(nnz) + 0.5).to(dev))
    # against the pattern itself: row-major (row, col) order, coefficients of the CSR form
    g = got_lt.cpu().numpy().reshape(-1, 2)
    csr = sp.csc_matrix((nz.cpu().numpy(), csc.indices, csc.indptr), shape=(m, n)).tocsr()
    csr.sort_indices()
    _lib.call("pmt_sparse_assemble_blocks_f64", dp(nz), dp(ddesc), dp(didx), dp(dband), dp(dxvar), m, n, nnz, cw, None, 0, dp(got_lt), dp(gconst_lt), stream)
    torch.cuda.synchronize()
    g = got_lt.cpu().numpy().reshape(-1, 2)
    assert np.array_equal(g[:, 0].view(np.float64), csr.data) and np.array_equal(g[:, 1], xvar[csr.indices])


def test_block_form_is_the_one_the_config5_model_runs():
    """the host API picks the block form for a config-5-like pattern (and the profile shows its kernel), the slab form for a very sparse one"""
    rng = np.random.default_rng(4)
    for density, want_block in ((0.05, True), (0.002, False)):
        m, n = 256, 4096
        Cs = _pattern(m, n, density, 21)
        model = P.Model(P.MockOptimizer())
        x = [Variable(model) for _ in range(n)]
        Cp = P.Parameter(model, val=Cs)
        d = P.Parameter(model, val=rng.random(m))
        P.constraint(model, Cp * x == d)
        P.solve(model)
        P.profile_enable(True)
        P.solve(model)
        rep = P.profile_report()
        P.profile_enable(False)
        assert ("sparse_block_kernel<VAT>" in rep) == want_block and ("sparse_slab_kernel<VAT,u32>" in rep) == (not want_block)
        t = list(model.constraints)[0].f.terms
        csr = Cs.tocsr()
        csr.sort_indices()
        assert np.array_equal(t["coeff"], csr.data) and np.array_equal(t["var"], csr.indices + 1)
        assert np.array_equal(t["out"], np.repeat(np.arange(1, m + 1), np.diff(csr.indptr)))


class _PermutingMock(P.MockOptimizer):
    """MockOptimizer whose index map permutes the variables and shifts them (MOI.copy_to may return any map, src/model.jl:100-118)"""

    def __init__(self, variable_offset, seed):
        super().__init__(variable_offset)
        self.seed = seed

    def copy_to(self, backend):
        out = super().copy_to(backend)
        out["variables"] = np.random.default_rng(self.seed).permutation(backend.nvars).astype(np.int64) + 1 + self.variable_offset
        return out


def _oracle_sparse_rows(Cs, xidx, dvec, sign):
    """the reference's dense builders on the densified matrix (matvecmul!, src/functions.jl:775-798; vecsubtract!/vecadd!, :840-858)"""
    m = Cs.shape[0]
    mv = O.AffVec(m).matvecmul_vars(Cs.toarray(), xidx)
    return O.AffVec(m).vecaddsub(mv, dvec, sign < 0) if dvec is not None else mv


@pytest.mark.parametrize("m,n,density,kw", [
    (300, 3000, 0.05, {}),                                                           # three row blocks, the last ragged; 1024-wide bands
    (513, 2049, 0.03, {"dense_cols": (0, 700, 2048), "empty_rows": (0, 17, 512)}),
])
def test_block_form_model_against_the_oracle(m, n, density, kw):
    """the kernel config 5 runs (sparse_block_kernel) against the ORACLE — the reference's dense matvecmul!/vecsubtract! output and its
    update!(::MOI.VectorAffineFunction) (src/moi_interop.jl:64-81) minus the structural zeros — through a permuting, shifted index map,
    as MOI triplets (VAT) and as native terms (LT), solve after solve with new values; the profile shows which kernel ran"""
    rng = np.random.default_rng(m + n)
    Cs = _pattern(m, n, density, 31, **kw)
    model = P.Model(_PermutingMock(variable_offset=9, seed=4))
    xs = [Variable(model) for _ in range(n + 3)]
    x = [xs[i] for i in rng.permutation(n + 3)[:n]]                                    # the columns' variables in no particular order
    def upd(Cm):
        Cm.data[:] = rng.random(Cm.nnz) + 0.1
    Cp = P.Parameter(upd, Cs, model)
    d = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(m) - 0.5), np.zeros(m), model)
    expr = Cp * x - d
    P.constraint(model, Cp * x == d)
    xidx = [v.index for v in x]
    P.solve(model)
    assert not np.array_equal(np.sort(model.model_var_to_optimizer), model.model_var_to_optimizer)      # the map does permute
    for it in range(3):
        P.profile_enable(True)
        P.solve(model)
        rep = P.profile_report()
        P.profile_enable(False)
        assert "sparse_block_kernel<VAT>" in rep and "sparse_slab_kernel" not in rep
        f = list(model.constraints)[0].f
        terms, consts = _oracle_sparse_rows(Cp(), xidx, d(), -1).moi(model.model_var_to_optimizer)
        want = terms[terms["coeff"] != 0.0]
        assert len(want) == Cs.nnz
        assert np.array_equal(f.terms.view(np.int64), want.view(np.int64))
        assert np.array_equal(f.constants.view(np.int64), consts.view(np.int64))
    # native (LinearTerm) form of the same node: the LT instantiation of the block kernel
    P.profile_enable(True)
    got = expr()
    rep = P.profile_report()
    P.profile_enable(False)
    assert "sparse_block_kernel<LT>" in rep
    rows = _oracle_sparse_rows(Cp(), xidx, d(), -1).as_tuples()
    assert len(got) == m
    for g_, (t, c) in zip(got, rows):
        assert [(tt.coeff, tt.var.index) for tt in g_.linear] == [p for p in t if p[0] != 0.0] and g_.constant == c


def test_block_kernel_with_varmap_gather_and_row_offset_against_the_oracle():
    """the block kernels' own varmap gather and row offset (a block stacked under another one, vcat!, src/functions.jl:870-885) against the
    oracle's vcat + update!(::MOI.VectorAffineFunction): the second block's triplets are the rows m1+1.. of the stacked function"""
    import ctypes as C
    from parametron_jl_amd import _lib
    m1, m, n = 11, 300, 3000
    csc = _pattern(m, n, 0.05, 41)
    nnz = csc.nnz
    rng = np.random.default_rng(6)
    colptr, rowval = csc.indptr.astype(np.int64) + 1, csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.empty(nnz, dtype=np.int64) for _ in range(3))
    rptr = np.empty(m + 1, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    cw = C.c_int(0)
    _lib.call("pmt_sparse_blocks_width", m, n, vp(colptr), vp(rowval), C.byref(cw))
    cw = cw.value
    nrb, ncb = -(-m // 128), -(-n // cw)
    desc, idx, band = np.zeros(nrb * n, dtype=np.uint64), np.zeros(nnz, dtype=np.uint32), np.zeros(m * (ncb + 1), dtype=np.int64)
    _lib.call("pmt_sparse_blocks_build", m, n, vp(colptr), vp(rowval), vp(perm), vp(tcol), vp(rptr), cw, vp(desc), vp(idx), vp(band))
    xvar = rng.permutation(np.arange(1, n + 1)).astype(np.int64) + 3
    varmap = rng.permutation(np.arange(1, n + 4)).astype(np.int64) + 100
    top = O.AffVec(m1).matvecmul_vars(rng.random((m1, 5)), [1, 2, 3, 4, 5])               # the block above: m1 rows of anything
    dev = torch.device("cuda:0")
    dp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else np.ascontiguousarray(a)).to(dev)
    ddesc, dband, dxvar, dvarmap = t(desc), t(band), t(xvar), t(varmap)
    didx = torch.from_numpy(idx.view(np.int32)).to(dev)
    got = torch.zeros(nnz * 3, dtype=torch.int64, device=dev)
    got_lt = torch.zeros(nnz * 2, dtype=torch.int64, device=dev)
    gconst = torch.full((m,), 7.0, dtype=torch.float64, device=dev)
    gconst_lt = torch.full((m,), 7.0, dtype=torch.float64, device=dev)
    for sign in (-1, 1):
        csc.data[:] = rng.random(nnz) + 0.1
        dvec = rng.random(m) - 0.5
        nz, dd = torch.from_numpy(csc.data.copy()).to(dev), torch.from_numpy(dvec).to(dev)
        _lib.call("pmt_sparse_pack_vector_blocks_f64", dp(nz), dp(ddesc), dp(didx), dp(dband), dp(dxvar), m, n, nnz, cw, dp(dvarmap), m1, dp(dd), sign,
                  dp(got), dp(gconst), stream)
        _lib.call("pmt_sparse_assemble_blocks_f64", dp(nz), dp(ddesc), dp(didx), dp(dband), dp(dxvar), m, n, nnz, cw, dp(dd), sign, dp(got_lt), dp(gconst_lt), stream)
        torch.cuda.synchronize()
        block = _oracle_sparse_rows(csc, xvar, dvec, sign)
        terms, consts = O.AffVec(m1 + m).vcat(top, block).moi(varmap)
        want = terms[(terms["out"] > m1) & (terms["coeff"] != 0.0)]
        assert len(want) == nnz
        assert np.array_equal(got.cpu().numpy(), want.view(np.int64).reshape(-1))
        assert np.array_equal(gconst.cpu().numpy().view(np.int64), consts[m1:].view(np.int64))
        lt, _, lc = block.flat()
        lt = lt[lt["coeff"] != 0.0]
        assert np.array_equal(got_lt.cpu().numpy(), lt.view(np.int64).reshape(-1))
        assert np.array_equal(gconst_lt.cpu().numpy().view(np.int64), lc.view(np.int64))
