"""Generic canonicalize!: the plan-time ordering helpers (CPU) and the device segmented sum (gpu) against the oracle's
canonicalize! (Julia QuickSort restated) — indices bit-exact, coefficients within 1e-12 (summation order differs)."""
import ctypes as C

import numpy as np
import pytest

import parametron_jl_amd as P
from parametron_jl_amd import Variable, _lib
from oracle import oracle as O


def _order_quad(rows, cols):
    n = len(rows)
    vp = C.c_void_p
    rows, cols = np.ascontiguousarray(rows, dtype=np.int64), np.ascontiguousarray(cols, dtype=np.int64)
    perm, seg = np.empty(n, np.int64), np.empty(n + 1, np.int64)
    orow, ocol = np.empty(n, np.int64), np.empty(n, np.int64)
    nseg = C.c_int64()
    _lib.call("pmt_canonical_order_quadratic", n, rows.ctypes.data_as(vp), cols.ctypes.data_as(vp), perm.ctypes.data_as(vp),
              seg.ctypes.data_as(vp), orow.ctypes.data_as(vp), ocol.ctypes.data_as(vp), C.byref(nseg))
    s = nseg.value
    return perm, seg[:s + 1], orow[:s], ocol[:s]


def test_canonical_order_helpers_match_oracle_known_answers():
    # x*y + y*x + y + y + x - y + 4  ->  2*x*y + x + y + 4   (test/functions.jl:15)
    perm, seg, orow, ocol = _order_quad([1, 2], [2, 1])
    assert seg.tolist() == [0, 2] and (orow.tolist(), ocol.tolist()) == ([1], [2])
    # a singleton keeps its original (row, col) (util.jl:18-19): 3*y*x alone stays (2, 1)
    perm, seg, orow, ocol = _order_quad([2, 1], [1, 1])
    assert (orow.tolist(), ocol.tolist()) == ([1, 2], [1, 1])
    rng = np.random.default_rng(0)
    rows, cols = rng.integers(1, 9, 500), rng.integers(1, 9, 500)
    perm, seg, orow, ocol = _order_quad(rows, cols)
    q = O.Quad(quad=[(1.0, int(r), int(c)) for r, c in zip(rows, cols)]).canonicalize()
    t = q.terms()
    assert (orow.tolist(), ocol.tolist()) == (t["row"].tolist(), t["col"].tolist())
    counts = np.diff(seg)
    assert t["coeff"].tolist() == counts.astype(float).tolist()


@pytest.mark.gpu
def test_device_canonicalize_matches_oracle():
    pytest.importorskip("torch")
    model = P.mock_model(quadratic_mode="literal")
    n, r = 6, 5
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(1)
    A = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape) - 0.3), np.zeros((r, n)), model)
    b = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(r)), np.zeros(r), model)
    w = P.Parameter(model, val=rng.random(n))
    residual = A * x - b
    expr = P.dot(residual, residual) + P.dot(w, x)                         # quadratic + affine: not the plain Gram pattern
    canon = expr.canonicalize()
    for _ in range(3):
        model.setdirty()
        got = canon()
        ref = O.LsqWorkspace(n, r, 1)
        xi = np.arange(1, n + 1, dtype=np.int64)
        ref.eval_objective(np.asfortranarray(A()).reshape(-1, order="F"), b(), xi)
        q = ref.objective
        for c, v in zip(w(), xi):
            q.affine.push(float(c), int(v))
        q.canonicalize()
        rq, rl, rc = q.as_tuple()
        assert [(t.rowvar.index, t.colvar.index) for t in got.quadratic] == [(r_, c_) for _, r_, c_ in rq]
        assert [t.var.index for t in got.affine.linear] == [v for _, v in rl]
        np.testing.assert_allclose([t.coeff for t in got.quadratic], [c for c, _, _ in rq], rtol=1e-12)
        np.testing.assert_allclose([t.coeff for t in got.affine.linear], [c for c, _ in rl], rtol=1e-12)
        assert got.affine.constant == rc


@pytest.mark.gpu
def test_canonical_mode_applies_generic_canonicalize_to_non_gram_objectives():
    pytest.importorskip("torch")
    from qp_solver import DenseQPOptimizer
    n = 5
    results = []
    for mode in ("literal", "canonical"):
        model = P.Model(DenseQPOptimizer(), quadratic_mode=mode)
        x = [Variable(model) for _ in range(n)]
        rng = np.random.default_rng(7)
        A = P.Parameter(model, val=rng.random((n + 2, n)))
        b = P.Parameter(model, val=rng.random(n + 2))
        w = P.Parameter(model, val=rng.random(n))
        res = A * x - b
        P.objective(model, P.Minimize, P.dot(res, res) + P.dot(w, x))
        P.solve(model)
        results.append((P.value(model, x), len(model.objective.f.quadratic_terms)))
    np.testing.assert_allclose(results[0][0], results[1][0], rtol=1e-9)
    assert results[0][1] == (n + 2) * n * n and results[1][1] == n * (n + 1) // 2


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,nvars", [("quad", 20000, 37), ("quad", 1, 5), ("aff", 5000, 11), ("quad", 300, 300), ("aff", 0, 3)])
def test_device_ordering_equals_host_ordering(kind, n, nvars):
    """pmt_canonical_order_device (radix sort + run boundaries in HBM) against the host ordering pmt_canonical_order_* on random term lists
    with many duplicates: the same permutation (both are stable), the same run boundaries, the same output indices — including the
    reference's convention that a run of one keeps its original (row, col) (src/util.jl:18-19)."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    import gpu_util as g
    from parametron_jl_amd.lazyexpression import _canonical_order
    rng = np.random.default_rng(n + nvars)
    dtype = g.QT if kind == "quad" else g.LT
    t = np.zeros(n, dtype=dtype)
    t["coeff"] = rng.random(n)
    empty = (np.zeros(0, np.int64), np.zeros(1, np.int64))
    if kind == "quad":
        t["row"] = rng.integers(1, nvars + 1, n); t["col"] = rng.integers(1, nvars + 1, n)
        perm, seg, (orow, ocol) = _canonical_order("quad", t["row"], t["col"]) if n else empty + ((np.zeros(0, np.int64),) * 2,)
    else:
        t["var"] = rng.integers(1, nvars + 1, n)
        perm, seg, (ov,) = _canonical_order("aff", t["var"]) if n else empty + ((np.zeros(0, np.int64),),)
    dt = g.to_dev(t) if n else g.empty_terms(1, dtype)
    dperm = torch.full((max(n, 1),), -1, dtype=torch.int64, device=g.DEV)
    dseg = torch.full((n + 1,), -1, dtype=torch.int64, device=g.DEV)
    nseg = C.c_int64(-1)
    torch.cuda.synchronize()
    g.call("pmt_canonical_order_device", g.ptr(dt), n, dtype.itemsize, g.ptr(dperm), g.ptr(dseg), C.byref(nseg), g.stream())
    torch.cuda.synchronize()
    assert nseg.value == len(seg) - 1
    assert np.array_equal(dperm.cpu().numpy()[:n], perm)
    assert np.array_equal(dseg.cpu().numpy()[:nseg.value + 1], seg)
    out = g.empty_terms(max(nseg.value, 1), dtype)
    g.call("pmt_canonical_init_terms", g.ptr(dt), dtype.itemsize, g.ptr(dperm), g.ptr(dseg), nseg.value, g.ptr(out), g.stream())
    got = g.terms_to_host(out, nseg.value, dtype)
    assert np.all(got["coeff"] == 0.0)
    if kind == "quad":
        assert np.array_equal(got["row"], orow) and np.array_equal(got["col"], ocol)
    else:
        assert np.array_equal(got["var"], ov)


@pytest.mark.gpu
def test_device_ordering_reports_indices_beyond_the_packed_key():
    import ctypes as C
    torch = pytest.importorskip("torch")
    import gpu_util as g
    from parametron_jl_amd import _lib
    t = np.zeros(4, dtype=g.QT)
    t["row"] = [1, 2, 1 << 33, 4]; t["col"] = [1, 1, 2, 4]
    dt = g.to_dev(t)
    dperm = torch.zeros(4, dtype=torch.int64, device=g.DEV); dseg = torch.zeros(5, dtype=torch.int64, device=g.DEV)
    nseg = C.c_int64()
    torch.cuda.synchronize()
    with pytest.raises(_lib.ArgumentError, match="host ordering"):
        g.call("pmt_canonical_order_device", g.ptr(dt), 4, 24, g.ptr(dperm), g.ptr(dseg), C.byref(nseg), g.stream())
