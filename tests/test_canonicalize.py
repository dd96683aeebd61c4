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
