"""-m gpu: overlapped (staged) uploads of host-updated Parameters (src/parameter.jl:57,88,101-102): a model driven with
stage_parameters() + update() produces, solve after solve, exactly the buffers of the same model driven serially, while every Parameter
changes every solve — column-major, row-major (device transposition), vector, scalar and sparse Parameters, in-place and `val=` forms."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402


def _build(pinned, rowmajor):
    import scipy.sparse as sp
    n, r, m = 70, 90, 12
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [P.Variable(model) for _ in range(n)]
    alloc = model.parameter_array if pinned else (lambda *s: np.zeros(s, order="C" if rowmajor else "F"))
    bufs = {"A": alloc(r, n), "b": alloc(r), "G": alloc(m, n), "h": alloc(m), "l": alloc(n)}
    A, b, G, h, l = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "G", "h", "l"))
    state = {"w": 1.0}
    w = P.Parameter(lambda: state["w"], model)                               # out-of-place scalar
    pattern = sp.random(m, n, density=0.2, format="csc", random_state=np.random.default_rng(7))
    pattern.data[:] = 1.0
    S = P.Parameter(model, val=pattern)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, G * x, "<=", h)
    P.constraint(model, x, ">=", l)
    P.constraint(model, S * x == h)
    P.constraint(model, w * x[0] + x[1], "<=", 3.0)
    return model, bufs, state, pattern


def _randomise(rng, bufs, state, pattern):
    for a in bufs.values():
        a[...] = rng.random(a.shape)
    state["w"] = float(rng.random()) + 0.5
    pattern.data[:] = rng.random(pattern.nnz) + 0.1


def _snapshot(model):
    f = model.objective.f
    out = [f.quadratic_terms.copy(), f.affine_terms.copy(), np.array([f.constant])]
    for c in model.constraints:
        if hasattr(c.f, "terms"):
            out.append(np.array(c.f.terms).copy())
        out.append(np.array(c.f.constants if hasattr(c.f, "constants") else [c.f.constant]).copy())
    return out


@pytest.mark.parametrize("pinned,rowmajor", [(True, False), (False, False), (False, True)])
def test_staged_updates_equal_serial_updates(pinned, rowmajor):
    serial = _build(pinned, rowmajor)
    staged = _build(pinned, rowmajor)
    rngs = [np.random.default_rng(11), np.random.default_rng(11)]
    for (model, bufs, state, pattern), rng in zip((serial, staged), rngs):
        _randomise(rng, bufs, state, pattern)
        P.solve(model)
    for it in range(4):
        m1, b1, s1, p1 = serial
        _randomise(rngs[0], b1, s1, p1)
        P.solve(m1)
        want = _snapshot(m1)
        m2, b2, s2, p2 = staged
        m2.wait_staged()                                                     # the previous staged copy has left the host buffers
        _randomise(rngs[1], b2, s2, p2)
        m2.stage_parameters()                                                # values of this solve -> copy stream
        before = m2.device().bytes_allocated() if it > 0 else None
        P.solve(m2)                                                          # consumes the staged values (no second evaluation, no serial upload)
        if before is not None:
            assert m2.device().bytes_allocated() == before                   # steady state allocates nothing
        got = _snapshot(m2)
        assert len(got) == len(want)
        for g, w_ in zip(got, want):
            assert g.dtype == w_.dtype and g.shape == w_.shape
            assert g.tobytes() == w_.tobytes(), "staged update differs from the serial update in iteration %d" % it


def test_staged_values_are_not_evaluated_twice():
    model = P.mock_model()
    x = [P.Variable(model) for _ in range(3)]
    calls = {"n": 0}

    def upd(v):
        calls["n"] += 1
        v[:] = calls["n"]
    p = P.Parameter(upd, np.zeros(3), model)
    P.constraint(model, x, ">=", p)
    P.solve(model)
    base = calls["n"]
    model.stage_parameters()
    assert calls["n"] == base + 1
    P.solve(model)
    assert calls["n"] == base + 1                                            # consumed, not re-evaluated
    c = list(model.constraints)[0]
    assert np.array_equal(np.array(c.f.constants), 0.0 - np.full(3, float(base + 1)))
    P.solve(model)                                                           # nothing staged: evaluated as usual
    assert calls["n"] == base + 2
