"""-m gpu: transcription of test/lazyexpression.jl — every rewrite rule evaluated on the device equals the out-of-place
expression (computed by the host algebra exactly as Julia would), and steady-state evaluation allocates nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import Variable, hostops  # noqa: E402


def V(*idx):
    return [Variable(i) for i in idx]


def no_alloc(model, expr):
    expr()
    before = model.device().bytes_allocated()
    model.setdirty()
    expr()
    return model.device().bytes_allocated() == before


def test_mul_optimization():                                   # test/lazyexpression.jl:138-148
    m = P.mock_model()
    weight = P.Parameter(lambda: 3, m)
    x = V(1, 2, 3)
    expr = weight * hostops.vecdot(x, x)                       # @expression weight * (x ⋅ x)
    vals = {x[0]: 1, x[1]: 2, x[2]: 3}
    assert expr()(vals) == 3 * (1 + 4 + 9)
    assert expr() == 3 * hostops.vecdot(x, x)
    assert no_alloc(m, expr)


def test_scale_optimization():                                 # test/lazyexpression.jl:150-185
    model = P.mock_model()
    x = [Variable(model) for _ in range(3)]
    dt = P.Parameter(lambda: 2.0, model)
    e = dt * np.array([1.0, 2, 3, 4, 5])                        # numbers only
    assert np.array_equal(e(), 2.0 * np.array([1.0, 2, 3, 4, 5]))
    for expr in (dt * x, x * dt):
        assert expr() == hostops.scale(2.0, x)
        assert no_alloc(model, expr)
    Ax = hostops.matvecmul(np.ones((3, 3)), x)
    for expr in (dt * Ax, Ax * dt):
        assert expr() == hostops.scale(2.0, Ax)
        assert no_alloc(model, expr)


def test_vcat_optimization():                                  # test/lazyexpression.jl:188-245
    rng = np.random.default_rng(42)
    m = P.mock_model()
    A = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape)), np.zeros((3, 4)), m)
    B = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape)), np.zeros((3, 3)), m)
    x, y, z = V(1, 2, 3, 4), V(5, 6, 7), V(8, 9)
    Cm = P.Parameter(lambda: np.array([[1.0, 2.0], [3.0, 4.0]]), m)
    f1, f2, f3 = A * x, B * y, Cm * z
    v1 = P.vcat(f1, f2)
    assert v1() == f1() + f2()
    assert no_alloc(m, v1)
    v3 = P.vcat(f3, f3)
    assert v3() == f3() + f3() and len(v3()) == 4
    v4 = P.vcat(f3)
    assert v4() == f3()
    v5 = P.vcat(f1, f2, f3)                                    # ragged rows: 4, 3 and 2 terms
    m.setdirty()
    assert v5() == f1() + f2() + f3()
    assert no_alloc(m, v5)
    assert f1() == hostops.matvecmul(A(), x)


def test_vect_optimization():                                  # test/lazyexpression.jl:247-263
    m = P.mock_model()
    x = V(1, 2)
    p = P.Parameter(lambda: np.array([1.0, 2.0]), m)
    expr = P.vect(P.dot(p, x))
    assert expr() == [hostops.vecdot(p(), x)]
    assert no_alloc(m, expr)


def test_adjoint_optimization():                               # test/lazyexpression.jl:300-314, :406-414
    rng = np.random.default_rng(0)
    m = P.mock_model()
    x = V(1, 2)
    p = P.Parameter(lambda: np.array([[1.0, 2.0], [3.0, 4.0]]), m)
    ex = P.adjoint(p) * x
    assert ex() == hostops.matvecmul(p().T, x)
    p2 = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape)), np.zeros((2, 2)), m)
    ex2 = p2.T * x
    assert ex2() == hostops.matvecmul(p2().T, x)
    assert no_alloc(m, ex2)
    n, mm = 5, 15
    Adata = rng.standard_normal((n, mm))
    A = P.Parameter(m, val=Adata)
    xs = V(*range(1, n + 1))
    assert (A.T * xs)() == hostops.matvecmul(Adata.T, xs)


def test_issue_26():                                           # test/lazyexpression.jl:316-328
    model = P.mock_model()
    n = 2
    x = V(1, 2)
    rng = np.random.default_rng(1)
    vals = {x[0]: rng.random(), x[1]: rng.random()}
    def upd(q):
        q[0] = 1; q[1] = 2
    q = P.Parameter(upd, np.zeros(n), model)
    expr1 = P.transpose(x) * np.eye(n) * x + q.T * x
    expr2 = q.T * x + P.transpose(x) * np.eye(2) * x
    assert expr1()(vals) == pytest.approx(expr2()(vals), abs=1e-14)
    want = vals[x[0]] ** 2 + vals[x[1]] ** 2 + vals[x[0]] + 2 * vals[x[1]]
    assert expr1()(vals) == pytest.approx(want, abs=1e-14)


def test_issue_32():                                           # test/lazyexpression.jl:330-347
    rng = np.random.default_rng(2)
    model = P.mock_model()
    v = [Variable(model) for _ in range(2)]
    v0 = np.zeros(2)
    dt = 0.01
    u = [Variable(model) for _ in range(2)]
    def updH(H):
        H[0, 0] = rng.random(); H[1, 1] = rng.random()
    H = P.Parameter(updH, np.zeros((2, 2)), model)
    def updc(c):
        c[0] = rng.random(); c[1] = rng.random()
    c = P.Parameter(updc, np.zeros(2), model)
    vmv0 = hostops.vecaddsub(v, v0, -1)                         # v - v0: no Parameter -> evaluated immediately
    expr = H * vmv0 - dt * (u - c)
    for _ in range(2):
        model.setdirty()
        got = expr()
        want = hostops.vecaddsub(hostops.matvecmul(H(), vmv0), hostops.scale(dt, hostops.vecaddsub(u, c(), -1)), -1)
        assert got == want
    assert no_alloc(model, expr)


def test_issue_23_numbers_only():                              # test/lazyexpression.jl:349-354
    model = P.mock_model()
    p = P.Parameter(lambda: 2, model)
    assert (2 + p)() == 4


def test_bad_syntax_raises_argument_error():                   # test/lazyexpression.jl:21-23
    model = P.mock_model()
    x = V(1, 2)
    p = P.Parameter(lambda: np.eye(2), model)
    with pytest.raises(P.ArgumentError):
        P.lazy("hcat", p, x)
    with pytest.raises(P.DimensionMismatch):
        p * V(1, 2, 3)


def test_findallocs_reports_kernels_and_no_plan_growth():         # test/debug.jl:8-14
    import io
    model = P.mock_model()
    x = Variable(model)
    p = P.Parameter(lambda: 3, model)
    expr = p * 4 + x
    model.setdirty()
    out = io.StringIO()
    P.findallocs(out, expr)
    text = out.getvalue()
    assert "device node" in text and "plan memory growth during re-evaluation: 0 bytes" in text
    out2 = io.StringIO()
    P.findallocs(out2, p)
    assert "no device work" in out2.getvalue()


def test_prune_zero_on_the_device_matches_the_host_rule():       # src/functions.jl:294-297, 409-413; test/functions.jl:9-28
    import ctypes as C
    import gpu_util as g
    # C ABI: stable compaction of LinearTerms and QuadraticTerms, data-dependent count
    rng = np.random.default_rng(0)
    n = 5000
    lt = np.zeros(n, dtype=g.LT); lt["coeff"] = np.where(rng.random(n) < 0.4, 0.0, rng.standard_normal(n) * 1e-3); lt["var"] = rng.integers(1, 50, n)
    lt["coeff"][::7] = -0.0
    for atol in (0.0, 5e-4):
        d_in, d_out, cnt = g.to_dev(lt), g.empty_terms(n, g.LT), torch.zeros(1, dtype=torch.int64, device=g.DEV)
        wsb = g.lib().pmt_prune_zero_workspace_bytes(n, 16)
        ws = g.empty_f64(wsb // 8 + 1)
        g.call("pmt_prune_zero_f64", g.ptr(d_in), n, 16, atol, g.ptr(d_out), g.ptr(cnt), g.ptr(ws), C.c_size_t(wsb), g.stream())
        keep = lt[np.abs(lt["coeff"]) > atol]
        assert int(cnt.item()) == len(keep)
        g.assert_terms_equal(g.terms_to_host(d_out, len(keep), g.LT), keep)
    qt = np.zeros(300, dtype=g.QT); qt["coeff"] = np.where(rng.random(300) < 0.5, 0.0, 1.5); qt["row"] = rng.integers(1, 9, 300); qt["col"] = rng.integers(1, 9, 300)
    d_in, d_out, cnt = g.to_dev(qt), g.empty_terms(300, g.QT), torch.zeros(1, dtype=torch.int64, device=g.DEV)
    wsb = g.lib().pmt_prune_zero_workspace_bytes(300, 24)
    ws = g.empty_f64(wsb // 8 + 1)
    g.call("pmt_prune_zero_f64", g.ptr(d_in), 300, 24, 0.0, g.ptr(d_out), g.ptr(cnt), g.ptr(ws), C.c_size_t(wsb), g.stream())
    keep = qt[np.abs(qt["coeff"]) > 0]
    assert int(cnt.item()) == len(keep)
    g.assert_terms_equal(g.terms_to_host(d_out, len(keep), g.QT), keep)
    # expression level: x'Qx + q'x with zeros in Q and q; the affine part is pruned with the DEFAULT atol (the reference's quirk)
    model = P.mock_model()
    x = [Variable(model) for _ in range(3)]
    Qv = np.array([[1.0, 0.0, 1e-9], [0.0, 0.0, 2.0], [0.0, 0.0, 0.0]])
    Q = P.Parameter(model, val=np.asfortranarray(Qv))
    qv = P.Parameter(model, val=np.array([0.0, 1e-9, 3.0]))
    expr = P.bilinear(x, Q, x) + P.dot(qv, x)
    full = expr()
    got = P.prune_zero(expr, atol=1e-6)                                    # device expression: compacted on the device
    want = P.prune_zero(full, atol=1e-6)                                   # host function: the host rule
    assert repr(got) == repr(want)
    assert len(got.quadratic) == 2 and len(got.affine.linear) == 2         # 1e-9 survives in the affine part (default atol = 0)


def test_matrix_products_of_parameters():                       # generic rule :198 on number-only arguments
    """A * B and A * b of Parameters are Julia matrix products (derived data), recomputed when a source changes."""
    m = P.mock_model()
    Av = np.array([[1.0, 2.0], [3.0, 4.0]])
    A = P.Parameter(m, val=Av)
    B = P.Parameter(lambda: np.array([[5.0, 6.0], [7.0, 8.0]]), m)
    b = P.Parameter(lambda: np.array([5.0, 6.0]), m)
    x = [Variable(m) for _ in range(2)]
    AB, Ab = A * B, A * b
    assert np.array_equal(AB(), Av @ B()) and np.array_equal(Ab(), Av @ b())
    e = AB * x                                                  # feeds matvecmul! on the device
    assert e() == hostops.matvecmul(Av @ B(), x)
    Av[0, 0] = -1.0
    m.setdirty()
    assert np.array_equal(AB(), Av @ B())
    assert e() == hostops.matvecmul(Av @ B(), x)


def test_adjoint_node_feeding_plain_data_is_evaluated():        # round-1 advisor finding: fetch without evaluate
    """dot(A', B) / A' * B read the adjoint node's buffer: it must be (re)evaluated, not just fetched — first call and after
    an update of A."""
    m = P.mock_model()
    Av = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    A = P.Parameter(m, val=Av)
    Bv = np.arange(6.0).reshape(3, 2) + 1
    B = P.Parameter(lambda: Bv, m)
    At = A.T
    d = P.dot(At, B)
    assert d() == float(np.sum(Av.T * Bv))
    prod = At * P.Parameter(lambda: np.array([1.0, -1.0]), m)   # A' * v
    assert np.array_equal(prod(), Av.T @ np.array([1.0, -1.0]))
    Av[:] = Av[::-1].copy()
    m.setdirty()
    assert d() == float(np.sum(Av.T * Bv))
    assert np.array_equal(prod(), Av.T @ np.array([1.0, -1.0]))


def test_scalar_parameter_times_scalar_decision_terms():         # generic rule :198: Number * Variable / LinearTerm / QuadraticTerm
    m = P.mock_model()
    x, y = Variable(m), Variable(m)
    state = {"p": 3.0}
    p = P.Parameter(lambda: state["p"], m)
    e1 = p * x
    assert e1() == P.AffineFunction.of(3.0 * x)
    e2 = 2 * x + p
    assert e2() == 2 * x + 3.0
    e3 = p * x ** 2
    assert e3() == P.QuadraticFunction.of(3.0 * x * x)
    e4 = p * (x * y) + 2 * y
    assert e4()({x: 2.0, y: 5.0}) == 3.0 * 10.0 + 10.0
    state["p"] = -1.5
    m.setdirty()
    assert e1() == P.AffineFunction.of(-1.5 * x)
    assert e3() == P.QuadraticFunction.of(-1.5 * x * x)
    assert no_alloc(m, e1) and no_alloc(m, e3)
