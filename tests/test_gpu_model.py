"""-m gpu: the host API end to end on the device — transcriptions of test/model.jl and test/lazyexpression.jl.
Expressions are evaluated by the HIP kernels through the plan tape; values are compared with the CPU oracle
(bit-exact for literal forms) and with the reference's closed-form answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import Variable  # noqa: E402
from oracle import oracle as O  # noqa: E402
from qp_solver import DenseQPOptimizer  # noqa: E402


def affvec_tuples(fs):
    return [([(t.coeff, t.var.index) for t in f.linear], f.constant) for f in fs]


def quad_tuples(q):
    return ([(t.coeff, t.rowvar.index, t.colvar.index) for t in q.quadratic], [(t.coeff, t.var.index) for t in q.affine.linear], q.affine.constant)


# ------------------------------------------------------------------ test/lazyexpression.jl:84-108 (matvecmul!, vecsubtract!, dirty flags)
def test_expression_matvecmul_and_residual_match_oracle_and_track_parameters():
    model = P.mock_model()
    rng = np.random.default_rng(1)
    m, n = 5, 7
    x = [Variable(model) for _ in range(n)]
    A = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape)), np.zeros((m, n)), model)
    b = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(v.shape)), np.zeros(m), model)
    Ax = A * x
    residual = Ax - b
    assert P.wrap(P.wrap(residual)) is P.wrap(residual)
    for _ in range(3):
        model.setdirty()
        got = residual()
        ref = O.AffVec(m).vecsubtract(O.AffVec(m).matvecmul_vars(A(), [v.index for v in x]), b())
        assert affvec_tuples(got) == ref.as_tuples()
        assert affvec_tuples(Ax()) == O.AffVec(m).matvecmul_vars(A(), [v.index for v in x]).as_tuples()
    before = model.device().bytes_allocated()
    model.setdirty(); residual(); residual()
    assert model.device().bytes_allocated() == before                    # steady state allocates nothing (↔ @allocated == 0)
    # without setdirty! the Parameter is not re-evaluated (test/parameter.jl semantics through the DAG)
    Aval = A().copy()
    residual()
    assert np.array_equal(A(), Aval)


# ------------------------------------------------------------------ test/lazyexpression.jl:279-298 (vecdot rule) + literal expansion
def test_expression_vecdot_forms():
    model = P.mock_model(quadratic_mode="literal")
    n, r = 4, 3
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(2)
    A = P.Parameter(model, val=rng.random((r, n)))
    b = P.Parameter(model, val=rng.random(r))
    w = P.Parameter(model, val=rng.random(n))
    xi = [v.index for v in x]
    residual = A * x - b
    q = P.dot(residual, residual)()
    w_ref = O.LsqWorkspace(n, r, 1)
    w_ref.eval_objective(np.asfortranarray(A()).reshape(-1, order="F"), b(), np.array(xi, dtype=np.int64))
    assert quad_tuples(q) == w_ref.objective.as_tuple()
    f = P.dot(w, x)()                                                      # p ⋅ x (docstring src/lazyexpression.jl:119-135)
    assert ([(t.coeff, t.var.index) for t in f.linear], f.constant) == O.vecdot_aff_numbers_vars(w(), xi).as_tuple()
    g = P.dot(x, w)()
    assert g == f
    h = P.dot(b, residual)()                                               # numbers . Vector{AffineFunction}
    href = O.vecdot_aff_numbers_affs(b(), O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(A(), xi), b()))
    assert ([(t.coeff, t.var.index) for t in h.linear], h.constant) == href.as_tuple()
    xx = P.dot(residual, x[:r])()                                          # Vector{AffineFunction} . Vector{Variable}
    xref = O.Quad().vecdot_affs_vars(O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(A(), xi), b()), xi[:r])
    assert quad_tuples(xx) == xref.as_tuple()
    with pytest.raises(P.DimensionMismatch):
        P.dot(residual, x)                                                 # lengths 3 and 4


# ------------------------------------------------------------------ test/lazyexpression.jl:138-185, :188-245 (mul!, scale!, vcat!)
def test_expression_scale_mul_vcat_chained_matvec():
    model = P.mock_model()
    n = 3
    x = [Variable(model) for _ in range(n)]
    xi = [v.index for v in x]
    rng = np.random.default_rng(3)
    A = P.Parameter(model, val=rng.random((2, n)))
    B = P.Parameter(model, val=rng.random((4, 2)))
    b = P.Parameter(model, val=rng.random(2))
    s = P.Parameter(lambda: 2.5, model)
    r = A * x + b
    ref_r = O.AffVec(2).vecadd(O.AffVec(2).matvecmul_vars(A(), xi), b())
    assert affvec_tuples((s * r)()) == O.AffVec(2).scale_number_affs(2.5, ref_r).as_tuples()
    assert affvec_tuples((r * s)()) == O.AffVec(2).scale_number_affs(2.5, ref_r).as_tuples()
    Br = (B * r)()                                                         # matvecmul!(y, B, ::Vector{AffineFunction})
    assert affvec_tuples(Br) == O.AffVec(4).matvecmul_affs(B(), ref_r).as_tuples()
    l = P.Parameter(model, val=rng.random(n))
    bnd = x - l                                                            # Vector{Variable} - numbers
    ref_b = O.AffVec(n).vecsubtract(xi, l())
    assert affvec_tuples(bnd()) == ref_b.as_tuples()
    stacked = P.vcat(r, bnd, r)()                                          # ragged rows (3 terms, 1 term)
    assert affvec_tuples(stacked) == O.AffVec(2 + n + 2).vcat(ref_r, ref_b, ref_r).as_tuples()
    d = (r - (A * x))()                                                    # Vector{AffineFunction} - Vector{AffineFunction}
    assert affvec_tuples(d) == O.AffVec(2).vecsubtract(ref_r, O.AffVec(2).matvecmul_vars(A(), xi)).as_tuples()
    sx = (s * x)()                                                         # scale!(dest::Vector{LinearTerm}, x, y)
    assert [(t.coeff, t.var.index) for t in sx] == [(2.5, i) for i in xi]


# ------------------------------------------------------------------ test/model.jl:26-81 (unconstrained, bilinear objective, scalar Parameter)
def test_model_unconstrained_bilinear():
    opt = DenseQPOptimizer(variable_offset=3)
    model = P.Model(opt)
    n = 2
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(1)
    sval = [1.0]
    def updQ(Q):
        Q[0, 0] = rng.random(); Q[1, 1] = rng.random()
    Q = P.Parameter(updQ, np.eye(n), model)
    r = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(n)), np.zeros(n), model)
    s = P.Parameter(lambda: sval[0], model)
    P.objective(model, P.Minimize, P.transpose(x) * Q * x + P.dot(r, x) + s)
    P.initialize(model)
    assert model.initialized
    for trial in range(4):
        if trial == 2:
            sval[0] = 2.0                                                   # constant modification (test/model.jl:76-80)
        P.solve(model)
        assert P.terminationstatus(model) == "OPTIMAL"
        xval = P.value(model, x)
        expected = -np.linalg.solve(2 * Q(), r())                           # -2 * Q() \ r()
        np.testing.assert_allclose(xval, expected, atol=1e-8)
        assert P.objectivevalue(model) == pytest.approx(xval @ Q() @ xval + r() @ xval + s(), abs=1e-8)
    before = model.device().bytes_allocated()
    P.solve(model)
    assert model.device().bytes_allocated() == before


# ------------------------------------------------------------------ test/model.jl:83-125 = BASELINE config 1 (README Example 1)
@pytest.mark.parametrize("mode", ["literal", "canonical"])
def test_model_equality_constrained_readme_example_1(mode):
    n, m = 8, 2
    opt = DenseQPOptimizer(variable_offset=0, permute_seed=7)
    model = P.Model(opt, quadratic_mode=mode)
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(1234)
    def randrng(a):
        a[...] = rng.random(a.shape)
    A = P.Parameter(randrng, np.zeros((n, n)), model)
    b = P.Parameter(randrng, np.zeros(n), model)
    Cm = P.Parameter(randrng, np.zeros((m, n)), model)
    d = P.Parameter(randrng, np.zeros(m), model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x == d)
    with pytest.raises(P.ArgumentError):
        P.constraint(model, Cm * x, "≈", d)
    alloc = None
    for i in range(20):
        P.solve(model)
        assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
        if i > 0:
            assert model.device().bytes_allocated() == alloc                  # allocs == 0 after the first solve!
        alloc = model.device().bytes_allocated()
        Cp = np.linalg.pinv(Cm())
        Pn = np.eye(n) - Cp @ Cm()
        expected = Pn @ (np.linalg.pinv(A() @ Pn) @ (b() - A() @ Cp @ d())) + Cp @ d()
        np.testing.assert_allclose(P.value(model, x), expected, rtol=1e-4)
    # what the optimizer received is exactly the reference's MOI functions for these parameter values
    vm = model.model_var_to_optimizer
    w = O.LsqWorkspace(n, n, m)
    xi = np.arange(1, n + 1, dtype=np.int64)
    w.eval_objective(np.asfortranarray(A()).reshape(-1, order="F"), b(), xi)
    w.eval_constraint(np.asfortranarray(Cm()).reshape(-1, order="F"), d(), xi)
    f = model.objective.f
    if mode == "literal":
        at, qt, const = w.objective.moi(vm)
        assert np.array_equal(f.quadratic_terms.view(np.int64), qt.view(np.int64))
        assert np.array_equal(f.affine_terms.view(np.int64), at.view(np.int64))
        assert f.constant == const
    else:
        w.objective.canonicalize()
        at, qt, const = w.objective.moi(vm)
        assert np.array_equal(f.quadratic_terms["row"], qt["row"]) and np.array_equal(f.quadratic_terms["col"], qt["col"])
        np.testing.assert_allclose(f.quadratic_terms["coeff"], qt["coeff"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(f.affine_terms["coeff"], at["coeff"], rtol=1e-12, atol=0)
        assert np.array_equal(f.affine_terms["var"], at["var"]) and f.constant == const
    ct, cc = w.constraint.moi(vm)
    cf = list(model.constraints)[0].f
    assert np.array_equal(cf.terms.view(np.int64), ct.view(np.int64)) and np.array_equal(cf.constants, cc)


# ------------------------------------------------------------------ test/model.jl:127-171 (box constrained: bounds path, derived number p ⋅ p)
def test_model_box_constrained():
    n = 10
    model = P.Model(DenseQPOptimizer())
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(1234)
    l = P.Parameter(lambda v: v.__setitem__(slice(None), -rng.random(n)), np.zeros(n), model)
    u = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(n)), np.zeros(n), model)
    def updp(p):
        p[:] = 2 * l() if rng.random() < 0.5 else 2 * u()
    p = P.Parameter(updp, np.zeros(n), model)
    residual = x - p
    P.objective(model, P.Minimize, P.dot(residual, residual) - P.dot(p, p))
    P.constraint(model, x >= l) if False else P.constraint(model, x, ">=", l)
    P.constraint(model, x, "<=", u)
    for _ in range(10):
        P.solve(model)
        np.testing.assert_allclose(P.value(model, x), p() / 2, rtol=1e-4, atol=1e-9)
    specs = [c.spec for c in model.constraints]
    assert specs == ["vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives"]
    c0 = list(model.constraints)[0].f
    assert c0.terms.tolist() == [(i + 1, 1.0, i + 1) for i in range(n)]
    assert np.array_equal(c0.constants, 0.0 - l())


# ------------------------------------------------------------------ test/model.jl:340-362 (README Example 2: val= Parameters, X' * g - p)
def test_model_readme_example_2_adjoint_and_manual_parameters():
    rng = np.random.default_rng(1)
    n, m = 5, 15
    Xdata = rng.standard_normal((n, m))
    pdata = np.zeros(m)
    model = P.Model(DenseQPOptimizer(), quadratic_mode="literal")
    X = P.Parameter(model, val=Xdata)
    p = P.Parameter(model, val=pdata)
    g = [Variable(model) for _ in range(n)]
    resid = X.T * g - p
    P.objective(model, P.Minimize, resid.T * resid)
    for _ in range(2):
        ggt = rng.standard_normal(n)
        pdata[:] = Xdata.T @ ggt
        P.solve(model)
        np.testing.assert_allclose(P.value(model, g), ggt, rtol=0.01)
    got = resid()
    ref = O.AffVec(m).vecsubtract(O.AffVec(m).matvecmul_vars(Xdata.T, [v.index for v in g]), pdata)
    assert affvec_tuples(got) == ref.as_tuples()


# ------------------------------------------------------------------ device-resident Parameters: the synthetic BASELINE stream
def test_device_uniform_parameters_match_oracle_stream_and_advance():
    model = P.mock_model(quadratic_mode="canonical")
    n, r = 40, 24
    x = [Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r, n), 1, model)
    b = P.DeviceUniformParameter((r,), 2, model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    for epoch in range(2):
        P.solve(model)
        Ah = O.fill_uniform(r * n, 1 + 1000 * A.epoch).reshape(n, r).T
        bh = O.fill_uniform(r, 2 + 1000 * b.epoch)
        assert np.array_equal(A(), Ah) and np.array_equal(b(), bh)
        f = model.objective.f
        iu = np.triu_indices(n)
        np.testing.assert_allclose(f.quadratic_terms["coeff"], (2 * Ah.T @ Ah)[iu], rtol=1e-12)
        assert np.array_equal(f.quadratic_terms["row"], iu[0] + 1) and np.array_equal(f.quadratic_terms["col"], iu[1] + 1)
        np.testing.assert_allclose(f.affine_terms["coeff"], -2 * Ah.T @ bh, rtol=1e-12)
    assert A.epoch >= 1
    assert model.objective.mode == "canonical"


def test_hipgraph_replay_gives_identical_buffers():
    outs = []
    for use_graph in (False, True):
        model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", use_graph=use_graph)
        n, r, m = 33, 20, 4
        x = [Variable(model) for _ in range(n)]
        A = P.DeviceUniformParameter((r, n), 11, model, advance=False)
        b = P.DeviceUniformParameter((r,), 12, model, advance=False)
        Cm = P.DeviceUniformParameter((m, n), 13, model, advance=False)
        d = P.DeviceUniformParameter((m,), 14, model, scale=2.0, advance=False)
        res = A * x - b
        P.objective(model, P.Minimize, P.dot(res, res))
        P.constraint(model, Cm * x == d)
        P.solve(model); P.solve(model)
        f, c = model.objective.f, list(model.constraints)[0].f
        outs.append((f.quadratic_terms.copy(), f.affine_terms.copy(), f.constant, c.terms.copy(), c.constants.copy()))
        model.close()
    for a, b_ in zip(outs[0], outs[1]):
        assert np.array_equal(np.asarray(a).view(np.int64) if hasattr(a, "dtype") and a.dtype.fields else np.asarray(a),
                              np.asarray(b_).view(np.int64) if hasattr(b_, "dtype") and b_.dtype.fields else np.asarray(b_))


# ------------------------------------------------------------------ BASELINE config 3 at full size (n = 4096): <= constraints, bounds, val= Parameters
def test_config3_full_size_inequalities_bounds_and_manual_parameters():
    n, r, mi = 4096, 4096, 512
    rng = np.random.default_rng(5)
    model = P.Model(P.MockOptimizer(variable_offset=100), quadratic_mode="canonical")
    x = [Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r, n), 1, model, advance=False)
    b = P.DeviceUniformParameter((r,), 2, model, advance=False)
    Gd, hd = np.asfortranarray(rng.random((mi, n))), rng.random(mi)
    ld, ud = -rng.random(n), rng.random(n)
    G, h = P.Parameter(model, val=Gd), P.Parameter(model, val=hd)       # manually updated work buffers (src/parameter.jl:88)
    l, u = P.Parameter(model, val=ld), P.Parameter(model, val=ud)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, G * x, "<=", h)
    P.constraint(model, x, ">=", l)
    P.constraint(model, x, "<=", u)
    P.solve(model)
    hd[:] = rng.random(mi); ud[:] = rng.random(n)                        # user overwrites the buffers between solves
    P.solve(model)
    cons = list(model.constraints)
    assert [c.spec for c in cons] == ["vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives", "vectoraffinefunction_in_nonpositives"]
    lo, gc, up = cons[0].f, cons[1].f, cons[2].f
    vm = model.model_var_to_optimizer
    assert np.array_equal(vm, np.arange(1, n + 1) + 100)
    assert np.array_equal(gc.terms["coeff"].reshape(mi, n), Gd) and np.array_equal(gc.constants, 0.0 - hd)
    assert np.array_equal(gc.terms["out"].reshape(mi, n), np.repeat(np.arange(1, mi + 1), n).reshape(mi, n))
    assert np.array_equal(gc.terms["var"].reshape(mi, n)[7], vm)
    assert np.array_equal(lo.terms["coeff"], np.ones(n)) and np.array_equal(lo.terms["var"], vm) and np.array_equal(lo.constants, 0.0 - ld)
    assert np.array_equal(up.constants, 0.0 - ud) and np.array_equal(up.terms["out"], np.arange(1, n + 1))
    # ... and byte for byte against the ORACLE at full size: the reference's matvecmul! + vecsubtract! (src/functions.jl:775-798,
    # 751-764) / `Variable - Number` rows (test/model.jl:162-163) followed by update!(::MOI.VectorAffineFunction) (src/moi_interop.jl:64-81)
    from oracle import oracle as O
    xi = np.arange(1, n + 1, dtype=np.int64)                              # Variable.index of x (1-based, creation order)
    for got, want in ((gc, O.AffVec(mi).vecsubtract(O.AffVec(mi).matvecmul_vars(Gd, xi), hd)),
                      (lo, O.AffVec(n).vecsubtract(xi, ld)), (up, O.AffVec(n).vecsubtract(xi, ud))):
        wt, wc = want.moi(vm)
        assert got.terms.dtype == wt.dtype and np.array_equal(got.terms.view(np.int64), wt.view(np.int64))
        assert np.array_equal(got.constants.view(np.int64), wc.view(np.int64))
    f = model.objective.f
    assert model.objective.mode == "canonical" and len(f.quadratic_terms) == n * (n + 1) // 2
    Ah = A()
    iu = np.triu_indices(n)
    rows = rng.integers(0, len(iu[0]), 2000)
    want = 2 * np.einsum("ij,ij->j", Ah[:, iu[0][rows]], Ah[:, iu[1][rows]])
    np.testing.assert_allclose(f.quadratic_terms["coeff"][rows], want, rtol=1e-12)
    assert np.array_equal(f.quadratic_terms["row"][rows], vm[iu[0][rows]]) and np.array_equal(f.quadratic_terms["col"][rows], vm[iu[1][rows]])


# ------------------------------------------------------------------ test/model.jl:364-395 (quadratic constraint functions; build only — the
# reference needs Gurobi to SOLVE this model, the hot path is the function assembly)
def test_quadratic_constraint_model_functions():
    rng = np.random.default_rng(1)
    opt = P.MockOptimizer(variable_offset=4)
    model = P.Model(opt)

    def newdir(v):
        v[:] = rng.standard_normal(2)
        v /= np.linalg.norm(v)
    direction = P.Parameter(newdir, np.zeros(2), model)
    zmax = P.Parameter(lambda: float(rng.random()), model)
    x, y, z = Variable(model), Variable(model), Variable(model)
    mu = 0.7
    P.constraint(model, x ** 2 + y ** 2 <= mu ** 2 * z ** 2)
    P.constraint(model, z >= 0)
    P.constraint(model, z <= zmax)
    P.objective(model, P.Maximize, P.dot(direction, [x, y]))
    for _ in range(5):
        P.solve(model)
        f = model.objective.f                                               # direction . [x, y] with optimizer indices 5, 6
        assert f.terms.tolist() == [(direction()[0], 5), (direction()[1], 6)] and f.constant == 0.0
        cons = {c.spec: c for c in model.constraints}
        assert sorted(cons) == ["scalaraffinefunction_in_greaterthan", "scalaraffinefunction_in_lessthan", "scalarquadraticfunction_in_lessthan"]
        zc = cons["scalaraffinefunction_in_lessthan"].f                     # z - zmax <= 0
        assert zc.terms.tolist() == [(1.0, 7)] and zc.constant == 0.0 - zmax()
        qc = cons["scalarquadraticfunction_in_lessthan"].f                  # x^2 + y^2 - mu^2 z^2 <= 0, constant: model indices, set once
        assert qc.quadratic_terms.tolist() == [(2.0, 1, 1), (2.0, 2, 2), (2 * (0.0 - mu ** 2), 3, 3)] and qc.affine_terms.tolist() == []
    assert opt.sense == P.Maximize and opt.optimize_calls == 5
