"""CPU-only tests of the host-side mirror of the reference API: scalar algebra (test/functions.jl), Parameter caching
(test/parameter.jl), constant folding, macro error handling (test/lazyexpression.jl:21-23, test/model.jl:103) and
constant-only models on the mock optimizer.  Each test names the reference test it transcribes."""
import numpy as np
import pytest

import parametron_jl_amd as P
from parametron_jl_amd import (AffineFunction, LinearTerm, QuadraticFunction, QuadraticTerm, Variable, canonicalize, prune_zero)
from parametron_jl_amd import hostops, moi
from oracle import oracle as O


def V(i):
    return Variable(i)


# ------------------------------------------------------------------ test/functions.jl:30-60
def test_linear_term():
    x = V(1)
    assert LinearTerm(4.5, x) == 4.5 * x == x * 4.5
    assert +(2 * x) == 2 * x and -(2 * x) == -2 * x
    assert 2 * (3 * x) == x * 6 and (2 * x) * 3 == 6 * x
    assert repr(LinearTerm(4.5, x)) == "4.5 * x1"


def test_quadratic_term():
    x, y = V(1), V(2)
    assert x * y == QuadraticTerm(1, x, y)
    assert (2 * x) * y == QuadraticTerm(2, x, y) == x * (2 * y)
    assert 2 * (3 * x * y) == x * 6 * y
    assert (y * 2 * x) * 3 == 6 * y * x
    assert repr(QuadraticTerm(1, x, y)) == "1 * x1 * x2"


# ------------------------------------------------------------------ test/functions.jl:9-28
def test_canonicalize_prune_zero():
    x, y = V(1), V(2)
    assert (3 * x * y).canonicalize() == (3 * y * x).canonicalize() == QuadraticTerm(3, x, y)
    assert canonicalize(y + x - 2 * y + 3) == x - y + 3
    assert canonicalize(x * y + y * x + y + y + x - y + 4) == 2 * x * y + x + y + 4
    assert canonicalize(0 * y + x + 1) == x + 0 * y + 1
    assert prune_zero(canonicalize(0 * y + x + 1)) == x + 1
    assert canonicalize(0 * y ** 2 + x ** 2 + 0 * x + y + 1) == x ** 2 + 0 * y ** 2 + 0 * x + y + 1
    assert prune_zero(canonicalize(0 * y ** 2 + x ** 2 + 0 * x + y + 1)) == x ** 2 + y + 1
    assert repr(y + x - 2 * y + 3) == "1 * x2 + 1 * x1 + -2 * x2 + 3"          # docstring src/functions.jl:285-286


# ------------------------------------------------------------------ test/functions.jl:62-99
def test_affine_function_evaluation():
    x, y = V(1), V(2)
    vals1 = {x: 1.0, y: 2.0}
    f1 = 2 * x + 3 * y + 5
    assert f1(vals1) == 2.0 + 3.0 * 2.0 + 5
    assert repr(f1) == "2 * x1 + 3 * x2 + 5"
    vals2 = {x: 2, y: -1}
    f2 = 2.5 * x + 4 * y + 1
    assert f2(vals2) == 2.5 * 2 - 4 + 1
    f4 = f1 + f2
    assert f4(vals1) == f1(vals1) + f2(vals1)
    f5 = f1 + 4.0
    f6 = 1 + f5
    f7 = f6 - f4
    assert f7(vals1) == f6(vals1) - f4(vals1)
    f8 = f7 - LinearTerm(4, y)
    assert f8(vals1) == f7(vals1) - 4 * vals1[y]
    assert AffineFunction.of(x) == x + 0


# ------------------------------------------------------------------ test/functions.jl:101-146, :165-199
def test_quadratic_function_and_mul():
    x = [V(1), V(2), V(3)]
    vals = {x[0]: 1.0, x[1]: 2.0, x[2]: -1.5}
    f1 = x[0] ** 2 + 2 * x[0] * x[1] + 3 * x[1] + 4
    f2 = x[1] ** 2 - 2 * x[0] + 3 * x[1] - 1
    assert (f1 - f2)(vals) == f1(vals) - f2(vals)
    aff = hostops.vecdot(np.array([1.0, 2.0, 3.0]), x) + 4                       # [1,2,3]' * x + 4
    assert aff == 1 * x[0] + 2 * x[1] + 3 * x[2] + 4
    assert aff * 2 == 2 * x[0] + 4 * x[1] + 6 * x[2] + 8
    assert 3 * aff == 3 * x[0] + 6 * x[1] + 9 * x[2] + 12
    quad = x[0] ** 2 + 2 * x[0] * x[2] + 3 * x[1] + 4
    assert quad * 2 == 2 * x[0] ** 2 + 4 * x[0] * x[2] + 6 * x[1] + 8 == 2 * quad
    assert aff * x[0] == hostops.vecdot(np.array([1.0, 2.0, 3.0]), [xi * x[0] for xi in x]) + 4 * x[0]
    assert x[0] * aff == aff * x[0]
    sq = aff * aff
    assert sq == aff ** 2
    assert sq(vals) == (1.0 + 4.0 - 4.5 + 4) ** 2
    # the same expansion through the oracle (ordered equality like Julia's ==)
    oq = O.Quad().mul_aff_aff(O.Aff([(1.0, 1), (2.0, 2), (3.0, 3)], 4.0), O.Aff([(1.0, 1), (2.0, 2), (3.0, 3)], 4.0))
    q, l, c = sq.to_arrays()
    assert (q.tolist(), l.tolist(), c) == (oq.terms().tolist(), oq.affine.terms().tolist(), oq.affine.constant)


# ------------------------------------------------------------------ test/functions.jl:201-233 (constant folding path)
def test_host_array_ops_match_reference_known_answers():
    x = [V(1), V(2), V(3), V(4)]
    w = np.array([[0.1, 0.2], [0.3, 0.4]])
    assert hostops.vecdot(x, w) == hostops.vecdot(w, x) == 0.1 * x[0] + 0.3 * x[1] + 0.2 * x[2] + 0.4 * x[3] + 0.0
    x2 = x[:2]
    fs = hostops.matvecmul(np.array([[1.0, 2.0], [3.0, 4.0]]), x2)
    vals = {x[0]: 2.0, x[1]: 5.0}
    assert [f(vals) for f in fs] == [12.0, 26.0]
    gs = hostops.vecaddsub(fs, [1, 2], +1)
    assert hostops.vecdot(gs, gs)(vals) == 953.0
    assert repr(hostops.vecdot(x2, x2)) == "1 * x1 * x1 + 1 * x2 * x2 + 0"       # test/functions.jl:229-232
    A = np.ones((3, 4))
    y = hostops.matvecmul(A, x)
    assert all(f == x[0] + x[1] + x[2] + x[3] for f in y)                      # test/functions.jl:148-154
    with pytest.raises(P.DimensionMismatch):
        hostops.matvecmul(A, x[:3])
    Q = np.array([[1.0, 2.0], [3.0, 4.0]])
    assert hostops.bilinearmul(Q, x2, x2).to_arrays()[0].tolist() == O.Quad().bilinearmul(Q, [1, 2], [1, 2]).terms().tolist()


# ------------------------------------------------------------------ test/parameter.jl:10-39
def test_parameter_caching_semantics():
    model = P.mock_model()
    box = [1]
    p1 = P.Parameter(lambda: box[0], model)                                     # out-of-place
    assert p1() == 1
    box[0] = 2
    assert p1() == 1                                                            # cached until set dirty
    P.setdirty(p1)
    assert p1() == 2
    A = np.zeros((3, 4))
    calls = []
    def fill(a):
        calls.append(1)
        a[:] = len(calls)
    p2 = P.Parameter(fill, A, model)                                            # in-place
    assert p2() is A                                                            # p2() === A (test/parameter.jl:28)
    assert np.all(A == 1)
    p2()
    assert len(calls) == 1
    model.setdirty()
    assert np.all(p2() == 2)
    p3 = P.Parameter(model, val=A)                                              # identity: manually updated work buffer
    assert p3() is A
    assert model.params == [p1, p2, p3]


# ------------------------------------------------------------------ macro misuse (test/lazyexpression.jl:21-23, test/model.jl:103)
def test_argument_errors():
    model = P.mock_model()
    x = [Variable(model) for _ in range(2)]
    with pytest.raises(P.ArgumentError):
        P.constraint(model, x, "≈", [0.0, 0.0])                                  # Relation not recognized
    with pytest.raises(P.ArgumentError):
        P.constraint(model, x[0], "in", "whatever")
    with pytest.raises(P.ArgumentError):
        P.constraint(model, x)                                                   # Expected expression of the form `a relation b`
    with pytest.raises(P.ArgumentError):
        P.Model(P.MockOptimizer(), quadratic_mode="nonsense")


# ------------------------------------------------------------------ constant-only models never touch the device
def test_constant_model_on_mock_optimizer_scalar_constraints_2():
    # test/model.jl:281-300 structure: min x^2 + x*y + y^2 + y*z + z^2 s.t. x+2y+3z >= 4, x+y >= 1
    opt = P.MockOptimizer(variable_offset=10)
    model = P.Model(opt)
    x, y, z = (Variable(model) for _ in range(3))
    assert (x.index, y.index, z.index) == (1, 2, 3)
    P.objective(model, P.Minimize, x ** 2 + x * y + y ** 2 + y * z + z ** 2)
    P.constraint(model, x + 2 * y + 3 * z >= 4)
    P.constraint(model, x + y >= 1)
    P.solve(model)
    assert model.initialized and opt.optimize_calls == 1
    assert opt.set_calls == 0                                                    # isconstant: update! skips MOI.set (moi_interop.jl:132,169)
    f = model.objective.f
    assert f.quadratic_terms.tolist() == [(2.0, 1, 1), (1.0, 1, 2), (2.0, 2, 2), (1.0, 2, 3), (2.0, 3, 3)]   # diagonal doubled (:58)
    cons = list(model.constraints)
    assert [c.spec for c in cons] == ["scalaraffinefunction_in_greaterthan"] * 2
    assert cons[0].f.terms.tolist() == [(1.0, 1), (2.0, 2), (3.0, 3)] and cons[0].f.constant == -4.0
    assert model.value(x) == 0.0 and model.model_var_to_optimizer.tolist() == [11, 12, 13]
    with pytest.raises(P.ErrorException):
        Variable(model)                                                          # Model has already been initialized (model.jl:50)
    with pytest.raises(P.ErrorException):
        P.objective(model, P.Minimize, x)


def test_constant_vector_constraints_and_ordering():
    # test/model.jl:208-220 (MOI issue 426) + Issue 30: [x] <= [-3.]
    model = P.mock_model()
    x = Variable(model)
    P.constraint(model, [x], ">=", [0.0])
    P.constraint(model, [x], "<=", [1.0])
    P.constraint(model, [x], "==", [0.5])
    P.constraint(model, x, "in", "ℤ")
    specs = [c.spec for c in model.constraints]
    assert specs == ["vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives", "vectoraffinefunction_in_zeros",
                     "singlevariable_in_integer"]
    c = list(model.constraints)[1]
    assert c.f.terms.tolist() == [(1, 1.0, 1)] and c.f.constants.tolist() == [-1.0] and c.set == moi.Nonpositives(1)
    P.solve(model)                                                               # default objective (issue #62)
    assert model.objective.f.terms.tolist() == [] and model.objective.f.constant == 0.0


def test_device_expressions_fail_loudly_without_gpu():
    from parametron_jl_amd import _lib
    if _lib.load().pmt_device_count() > 0:
        pytest.skip("GPU present")
    model = P.mock_model()
    x = [Variable(model) for _ in range(2)]
    A = P.Parameter(model, val=np.eye(2))
    with pytest.raises(P.ErrorException):
        A * x                                                                    # needs HBM: no CPU fallback


def test_getfield_optimization_and_derived_parameters_track_their_sources():
    """test/lazyexpression.jl:364-381: `@expression p.x` and `@expression p.x + 1` follow the Parameter through setdirty!(p) alone
    (the reference re-evaluates its arguments on every call; a derived Parameter recomputes when a source is dirty or was updated)."""
    import random

    class MyWrapper:
        def __init__(self, x):
            self.x = x
    model = P.mock_model()
    p = P.Parameter(lambda: MyWrapper(random.random()), model)
    ex1 = P.getproperty(p, "x")
    assert ex1() == p().x
    P.setdirty(p)
    assert ex1() == p().x
    ex2 = ex1 + 1
    assert ex2() == p().x + 1
    P.setdirty(p)
    assert ex2() == p().x + 1 and ex1() == p().x
    before = p().x
    assert ex1() == before and ex2() == before + 1                         # nothing dirty: cached, the callback does not run again


def test_generic_functions_of_parameters_user_functions_hcat_getindex_reshape():
    """test/lazyexpression.jl:63-82 (user functions), :384-393 (hcat), :395-404 (getindex), :416-419 (tuple arguments)."""
    import numpy as np
    model = P.mock_model()
    rng = np.random.default_rng(0)
    p = P.Parameter(lambda: rng.random((3, 3)), model)
    hcat_expr = P.lazy(lambda a, b: np.hstack([a, b]), p, p)
    assert np.array_equal(hcat_expr(), np.hstack([p(), p()]))
    P.setdirty(p)
    assert np.array_equal(hcat_expr(), np.hstack([p(), p()]))
    col = P.getindex(p, slice(None), 1)
    assert np.array_equal(col(), p()[:, 1])
    P.setdirty(p)
    assert np.array_equal(col(), p()[:, 1])

    class SpatialMat:
        def __init__(self):
            self.angular, self.linear = np.zeros((3, 4)), np.zeros((3, 4))
    scalar = [1.0]
    mat = SpatialMat()

    def updatemat(m):
        m.angular[...] = scalar[0]
        m.linear[...] = scalar[0]
    pmat = P.Parameter(updatemat, mat, model)
    pmat_angular = P.lazy(lambda m: m.angular, pmat)
    result = pmat_angular()
    assert result is mat.angular and np.all(result == scalar[0])            # aliasing, not a copy (=== in the reference)
    scalar[0] = 2.0
    P.setdirty(model)
    assert np.all(pmat_angular() == 2.0)
    A = [1, 2, 3, 4]
    assert np.array_equal(P.lazy(np.reshape, A, (2, 2)), np.reshape(A, (2, 2)))   # no Parameter involved: evaluated on the spot


# ------------------------------------------------------------------ generic rule src/lazyexpression.jl:198 on plain numbers
def test_number_array_products_are_matrix_products():
    """Julia's `*` on number arrays is the matrix product; numpy's elementwise broadcast must never leak through
    (round-1 advisor finding: apply('*', A, B) returned A .* B)."""
    A = np.array([[1.0, 2.0], [3.0, 4.0]])
    B = np.array([[5.0, 6.0], [7.0, 8.0]])
    b = np.array([5.0, 6.0])
    assert np.array_equal(hostops.apply("*", A, B), np.array([[19.0, 22.0], [43.0, 50.0]]))
    assert np.array_equal(hostops.apply("*", A, b), np.array([17.0, 39.0]))
    assert np.array_equal(hostops.apply("*", 2.0, b), np.array([10.0, 12.0]))
    assert np.array_equal(hostops.apply("*", b, np.array([[1.0, 2.0, 3.0]])), np.outer(b, [1.0, 2.0, 3.0]))   # Vector * one-row Matrix
    with pytest.raises(P.ArgumentError):                       # Vector * Vector: MethodError in Julia
        hostops.apply("*", b, b)
    with pytest.raises(P.DimensionMismatch):
        hostops.apply("*", A, np.ones(3))
    with pytest.raises(P.DimensionMismatch):
        hostops.apply("*", A, np.ones((3, 2)))
    # x' * Q for numbers is the row vector (Q'x)'; times a vector it is the bilinear form
    row = hostops.apply("*", hostops.Transpose(b), A)
    assert isinstance(row, hostops.Transpose) and np.array_equal(row.parent, A.T @ b)
    assert hostops.apply("*", row, b) == float(b @ A @ b)


def test_scalar_times_number_array_of_any_rank_is_a_scaling():
    """`2 * A` is valid Julia (the generic rule src/lazyexpression.jl:198 calls `*` out of place); a 0-dimensional array is a
    scalar.  Only vector * vector has no method."""
    from parametron_jl_amd import hostops
    A = np.arange(6.0).reshape(2, 3)
    assert np.array_equal(hostops.apply("*", 2.0, A), 2.0 * A)
    assert np.array_equal(hostops.apply("*", A, 2.0), 2.0 * A)
    assert np.array_equal(hostops.apply("*", np.array(2.0), np.ones(3)), 2.0 * np.ones(3))
    assert np.array_equal(hostops.apply("*", np.ones(3), np.array(0.5)), 0.5 * np.ones(3))
    assert np.array_equal(hostops.apply("*", np.array(3.0), A), 3.0 * A)
    with pytest.raises(P.ArgumentError):
        hostops.apply("*", np.ones(3), np.ones(3))


def test_derived_parameter_scalar_times_matrix_parameter():
    """a Parameter-only expression scalar Parameter * matrix Parameter is a DerivedParameter that is recomputed out of place"""
    model = P.mock_model()
    s = P.Parameter(model, val=2.0)
    M = P.Parameter(model, val=np.eye(2))
    prod = P.lazy("*", s, M)
    assert np.array_equal(prod(), 2.0 * np.eye(2))
