"""Pins the CPU oracle against every known-answer test the reference holds for the hot path.

Each test names the reference test it transcribes (paths relative to /root/reference).
The reference tests use Julia operator syntax to build the expected values; the expected
term lists below are those values written out term by term (the derivation is given
in a comment where it is not obvious).  Variables are 1-based indices.
"""
import numpy as np
import pytest

from oracle import oracle as O
from moi_dense import dense_quadratic, dense_vector_affine, solve_eq_qp

X, Y = 1, 2


# ------------------------------------------------------------------ test/util.jl:13-16
def _sort_and_combine(pairs):
    f = O.Aff([(v, k) for k, v in pairs])
    f.canonicalize()
    return [(v, c) for c, v in f.as_tuple()[0]]


def test_sort_and_combine_known_answers():
    assert _sort_and_combine([(2, 1.0), (2, 2.0)]) == [(2, 3.0)]
    assert _sort_and_combine([(3, 1.0), (2, 4.0), (3, 2.0), (1, 2.0)]) == [(1, 2.0), (2, 4.0), (3, 3.0)]
    n = 100
    rng = np.random.default_rng(0)
    assert len(_sort_and_combine([(i + 1, rng.random()) for i in range(n)])) == n
    assert len(_sort_and_combine([(3, rng.random()) for _ in range(n)])) == 1


def test_sort_and_combine_random_matches_dict_sum():
    # test/util.jl:18 builds v = [rand(1:n/2) => rand()]; the sum per key is order-dependent
    # only in the last bits
    rng = np.random.default_rng(1)
    pairs = [(int(rng.integers(1, 51)), float(rng.random())) for _ in range(1000)]
    out = _sort_and_combine(pairs)
    keys = [k for k, _ in out]
    assert keys == sorted(set(k for k, _ in pairs))
    for k, s in out:
        assert s == pytest.approx(sum(v for kk, v in pairs if kk == k), rel=1e-13)


# ------------------------------------------------------------------ test/functions.jl:9-28
def test_canonicalize_quadratic_term():
    # canonicalize(3*x*y) === canonicalize(3*y*x) == QuadraticTerm(3, x, y); through combine
    q = O.Quad(quad=[(3.0, Y, X), (0.0, X, Y)])
    q.canonicalize()
    assert q.as_tuple()[0] == [(3.0, X, Y)]


def test_canonicalize_affine():
    # y + x - 2*y + 3  ==  1*x2 + 1*x1 + -2*x2 + 3   (docstring src/functions.jl:283-289)
    f = O.Aff([(1.0, Y), (1.0, X), (-2.0, Y)], 3.0)
    f.canonicalize()
    assert f.as_tuple() == ([(1.0, X), (-1.0, Y)], 3.0)          # == x - y + 3


def test_canonicalize_quadratic_function():
    # x*y + y*x + y + y + x - y + 4 (docstring src/functions.jl:398-404)
    f = O.Quad(quad=[(1.0, X, Y), (1.0, Y, X)], linear=[(1.0, Y), (1.0, Y), (1.0, X), (-1.0, Y)], constant=4.0)
    f.canonicalize()
    assert f.as_tuple() == ([(2.0, X, Y)], [(1.0, X), (1.0, Y)], 4.0)   # == 2*x*y + x + y + 4


def test_canonicalize_keeps_zeros_prune_drops_them():
    f = O.Aff([(0.0, Y), (1.0, X)], 1.0)                           # 0*y + x + 1
    f.canonicalize()
    assert f.as_tuple() == ([(1.0, X), (0.0, Y)], 1.0)            # == x + 0*y + 1
    f.prune_zero()
    assert f.as_tuple() == ([(1.0, X)], 1.0)                      # == x + 1
    # 0*y^2 + x^2 + 0*x + y + 1
    g = O.Quad(quad=[(0.0, Y, Y), (1.0, X, X)], linear=[(0.0, X), (1.0, Y)], constant=1.0)
    g.canonicalize()
    assert g.as_tuple() == ([(1.0, X, X), (0.0, Y, Y)], [(0.0, X), (1.0, Y)], 1.0)
    g.prune_zero()
    assert g.as_tuple() == ([(1.0, X, X)], [(1.0, Y)], 1.0)       # == x^2 + y + 1


def test_svector_dot_canonical():
    # test/functions.jl:272-276: canonicalize(dot([x,y],[x,y])) == x^2 + y^2
    q = O.Quad().vecdot_vars_vars([X, Y], [X, Y]).canonicalize()
    assert q.as_tuple() == ([(1.0, X, X), (1.0, Y, Y)], [], 0.0)


# ------------------------------------------------------------------ test/functions.jl:148-163
def test_matvecmul_known_answer():
    A = np.ones((3, 4))
    x = [1, 2, 3, 4]
    y = O.AffVec(3).matvecmul_vars(A, x)
    expected = ([(1.0, 1), (1.0, 2), (1.0, 3), (1.0, 4)], 0.0)     # sum(x)
    assert y.as_tuples() == [expected] * 3
    # second call reuses dest (0 allocations in the reference) and gives the same answer
    y.matvecmul_vars(A, x)
    assert y.as_tuples() == [expected] * 3


def test_matvecmul_affine_known_answer():
    # Bx = rand(4,4) * x ; matvecmul!(y2, ones(3,4), Bx) == fill(sum(Bx), 3)
    rng = np.random.default_rng(3)
    B = rng.random((4, 4))
    x = [1, 2, 3, 4]
    Bx = O.AffVec(4).matvecmul_vars(B, x)
    y2 = O.AffVec(3).matvecmul_affs(np.ones((3, 4)), Bx)
    terms = []
    for i in range(4):                                            # sum(Bx): add!(copy, next) appends
        terms += Bx[i].as_tuple()[0]
    assert y2.as_tuples() == [(terms, 0.0)] * 3


def test_matvecmul_dimension_mismatch():
    with pytest.raises(O.DimensionMismatch):
        O.AffVec(2).matvecmul_vars(np.ones((3, 4)), [1, 2, 3, 4])     # src/functions.jl:780
    with pytest.raises(O.DimensionMismatch):
        O.AffVec(3).matvecmul_vars(np.ones((3, 4)), [1, 2, 3])        # src/functions.jl:781


# ------------------------------------------------------------------ test/functions.jl:165-199
def _aff123():
    return O.Aff([(1.0, 1), (2.0, 2), (3.0, 3)], 4.0)               # [1,2,3]' * x + 4


def test_mul_affine():
    aff = _aff123()
    dest = O.Aff()
    for _ in range(2):
        dest.mul_aff_number(aff, 2.0)
        assert dest.as_tuple() == ([(2.0, 1), (4.0, 2), (6.0, 3)], 8.0)
        dest.mul_aff_number(aff, 3.0)
        assert dest.as_tuple() == ([(3.0, 1), (6.0, 2), (9.0, 3)], 12.0)


def test_mul_quadratic():
    aff = _aff123()
    quad = O.Quad(quad=[(1.0, 1, 1), (2.0, 1, 3)], linear=[(3.0, 2)], constant=4.0)   # x1^2 + 2*x1*x3 + 3*x2 + 4
    dest = O.Quad()
    for _ in range(2):
        dest.mul_quad_number(quad, 2.0)
        assert dest.as_tuple() == ([(2.0, 1, 1), (4.0, 1, 3)], [(6.0, 2)], 8.0)
        # mul!(dest, aff, x[1]) == [1,2,3]' * (x .* x[1]) + 4 * x[1]
        dest.mul_aff_var(aff, 1)
        assert dest.as_tuple() == ([(1.0, 1, 1), (2.0, 2, 1), (3.0, 3, 1)], [(4.0, 1)], 0.0)
        # mul!(dest, aff, aff) == ([1,2,3]' * x + 4)^2
        dest.mul_aff_aff(aff, aff)
        q, lin, const = dest.as_tuple()
        assert q == [(float(a * b), i, j) for i, a in ((1, 1), (2, 2), (3, 3)) for j, b in ((1, 1), (2, 2), (3, 3))]
        assert lin == [(4.0, 1), (8.0, 2), (12.0, 3), (4.0, 1), (8.0, 2), (12.0, 3)]
        assert const == 16.0
        for vals in ([1.0, 2.0, 3.0], [-0.5, 0.25, 2.0]):
            assert dest.eval(vals) == pytest.approx((vals[0] + 2 * vals[1] + 3 * vals[2] + 4) ** 2, rel=1e-15)


# ------------------------------------------------------------------ test/functions.jl:201-206
def test_dot_with_matrix_variables_is_column_major():
    # x = [V1 V3; V2 V4], w = [0.1 0.2; 0.3 0.4]:  x . w == 0.1*x[1] + 0.3*x[2] + 0.2*x[3] + 0.4*x[4] + 0.0
    w = np.array([[0.1, 0.2], [0.3, 0.4]])
    f = O.vecdot_aff_numbers_vars(w.T.reshape(-1), [1, 2, 3, 4])
    assert f.as_tuple() == ([(0.1, 1), (0.3, 2), (0.2, 3), (0.4, 4)], 0.0)


# ------------------------------------------------------------------ test/functions.jl:208-233
def test_matrix_operations_values():
    A1 = np.array([[1.0, 2.0], [3.0, 4.0]])
    x = [1, 2]
    vals = [2.0, 5.0]
    fs = O.AffVec(2).matvecmul_vars(A1, x)
    assert [fs[i].eval(vals) for i in range(2)] == [12.0, 26.0]    # A1 * xvals
    a = [1.0, 2.0]
    f = O.vecdot_aff_numbers_vars(a, x)                            # a . x == a[1]*x[1] + a[2]*x[2]
    assert f.as_tuple() == ([(1.0, 1), (2.0, 2)], 0.0)
    gs = O.AffVec(2).vecadd(fs, a)                                 # gs = fs .+ a
    gvals = [gs[i].eval(vals) for i in range(2)]
    assert gvals == [13.0, 28.0]
    assert O.Quad().vecdot_affs_affs(gs, gs).eval(vals) == 953.0  # (gs . gs)(vals) == gvals . gvals
    assert O.Quad().vecdot_affs_vars(gs, x).eval(vals) == 13.0 * 2 + 28.0 * 5
    h = O.Quad().vecdot_vars_vars(x, x)                            # show(h) == "1 * x1 * x1 + 1 * x2 * x2 + 0"
    assert h.as_tuple() == ([(1.0, 1, 1), (1.0, 2, 2)], [], 0.0)


# ------------------------------------------------------------------ test/functions.jl:101-146
def test_quadratic_function_evaluations():
    x = [1, 2]
    vals = np.array([1.0, 2.0])
    a = np.array([4.0, 5.0])
    b = np.array([6.0, 7.0])
    V = [(1.0, 1), (1.0, 2)]
    ax = [(4.0, 1), (5.0, 2)]                                      # a .* x
    y = O.Quad()
    assert y.vecdot_terms_terms(V, ax).eval(vals) == vals @ (a * vals)
    assert y.as_tuple()[0] == [(4.0, 1, 1), (5.0, 2, 2)]
    assert y.vecdot_terms_terms(ax, ax).eval(vals) == (a * vals) @ (a * vals)
    axb = O.AffVec(2)                                              # a .* x .+ b
    bxa = O.AffVec(2)                                              # b .* x .+ a
    for i in range(2):
        axb[i].push(a[i], x[i]).set_constant(b[i])
        bxa[i].push(b[i], x[i]).set_constant(a[i])
    assert y.vecdot_affs_vars(axb, x).eval(vals) == vals @ (a * vals + b)
    assert y.vecdot_affs_affs(bxa, axb).eval(vals) == (b * vals + a) @ (a * vals + b)


# ------------------------------------------------------------------ test/functions.jl:235-253
def _vcat_inputs():
    f1 = ([(3.0, X)], 10.0)            # 3x + 10
    f2 = ([(0.1, Y)], -0.5)            # 0.1*y - 0.5
    f3 = ([(1.0, X), (1.0, Y)], 0.0)   # x + y
    def vec(fs):
        v = O.AffVec(len(fs))
        for i, (t, c) in enumerate(fs):
            for coeff, var in t:
                v[i].push(coeff, var)
            v[i].set_constant(c)
        return v
    return vec([f1, f2]), vec([f2, f3]), vec([f3, f2, f1])


def test_vcat_known_answers():
    v1, v2, v3 = _vcat_inputs()
    assert O.AffVec(2).vcat(v1).as_tuples() == v1.as_tuples()
    assert O.AffVec(4).vcat(v1, v2).as_tuples() == v1.as_tuples() + v2.as_tuples()
    assert O.AffVec(7).vcat(v1, v2, v3).as_tuples() == v1.as_tuples() + v2.as_tuples() + v3.as_tuples()


@pytest.mark.parametrize("n,which", [(1, 1), (3, 1), (3, 2), (5, 2), (6, 3), (8, 3)])
def test_vcat_dimension_mismatch(n, which):
    vs = _vcat_inputs()[:which]
    with pytest.raises(O.DimensionMismatch):
        O.AffVec(n).vcat(*vs)


# ------------------------------------------------------------------ builders not covered by a dedicated reference test
def test_vecsubtract_forms():
    A = np.array([[1.0, 2.0], [3.0, 4.0]])
    Ax = O.AffVec(2).matvecmul_vars(A, [1, 2])
    r = O.AffVec(2).vecsubtract(Ax, [0.5, -0.0])                  # src/functions.jl:751-764, :502, :474
    assert r.as_tuples() == [([(1.0, 1), (2.0, 2)], -0.5), ([(3.0, 1), (4.0, 2)], 0.0)]
    assert np.signbit(r[1].constant) == False                     # 0.0 - (-0.0) == +0.0
    bnd = O.AffVec(2).vecsubtract([1, 2], [0.25, 0.75])           # x - l  (test/model.jl:162)
    assert bnd.as_tuples() == [([(1.0, 1)], -0.25), ([(1.0, 2)], -0.75)]
    s = O.AffVec(2).vecsubtract(Ax, r)                            # aff - aff appends negated terms (:477-485)
    assert s.as_tuples()[0] == ([(1.0, 1), (2.0, 2), (-1.0, 1), (-2.0, 2)], 0.5)
    with pytest.raises(O.DimensionMismatch):
        O.AffVec(2).vecsubtract(Ax, [1.0, 2.0, 3.0])


def test_bilinearmul_pairs_column_major_coeff_with_row_major_vars():
    # src/functions.jl:849-856: term k = (Q[k] column-major linear index, x[row], y[col]) row-major
    Q = np.array([[1.0, 2.0], [3.0, 4.0]])
    q = O.Quad().bilinearmul(Q, [1, 2], [1, 2])
    assert q.as_tuple()[0] == [(1.0, 1, 1), (3.0, 1, 2), (2.0, 2, 1), (4.0, 2, 2)]
    vals = np.array([0.3, -1.2])
    assert q.eval(vals) == pytest.approx(vals @ Q @ vals, rel=1e-15)   # same quadratic form


def test_scale():
    assert O.scale_number_vars(2.5, [1, 2]).tolist() == [(2.5, 1), (2.5, 2)]
    ys = O.AffVec(1)
    ys[0].push(2.0, 1).set_constant(3.0)
    assert O.AffVec(1).scale_number_affs(0.5, ys).as_tuples() == [([(1.0, 1)], 1.5)]


# ------------------------------------------------------------------ src/moi_interop.jl:35-81
def test_moi_copies_varmap_and_diagonal_doubling():
    varmap = [30, 10, 20]
    q = O.Quad(quad=[(1.5, 1, 1), (2.0, 1, 3), (0.5, 3, 1)], linear=[(3.0, 2)], constant=4.0)
    at, qt, const = q.moi(varmap)
    assert qt.tolist() == [(3.0, 30, 30), (2.0, 30, 20), (0.5, 20, 30)]
    assert at.tolist() == [(3.0, 10)] and const == 4.0
    at, qt, const = q.moi()                                        # IdentityVarMap
    assert qt.tolist() == [(3.0, 1, 1), (2.0, 1, 3), (0.5, 3, 1)]
    fs = O.AffVec(2).matvecmul_vars(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]), [1, 2, 3])
    fs = O.AffVec(2).vecsubtract(fs, [7.0, 8.0])
    terms, consts = fs.moi(varmap)
    assert terms.tolist() == [(1, 1.0, 30), (1, 2.0, 10), (1, 3.0, 20), (2, 4.0, 30), (2, 5.0, 10), (2, 6.0, 20)]
    assert consts.tolist() == [-7.0, -8.0]
    t, c = O.aff_moi(O.Aff([(1.0, 3), (2.0, 1)], 5.0), varmap)
    assert t.tolist() == [(1.0, 20), (2.0, 30)] and c == 5.0


# ------------------------------------------------------------------ test/model.jl closed forms (solver-free)
def _kkt_objective(q, nvars):
    q.canonicalize()
    at, qt, const = q.moi()
    return dense_quadratic(at, qt, const, nvars)


def test_model_contquadratic_13_over_7():
    # test/model.jl:281-300: min x^2 + x*y + y^2 + y*z + z^2  s.t. x+2y+3z >= 4, x+y >= 1
    # optimum 13/7 at (4/7, 3/7, 6/7); both constraints are active there.
    q = O.Quad(quad=[(1.0, 1, 1), (1.0, 1, 2), (1.0, 2, 2), (1.0, 2, 3), (1.0, 3, 3)])
    Q, a, c = _kkt_objective(q, 3)
    assert np.array_equal(Q, np.array([[2.0, 1, 0], [1, 2, 1], [0, 1, 2]]))     # pins the diagonal doubling
    x = solve_eq_qp(Q, a, np.array([[1.0, 2, 3], [1, 1, 0]]), np.array([4.0, 1.0]))
    assert x == pytest.approx([4 / 7, 3 / 7, 6 / 7], abs=1e-12)
    assert 0.5 * x @ Q @ x + a @ x + c == pytest.approx(13 / 7, abs=1e-12)


def test_model_contquadratic_2_875():
    # test/model.jl:302-321: min 2x^2 + y^2 + x*y + x + y + 1  s.t. x >= 0, -y <= 0, x + y == 1
    q = O.Quad(quad=[(2.0, 1, 1), (1.0, 2, 2), (1.0, 1, 2)], linear=[(1.0, 1), (1.0, 2)], constant=1.0)
    Q, a, c = _kkt_objective(q, 2)
    x = solve_eq_qp(Q, a, np.array([[1.0, 1.0]]), np.array([1.0]))
    assert x == pytest.approx([0.25, 0.75], abs=1e-12)
    assert 0.5 * x @ Q @ x + a @ x + c == pytest.approx(2.875, abs=1e-12)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_model_equality_constrained_lsq_closed_form(seed):
    # test/model.jl:83-125 (BASELINE config 1: n = 8, m = 2): ||Ax-b||^2 s.t. Cx = d, against the
    # closed form of :87-92.  The literal (uncombined) MOI output is used, as the reference hands it over.
    n, m = 8, 2
    rng = np.random.default_rng(seed)
    A, b, Cm, d = rng.random((n, n)), rng.random(n), rng.random((m, n)), rng.random(m)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    w = O.LsqWorkspace(n, n, m)
    Ac, Cc = np.ascontiguousarray(A.T).reshape(-1), np.ascontiguousarray(Cm.T).reshape(-1)
    w.eval_objective(Ac, b, xvar, twice=True)
    w.eval_constraint(Cc, d, xvar)
    at, qt, const = w.objective.moi()
    assert len(qt) == n * n * n and len(at) == 2 * n * n          # r*n^2 and 2*r*n uncombined terms
    Q, a, c = dense_quadratic(at, qt, const, n)
    assert Q == pytest.approx(2 * A.T @ A, rel=1e-13)
    assert a == pytest.approx(-2 * A.T @ b, rel=1e-13)
    assert c == pytest.approx(b @ b, rel=1e-14)
    terms, consts = w.constraint.moi()
    M, k = dense_vector_affine(terms, consts, n)
    assert np.array_equal(M, Cm) and np.array_equal(k, -d)
    x = solve_eq_qp(Q, a, M, -k)
    Cp = np.linalg.pinv(Cm)
    P = np.eye(n) - Cp @ Cm
    expected = P @ (np.linalg.pinv(A @ P) @ (b - A @ Cp @ d)) + Cp @ d
    assert x == pytest.approx(expected, rel=1e-4)


def test_lsq_literal_term_order_appendix_a2():
    # quadratic term ((i*n + j)*n + k) = A[i,j]*A[i,k] (x2 on j == k), vars (j,k)
    n, r = 3, 2
    rng = np.random.default_rng(5)
    A, b = rng.random((r, n)), rng.random(r)
    xvar = np.array([1, 2, 3], dtype=np.int64)
    w = O.LsqWorkspace(n, r, 1)
    w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
    at, qt, const = w.objective.moi()
    k = 0
    for i in range(r):
        for j in range(n):
            for kk in range(n):
                coeff = A[i, j] * A[i, kk]
                assert qt[k].tolist() == ((2 * coeff if j == kk else coeff), j + 1, kk + 1)
                k += 1
    nb = 0.0 - b
    exp_aff = []
    for i in range(r):
        exp_aff += [(nb[i] * A[i, j], j + 1) for j in range(n)] * 2
    assert at.tolist() == exp_aff
    acc = 0.0
    for i in range(r):
        acc += nb[i] * nb[i]
    assert const == acc


def test_canonicalize_of_literal_is_2AtA():
    n = r = 12
    A = O.fill_uniform(r * n, 1).reshape(n, r).T                    # column-major fill
    b = O.fill_uniform(r, 2)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    w = O.LsqWorkspace(n, r, 1)
    w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
    w.objective.canonicalize()
    at, qt, const = w.objective.moi()
    iu = [(j + 1, k + 1) for j in range(n) for k in range(j, n)]
    assert list(zip(qt["row"].tolist(), qt["col"].tolist())) == iu      # Appendix A.3 ordering
    G = 2 * A.T @ A
    assert qt["coeff"] == pytest.approx(np.array([G[j - 1, k - 1] for j, k in iu]), rel=1e-12)
    assert at["var"].tolist() == list(range(1, n + 1))
    assert at["coeff"] == pytest.approx(-2 * A.T @ b, rel=1e-12)


def test_fill_uniform_stream_is_fixed():
    # golden values of the counter-based input stream shared with the device fill kernel
    u = O.fill_uniform(4, 1)
    assert np.all((u >= 0) & (u < 1))
    assert np.array_equal(u, O.fill_uniform(8, 1)[:4])             # counter based: prefix-stable
    golden = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "fill_uniform_seed1.npy"))
    assert np.array_equal(O.fill_uniform(len(golden), 1), golden)


# ------------------------------------------------------------------ independent golden vectors (tests/golden/make_c1_golden.py)
def _c1_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_readme_example1.npz"))


def test_oracle_reproduces_the_independent_config1_golden_vectors():
    """BASELINE config 1 computed by a third implementation (plain Python written from SURVEY.md Appendix A): the C oracle must
    reproduce it — indices exactly, literal coefficients bit for bit, canonical coefficients within 1e-12."""
    g = _c1_golden()
    n, r, m = int(g["n"]), int(g["r"]), int(g["m"])
    vm = np.zeros(n, dtype=np.int64); vm[:] = g["varmap"]
    A = g["A"].reshape(n, r).T
    Cm = g["C"].reshape(n, m).T
    xi = np.arange(1, n + 1, dtype=np.int64)
    assert np.array_equal(g["A"], O.fill_uniform(r * n, 1)) and np.array_equal(g["d"], O.fill_uniform(m, 4, 2.0))
    res = O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(A, xi), g["b"])
    terms, _, consts = res.flat()
    assert np.array_equal(terms["coeff"].reshape(r, n), g["residual_coeff"]) and np.array_equal(terms["var"].reshape(r, n), g["residual_var"])
    assert np.array_equal(consts, g["residual_const"])
    obj = O.Quad().vecdot_affs_affs(res, res)
    at, qt, const = obj.moi(vm)
    assert np.array_equal(qt.view(np.int64), g["literal_quad"].view(np.int64))
    assert np.array_equal(at.view(np.int64), g["literal_aff"].view(np.int64)) and const == g["const"][0]
    obj.canonicalize()
    at, qt, const = obj.moi(vm)
    assert np.array_equal(qt["row"], g["canonical_quad"]["row"]) and np.array_equal(qt["col"], g["canonical_quad"]["col"])
    np.testing.assert_allclose(qt["coeff"], g["canonical_quad"]["coeff"], rtol=1e-12, atol=0)
    assert np.array_equal(at["var"], g["canonical_aff"]["var"])
    np.testing.assert_allclose(at["coeff"], g["canonical_aff"]["coeff"], rtol=1e-12, atol=0)
    ct, cc = O.AffVec(m).vecsubtract(O.AffVec(m).matvecmul_vars(Cm, xi), g["d"]).moi(vm)
    assert np.array_equal(ct.view(np.int64), g["constraint_terms"].view(np.int64)) and np.array_equal(cc, g["constraint_consts"])
    bt, bc = O.AffVec(n).vecsubtract(xi, g["lows"]).moi(vm)
    assert np.array_equal(bt.view(np.int64), g["bounds_terms"].view(np.int64)) and np.array_equal(bc, g["bounds_consts"])
    Q = g["Q"].reshape(n, n).T
    _, bq, _ = O.Quad().bilinearmul(Q, xi, xi).moi(vm)
    assert np.array_equal(bq.view(np.int64), g["bilinear_quad"].view(np.int64))


def test_sampled_canonical_entries_equal_the_full_canonicalize():
    """pmo_canonical_*_samples (used by the full-size GPU parity tests, where the literal objective cannot be materialised) against
    canonicalize!(literal) + the MOI copy of the same oracle at sizes where the literal form exists: same indices, coefficients to
    rounding (the summation ORDER is the only difference, and the reference's QuickSort does not pin one)."""
    import numpy as np
    from oracle import oracle as O
    for n, r in ((6, 5), (17, 33), (40, 12)):
        A = O.fill_uniform(r * n, 1)
        b = O.fill_uniform(r, 2)
        x = np.arange(1, n + 1, dtype=np.int64)
        w = O.LsqWorkspace(n, r, 1)
        w.eval_objective(A, b, x)
        w.objective.canonicalize()
        at, qt, _ = w.objective.moi()
        iu = np.triu_indices(n)
        assert np.array_equal(qt["row"], iu[0] + 1) and np.array_equal(qt["col"], iu[1] + 1)
        q, q_ld = O.canonical_quad_samples(A, r, r, n, iu[0] + 1, iu[1] + 1)
        np.testing.assert_allclose(q, qt["coeff"], rtol=1e-14, atol=0)
        np.testing.assert_allclose(q_ld, qt["coeff"], rtol=1e-14, atol=0)
        l, l_ld = O.canonical_lin_samples(A, r, r, n, b, -1, x)
        np.testing.assert_allclose(l, at["coeff"], rtol=1e-14, atol=0)
