"""-m gpu: the device-resident solver hand-off.  The CSC matrices built in HBM must equal what
MathOptInterface 0.8 + a solver wrapper would assemble on the host from the same MOI functions (tests/moi_dense.py restates
that interpretation), and a KKT solve on the device-built data must reproduce the reference's closed-form answers."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import Variable, moi  # noqa: E402
from parametron_jl_amd.handoff import DeviceQP  # noqa: E402
from moi_dense import dense_quadratic, dense_scalar_affine, dense_vector_affine, solve_eq_qp  # noqa: E402
from qp_solver import DenseQPOptimizer  # noqa: E402


def host_qp(model, nopt, infty):
    """P, q, r, A, l, u assembled on the host from the MOI functions the optimizer received."""
    f = model.objective.f
    vm = model.model_var_to_optimizer
    remap = (lambda a, fields: a) if not model.objective.isconstant else None
    if isinstance(f, moi.ScalarQuadraticFunction):
        Pm, q, r = dense_quadratic(f.affine_terms, f.quadratic_terms, f.constant, nopt)
    else:
        q, r = dense_scalar_affine(f.terms, f.constant, nopt)
        Pm = np.zeros((nopt, nopt))
    rows, lo, hi = [], [], []
    for c in model.constraints:
        if isinstance(c.f, moi.VectorAffineFunction):
            t = c.f.terms.copy()
            if c.isconstant:
                t["var"] = vm[t["var"] - 1]
            M, k = dense_vector_affine(t, c.f.constants, nopt)
            v = 0.0
        else:
            t = c.f.terms.copy()
            if c.isconstant:
                t["var"] = vm[t["var"] - 1]
            a, k0 = dense_scalar_affine(t, c.f.constant, nopt)
            M, k, v = a[None, :], np.array([k0]), c.set.value or 0.0
        b = v - k
        kind = {moi.EqualTo: 0, moi.Zeros: 0, moi.GreaterThan: 1, moi.Nonnegatives: 1, moi.LessThan: 2, moi.Nonpositives: 2}[type(c.set)]
        rows.append(M)
        lo.append(np.full(len(k), -infty) if kind == 2 else b)
        hi.append(np.full(len(k), infty) if kind == 1 else b)
    A = np.vstack(rows) if rows else np.zeros((0, nopt))
    return Pm, q, r, A, (np.concatenate(lo) if lo else np.zeros(0)), (np.concatenate(hi) if hi else np.zeros(0))


def dense_of(csc, shape):
    x, i, p = csc
    return sp.csc_matrix((x, i, p), shape=shape).toarray()


def check_csc_canonical(csc):
    x, i, p = csc
    assert p[0] == 0 and p[-1] == len(i) == len(x) and np.all(np.diff(p) >= 0)
    for c in range(len(p) - 1):
        col = i[p[c]:p[c + 1]]
        assert np.all(np.diff(col) > 0)                                    # strictly increasing rows: sorted, no duplicates


@pytest.mark.parametrize("mode", ["literal", "canonical"])
def test_handoff_matches_host_assembly_and_tracks_updates(mode):
    n, r, m, g = 9, 13, 3, 4
    opt = DenseQPOptimizer(variable_offset=0, permute_seed=11)
    model = P.Model(opt, quadratic_mode=mode)
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(5)
    fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape) - 0.3)
    A = P.Parameter(fill, np.zeros((r, n)), model)
    b = P.Parameter(fill, np.zeros(r), model)
    Cm = P.Parameter(fill, np.zeros((m, n)), model)
    d = P.Parameter(fill, np.zeros(m), model)
    G = P.Parameter(fill, np.zeros((g, n)), model)
    h = P.Parameter(fill, np.zeros(g), model)
    lo = P.Parameter(lambda v: v.__setitem__(Ellipsis, -1 - rng.random(n)), np.zeros(n), model)
    w = P.Parameter(fill, np.zeros(n), model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x == d)
    P.constraint(model, G * x <= h)
    P.constraint(model, x, ">=", lo)
    P.constraint(model, P.dot(w, x) <= 5.0)                                 # scalar affine row
    P.constraint(model, [1.0 * x[0] + 2.0 * x[3] - 1.0, 1.0 * x[2] + 0.5], "<=", [0.0, 0.0])   # constant vector block
    P.solve(model)
    qp = DeviceQP(model, infty=1e20)
    nopt = qp.nvars
    assert nopt == n and qp.nrows == m + g + n + 1 + 2
    for it in range(3):
        if it:
            P.solve(model)
            qp.refresh()
        got = qp.fetch()
        Pm, q, rr, Am, l, u = host_qp(model, nopt, 1e20)
        check_csc_canonical(got["P"]); check_csc_canonical(got["A"])
        Pd = dense_of(got["P"], (nopt, nopt))
        assert np.all(np.tril(Pd, -1) == 0)
        tol = dict(rtol=0, atol=0) if mode == "canonical" else dict(rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(Pd, np.triu(Pm), **tol)
        np.testing.assert_allclose(got["q"], q, rtol=1e-13, atol=1e-14)
        assert got["r"] == rr
        assert np.array_equal(dense_of(got["A"], (qp.nrows, nopt)), Am)     # no duplicates in A: bit-exact
        assert np.array_equal(got["l"], l) and np.array_equal(got["u"], u)
    # rows are stacked in the reference's update order: scalar rows first, then Nonnegatives, Nonpositives, Zeros blocks
    specs = [c.spec for c in model.constraints]
    assert specs == ["scalaraffinefunction_in_lessthan", "vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives",
                     "vectoraffinefunction_in_nonpositives", "vectoraffinefunction_in_zeros"]


def test_handoff_kkt_reproduces_closed_form_and_maximize_flips_sign():
    # test/model.jl:279-297: min x^2 + y^2 + z^2 ... closed form used by the oracle pin: equality-constrained LSQ -> pinv formula
    n, m = 6, 2
    model = P.Model(DenseQPOptimizer(variable_offset=3), quadratic_mode="canonical")
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(3)
    fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape))
    A = P.Parameter(fill, np.zeros((n, n)), model)
    b = P.Parameter(fill, np.zeros(n), model)
    Cm = P.Parameter(fill, np.zeros((m, n)), model)
    d = P.Parameter(fill, np.zeros(m), model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x == d)
    P.solve(model)
    qp = DeviceQP(model)
    got = qp.fetch()
    off = 3
    assert qp.nvars == n + off                                              # optimizer indices 4..n+3: columns 0..2 are empty
    Pu = dense_of(got["P"], (qp.nvars, qp.nvars))[off:, off:]
    Pfull = Pu + np.triu(Pu, 1).T
    Am = dense_of(got["A"], (m, qp.nvars))[:, off:]
    assert np.array_equal(got["l"], got["u"])
    xs = solve_eq_qp(Pfull, got["q"][off:], Am, got["l"])
    Cp = np.linalg.pinv(Cm())
    Pn = np.eye(n) - Cp @ Cm()
    expected = Pn @ (np.linalg.pinv(A() @ Pn) @ (b() - A() @ Cp @ d())) + Cp @ d()
    np.testing.assert_allclose(xs, expected, rtol=1e-6)
    np.testing.assert_allclose(0.5 * xs @ Pfull @ xs + got["q"][off:] @ xs + got["r"], np.sum((A() @ xs - b()) ** 2), rtol=1e-9)

    model2 = P.Model(DenseQPOptimizer())
    y = [Variable(model2) for _ in range(3)]
    c = P.Parameter(lambda v: v.__setitem__(Ellipsis, [1.0, -2.0, 3.0]), np.zeros(3), model2)
    P.objective(model2, P.Maximize, P.dot(c, y))
    P.constraint(model2, y, "<=", P.Parameter(lambda v: v.__setitem__(Ellipsis, 1.0), np.zeros(3), model2))
    P.constraint(model2, y, ">=", P.Parameter(lambda v: v.__setitem__(Ellipsis, -1.0), np.zeros(3), model2))
    P.solve(model2)
    got2 = DeviceQP(model2, infty=1e30).fetch()
    assert got2["q"].tolist() == [-1.0, 2.0, -3.0] and len(got2["P"][0]) == 0
    assert got2["l"].tolist() == [-1.0, -1.0, -1.0, -1e30, -1e30, -1e30] and got2["u"].tolist() == [1e30, 1e30, 1e30, 1.0, 1.0, 1.0]


def test_csc_order_and_values_c_abi_with_duplicates_and_errors():
    import ctypes as C
    import gpu_util as g
    from parametron_jl_amd.handoff import _csc_order
    rows = np.array([3, 1, 3, 2, 1, 3], dtype=np.int64)
    cols = np.array([1, 3, 1, 2, 3, 2], dtype=np.int64)
    perm, seg, col_ptr, row_idx = _csc_order(rows, cols, 3, 3, upper=True)   # folded: (1,3) x4, (2,2), (2,3)
    assert col_ptr.tolist() == [0, 0, 1, 3] and row_idx.tolist() == [1, 0, 1]
    assert perm.tolist() == [3, 0, 1, 2, 4, 5] and seg.tolist() == [0, 1, 5, 6]
    coeff = np.array([1.0, 2.0, 4.0, 8.0, 16.0, 32.0])
    terms = np.zeros(6, dtype=g.QT); terms["coeff"] = coeff
    dt, dperm, dseg, out = g.to_dev(terms), g.to_dev(perm), g.to_dev(seg), g.empty_f64(3)
    g.call("pmt_csc_values_f64", g.ptr(dt), 24, 6, g.ptr(dperm), g.ptr(dseg), 3, -1.0, None, g.ptr(out), g.stream())
    assert g.f64_to_host(out, 3).tolist() == [-8.0, -23.0, -32.0]
    with pytest.raises(P.DimensionMismatch):
        _csc_order(np.array([4], dtype=np.int64), np.array([1], dtype=np.int64), 3, 3, upper=False)
    with pytest.raises(P.ArgumentError):
        g.call("pmt_qp_bounds_f64", g.ptr(out), 3, 7, 0.0, 1e20, g.ptr(out), g.ptr(out), g.stream())
    # long runs take the wave kernel: 40 duplicates per entry
    k, dup = 5, 40
    rr = np.tile(np.arange(1, k + 1, dtype=np.int64), dup); cc = np.ones(k * dup, dtype=np.int64)
    perm, seg, col_ptr, row_idx = _csc_order(rr, cc, k, 1, upper=False)
    vals = np.arange(k * dup, dtype=np.float64)
    lt = np.zeros(k * dup, dtype=g.LT); lt["coeff"] = vals
    out = g.empty_f64(k)
    dlt, dperm, dseg = g.to_dev(lt), g.to_dev(perm), g.to_dev(seg)            # keep the tensors alive across the launch
    g.call("pmt_csc_values_f64", g.ptr(dlt), 16, k * dup, g.ptr(dperm), g.ptr(dseg), k, 1.0, None, g.ptr(out), g.stream())
    assert g.f64_to_host(out, k).tolist() == [vals[i::k].sum() for i in range(k)]


def test_model_with_device_handoff_never_fetches_moi_buffers_and_tracks_parameters():
    n, m = 7, 3
    model = P.Model(P.MockOptimizer(variable_offset=2), quadratic_mode="canonical", handoff="device")
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(8)
    fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape))
    A = P.Parameter(fill, np.zeros((n, n)), model)
    b = P.Parameter(fill, np.zeros(n), model)
    Cm = P.Parameter(fill, np.zeros((m, n)), model)
    d = P.Parameter(fill, np.zeros(m), model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x >= d)
    P.solve(model)
    qp = model.optimizer.device_qp
    assert qp is model.device_qp and qp.nvars == n + 2
    assert model.objective.mode == "canonical-csc" and "quad" not in model.objective.dev    # P's values come straight from the Gram epilogue
    alloc = model.device().bytes_allocated()
    stale = model.objective.f.quadratic_terms["coeff"].copy()
    for _ in range(3):
        P.solve(model)
        got = qp.fetch()
        Pu = dense_of(got["P"], (n + 2, n + 2))[2:, 2:]
        np.testing.assert_allclose(Pu, np.triu(2 * A().T @ A()), rtol=1e-12)
        np.testing.assert_allclose(got["q"][2:], -2 * A().T @ b(), rtol=1e-12)
        assert np.array_equal(dense_of(got["A"], (m, n + 2))[:, 2:], Cm())
        assert np.array_equal(got["l"], 0.0 - (0.0 - d())) and np.all(got["u"] == 1e20)
    assert model.device().bytes_allocated() == alloc
    assert np.array_equal(model.objective.f.quadratic_terms["coeff"], stale)      # host MOI buffers were not refreshed: nothing crossed PCIe


@pytest.mark.parametrize("rows,n", [(100, 300), (700, 300), (33, 129), (1024, 640)])
@pytest.mark.parametrize("with_terms", [False, True])
def test_gram_epilogue_writes_csc_values_identical_to_the_moi_coefficients(rows, n, with_terms):
    """pmt_quad_gram_csc_f64: P.x[k(k+1)/2 + j] = alpha * (MOI coefficient of (j, k)); same contraction, so bit-identical to the
    coefficients pmt_quad_gram_f64 writes — through the whole-tile epilogue (rows <= 256) and the split-K fix-up (rows > 256)."""
    import gpu_util as g
    from oracle import oracle as O
    A = O.fill_uniform(rows * n, 21).reshape(n, rows).T.copy() - 0.25
    b = O.fill_uniform(rows, 22)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    vm = xvar + 5
    dA, db, dx, dvm = g.colmajor(A), g.to_dev(b), g.to_dev(xvar), g.to_dev(vm)
    nq = n * (n + 1) // 2
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8)
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    g.call("pmt_quad_gram_f64", g.ptr(dA), rows, rows, n, g.ptr(dx), g.ptr(db), -1, 1, g.ptr(dvm), g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    ref = g.terms_to_host(oq, nq, g.QT)
    px, oq2, ol2, oc2 = g.empty_f64(nq), g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    g.call("pmt_quad_gram_csc_f64", g.ptr(dA), rows, rows, n, g.ptr(dx), g.ptr(db), -1, g.ptr(dvm), -1.0, g.ptr(px),
           g.ptr(oq2) if with_terms else None, g.ptr(ol2), g.ptr(oc2), g.ptr(ws), g.stream())
    got = g.f64_to_host(px, nq)
    j, k = np.triu_indices(n)                                              # canonical row-major upper-triangular order of `ref`
    want = np.empty(nq)
    want[k * (k + 1) // 2 + j] = -ref["coeff"]
    assert g.same_bits(got, want)
    if with_terms:
        g.assert_terms_equal(g.terms_to_host(oq2, nq, g.QT), ref)
    g.assert_terms_equal(g.terms_to_host(ol2, n, g.LT), g.terms_to_host(ol, n, g.LT))
    assert g.same_bits(g.f64_to_host(oc2, 1), g.f64_to_host(oc, 1))


def test_device_handoff_with_permuted_optimizer_indices_takes_the_generic_path():
    n, m = 6, 2
    model = P.Model(DenseQPOptimizer(permute_seed=4), quadratic_mode="canonical", handoff="device")
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(2)
    fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape))
    A = P.Parameter(fill, np.zeros((n + 3, n)), model)
    b = P.Parameter(fill, np.zeros(n + 3), model)
    Cm = P.Parameter(fill, np.zeros((m, n)), model)
    d = P.Parameter(fill, np.zeros(m), model)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x == d)
    model.initialize()
    assert model.objective.mode == "canonical" and "quad" in model.objective.dev       # not specialised: indices are permuted
    for _ in range(2):
        model.update()
        got = model.device_qp.fetch()
        vm = model.model_var_to_optimizer - 1
        Pu = dense_of(got["P"], (n, n))
        G = 2 * A().T @ A()
        want = np.zeros((n, n))
        want[np.ix_(vm, vm)] = G
        np.testing.assert_allclose(Pu + np.triu(Pu, 1).T, want, rtol=1e-12)
        q = np.zeros(n); q[vm] = -2 * A().T @ b()
        np.testing.assert_allclose(got["q"], q, rtol=1e-12)
        Aw = np.zeros((m, n)); Aw[:, vm] = Cm()
        assert np.array_equal(dense_of(got["A"], (m, n)), Aw)


def test_handoff_with_constant_objective_and_maximize_quadratic():
    """Constant (Parameter-free) objective: P and q are built once on the host from the MOI function; Maximize flips the sign of P, q
    and r.  Constraints are parameterised, so A, l, u still follow the Parameters."""
    n = 4
    model = P.Model(DenseQPOptimizer(variable_offset=1))
    x = [Variable(model) for _ in range(n)]
    rng = np.random.default_rng(9)
    lo = P.Parameter(lambda v: v.__setitem__(Ellipsis, -rng.random(n)), np.zeros(n), model)
    obj = 0.0 - 1.0 * (x[0] ** 2 + 2.0 * x[0] * x[1] + 3.0 * x[1] ** 2 + x[3] ** 2) + 4.0 * x[2] + 1.5      # concave: a Maximize problem
    P.objective(model, P.Maximize, obj)
    P.constraint(model, x, ">=", lo)
    P.constraint(model, 1.0 * x[2] + x[3] <= 2.0)
    P.solve(model)
    qp = DeviceQP(model)
    for _ in range(2):
        got = qp.fetch()
        off = 1
        Pu = dense_of(got["P"], (qp.nvars, qp.nvars))[off:, off:]
        want = np.zeros((n, n)); want[0, 0] = 2.0; want[0, 1] = 2.0; want[1, 1] = 6.0; want[3, 3] = 2.0      # -(-(...)): minimise the negated objective
        assert np.array_equal(Pu, want)
        assert got["q"][off:].tolist() == [0.0, 0.0, -4.0, 0.0] and got["r"] == -1.5
        Am = dense_of(got["A"], (qp.nrows, qp.nvars))[:, off:]
        assert qp.nrows == n + 1 and np.array_equal(Am[0], [0.0, 0.0, 1.0, 1.0]) and np.array_equal(Am[1:], np.eye(n))
        assert got["u"][0] == 2.0 and got["l"][0] == -1e20 and np.array_equal(got["l"][1:], lo()) and np.all(got["u"][1:] == 1e20)
        P.solve(model)
        qp.refresh()


def test_config2_full_size_device_handoff_properties():
    """Config 2 at full size with the device hand-off: P's CSC values are the Gram coefficients at k(k+1)/2 + j (checksum and sampled
    entries against torch), the CSC structure is the dense upper triangle shifted by the optimizer's index offset, A's CSC values are
    C column by column bit for bit, l = u = d."""
    n, r, m = 4096, 4096, 512
    off = 3
    model = P.Model(P.MockOptimizer(variable_offset=off), quadratic_mode="canonical", handoff="device")
    x = [Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r, n), 1, model, advance=False)
    b = P.DeviceUniformParameter((r,), 2, model, advance=False)
    Cm = P.DeviceUniformParameter((m, n), 3, model, advance=False)
    d = P.DeviceUniformParameter((m,), 4, model, scale=2.0, advance=False)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, Cm * x == d)
    P.solve(model)
    qp = model.device_qp
    assert model.objective.mode == "canonical-csc" and qp.nvars == n + off and qp.P.nnz == n * (n + 1) // 2 and qp.A.nnz == m * n
    got = qp.fetch()
    px, pi, pp = got["P"]
    assert pp[:off + 1].tolist() == [0] * (off + 1) and np.array_equal(np.diff(pp[off:]), np.arange(1, n + 1))
    assert pi[:6].tolist() == [off, off, off + 1, off, off + 1, off + 2] and pi[-1] == n + off - 1
    Ah, bh, Ch, dh = A(), b(), Cm(), d()
    row1 = Ah.sum(axis=1)
    assert px.sum() == pytest.approx(float(row1 @ row1 + (Ah * Ah).sum()), rel=1e-11)
    rng = np.random.default_rng(0)
    ks = rng.integers(0, n, 300); js = (rng.random(300) * (ks + 1)).astype(np.int64)
    want = 2 * np.einsum("ij,ij->j", Ah[:, js], Ah[:, ks])
    np.testing.assert_allclose(px[ks * (ks + 1) // 2 + js], want, rtol=1e-12)
    np.testing.assert_allclose(got["q"][off:], -2 * Ah.T @ bh, rtol=1e-12)
    assert np.all(got["q"][:off] == 0.0)
    ax, ai, ap = got["A"]
    assert np.array_equal(ap[off:], np.arange(0, m * n + 1, m)) and np.array_equal(ai[:m], np.arange(m))
    assert np.array_equal(ax.reshape(n, m).T, Ch)                          # column by column = C's own column-major order
    assert np.array_equal(got["l"], dh) and np.array_equal(got["u"], dh)
    model.close()


@pytest.mark.parametrize("case", ["alone", "stacked", "permuted"])
def test_sparse_constraint_block_is_handed_over_from_nzval(case):
    """A sparse constraint matrix (config 5 shape) reaches the solver's CSC `A` from the Parameter's nzval, not through the 24-byte terms:
    alone and in the optimizer's variable order its values ARE the nzval buffer (no launch); stacked under another block, or with permuted
    optimizer indices, it is one gather out of nzval.  Values must track the Parameter across updates either way."""
    m, n, g = 23, 40, 3
    rng = np.random.default_rng(13)
    Cs = sp.random(m, n, density=0.2, format="csc", random_state=rng, data_rvs=lambda k: rng.random(k) + 0.1)
    Cs.sort_indices()
    opt = DenseQPOptimizer(permute_seed=6) if case == "permuted" else P.MockOptimizer()
    model = P.Model(opt, quadratic_mode="canonical", handoff="device")
    x = [Variable(model) for _ in range(n)]
    Cp = P.Parameter(model, val=Cs.copy())
    d = P.Parameter(model, val=rng.random(m))
    G = P.Parameter(model, val=rng.random((g, n)))
    h = P.Parameter(model, val=rng.random(g))
    P.objective(model, P.Minimize, P.dot(np.ones(n), x))
    if case == "stacked":
        P.constraint(model, G * x <= h)
    P.constraint(model, Cp * x == d)
    model.initialize()
    qp = model.device_qp
    rows = m + (g if case == "stacked" else 0)
    assert (qp.A.values_ptr == list(model.constraints)[-1].expr.out.spmat.buf) == (case == "alone")
    for k in range(3):
        Cp.val.data[...] = rng.random(Cs.nnz) + 0.1 * k
        d.val[...] = rng.random(m)
        model.update()
        got = qp.fetch()
        check_csc_canonical(got["A"])
        want = np.zeros((rows, n))
        vm = model.model_var_to_optimizer - 1
        want[rows - m:, vm] = Cp.val.toarray()
        if case == "stacked":
            want[:g, vm] = G.val
        assert np.array_equal(dense_of(got["A"], (rows, n)), want)
        assert np.array_equal(got["l"][rows - m:], d.val) and np.array_equal(got["u"][rows - m:], d.val)


@pytest.mark.parametrize("use_graph", [False, True])
def test_handoff_recorded_on_the_side_lane_equals_the_launches_behind_the_tape(use_graph):
    """With a Gram objective and constraints built straight from Parameters the hand-off launches (one fused gather for A, one for q, one
    bounds kernel) are side-lane entries of the model's tape (DeviceQP._in_tape); with side_lane=False they are launched behind the tape on
    every update.  Both must hand over the same QP, update after update, with the launch tape and with hipGraph replay."""
    n, r, m = 40, 64, 6

    def build(side_lane):
        model = P.Model(P.MockOptimizer(variable_offset=1), quadratic_mode="canonical", handoff="device", use_graph=use_graph, side_lane=side_lane)
        x = [Variable(model) for _ in range(n)]
        rng = np.random.default_rng(31)
        fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape) - 0.4)
        A = P.Parameter(fill, np.zeros((r, n)), model)
        b = P.Parameter(fill, np.zeros(r), model)
        G = P.Parameter(fill, np.zeros((m, n)), model)
        h = P.Parameter(fill, np.zeros(m), model)
        lo = P.Parameter(fill, np.zeros(n), model)
        residual = A * x - b
        P.objective(model, P.Minimize, P.dot(residual, residual))
        P.constraint(model, G * x, "<=", h)
        P.constraint(model, x, ">=", lo)
        model.initialize()
        return model

    a, b = build(True), build(False)
    assert a.device_qp._in_tape and not b.device_qp._in_tape
    for _ in range(3):
        a.update(); b.update()
        qa, qb = a.device_qp.fetch(), b.device_qp.fetch()
        for key in ("q", "l", "u"):
            assert np.array_equal(qa[key], qb[key])
        for key in ("P", "A"):
            for u, v in zip(qa[key], qb[key]):
                assert np.array_equal(u, v)
        assert qa["r"] == qb["r"]
    a.close(); b.close()
