"""-m gpu parity tests: every kernel entry point of the C ABI against the CPU oracle (bit-exact for indices
and for coefficients that are copies or single products; 1e-12 relative for the canonical sums)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402


def G():
    import gpu_util
    return gpu_util


def _rand_mat(rows, cols, seed):
    return O.fill_uniform(rows * cols, seed).reshape(cols, rows).T.copy()     # column-major fill like the device stream


# ------------------------------------------------------------------ synthetic input stream
def test_fill_uniform_matches_oracle_stream():
    g = G()
    n = 100003
    d = g.empty_f64(n)
    g.call("pmt_fill_uniform_f64", g.ptr(d), n, 12345, 2.0, g.stream())
    assert g.same_bits(g.f64_to_host(d, n), O.fill_uniform(n, 12345, 2.0))


@pytest.mark.parametrize("n,shift", [(1, 0), (1, 1), (2, 1), (7, 0), (8, 1), (513, 1), (100003, 1), (70000, 0)])
def test_fill_uniform_odd_lengths_and_unaligned_bases(n, shift):
    """the fill writes 16 bytes per lane: an 8-byte-aligned base, an odd length and the one-element cases take the single-element head / tail;
    neighbours of the range stay untouched; the offset form continues the same stream"""
    g = G()
    buf = g.empty_f64(n + 4)
    buf.fill_(-7.0)
    d = buf[1 + shift:]                                             # (torch allocations are 16-byte aligned: shift picks the parity)
    g.call("pmt_fill_uniform_f64", g.ptr(d), n, 77, 1.5, g.stream())
    host = g.f64_to_host(buf, n + 4)
    want = O.fill_uniform(n, 77, 1.5)
    assert g.same_bits(host[1 + shift:1 + shift + n], want)
    assert np.all(host[:1 + shift] == -7.0) and np.all(host[1 + shift + n:] == -7.0)
    k = n // 3
    g.call("pmt_fill_uniform_offset_f64", g.ptr(d), n - k, C.c_uint64(77), C.c_uint64(k), 1.5, g.stream())
    assert g.same_bits(g.f64_to_host(d, n - k), want[k:])


@pytest.mark.parametrize("rows,cols,pad,shift", [(1, 1, 0, 0), (1, 9, 2, 1), (2, 3, 0, 0), (5, 7, 1, 0), (64, 33, 0, 1), (513, 70, 3, 0), (4096, 40, 64, 0),
                                                 (100, 70000, 0, 0)])
def test_fill_uniform_matrix_is_the_contiguous_stream_placed_with_a_pitch(rows, cols, pad, shift):
    """pmt_fill_uniform_matrix_f64: column c of the padded device matrix holds elements c*rows .. of the CONTIGUOUS stream (what the oracle
    fills); the padding rows are not written; odd row counts, odd leading dimensions and unaligned bases take the single-element stores"""
    g = G()
    lda = rows + pad
    buf = g.empty_f64(lda * cols + 2)
    buf.fill_(-3.0)
    d = buf[shift:]
    g.call("pmt_fill_uniform_matrix_f64", g.ptr(d), rows, cols, lda, C.c_uint64(9), 2.0, g.stream())
    host = g.f64_to_host(buf, lda * cols + 2)
    m = host[shift:shift + lda * cols].reshape(cols, lda)
    assert g.same_bits(m[:, :rows].copy().reshape(-1), O.fill_uniform(rows * cols, 9, 2.0))
    assert np.all(m[:, rows:] == -3.0) and np.all(host[:shift] == -3.0) and np.all(host[shift + lda * cols:] == -3.0)


# ------------------------------------------------------------------ affine_assemble (matvecmul! + vecadd!/vecsubtract!)
@pytest.mark.parametrize("rows,cols,pad", [(3, 4, 0), (8, 8, 0), (2, 8, 0), (64, 64, 0), (65, 129, 1), (100, 37, 3),
                                           (128, 256, 0), (1, 1, 0), (512, 1024, 0)])
@pytest.mark.parametrize("sign", [-1, 1, 0])
def test_affine_assemble_bit_exact(rows, cols, pad, sign):
    g = G()
    A = _rand_mat(rows, cols, 1)
    A[0, 0] = -0.0
    b = O.fill_uniform(rows, 2) - 0.5
    b[-1] = -0.0
    xvar = (np.arange(cols, dtype=np.int64) * 3 + 5)
    lda = rows + pad
    Apad = np.zeros((lda, cols)); Apad[:rows] = A
    dA, db, dx = g.colmajor(Apad), g.to_dev(b), g.to_dev(xvar)
    out, consts = g.empty_terms(rows * cols, g.LT), g.empty_f64(rows)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), lda, rows, cols, g.ptr(dx), g.ptr(db) if sign else None, sign,
           g.ptr(out), g.ptr(consts), g.stream())
    ref = O.AffVec(rows).matvecmul_vars(A, xvar)
    if sign:
        ref = O.AffVec(rows).vecaddsub(ref, b, subtract=(sign < 0))
    terms, row_ptr, rconsts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(out, rows * cols, g.LT), terms)
    assert g.same_bits(g.f64_to_host(consts, rows), rconsts)


@pytest.mark.parametrize("rows,cols,pad", [(2, 8, 0), (64, 64, 0), (64, 128, 0), (65, 129, 1), (100, 37, 0), (130, 200, 2), (512, 4096 // 4, 0)])
def test_affine_pack_vector_bit_exact(rows, cols, pad):
    g = G()
    A = _rand_mat(rows, cols, 3)
    d = O.fill_uniform(rows, 4, 2.0)
    nvars = cols + 7
    rng = np.random.default_rng(0)
    xvar = np.sort(rng.choice(np.arange(1, nvars + 1), size=cols, replace=False)).astype(np.int64)
    varmap = rng.permutation(nvars).astype(np.int64) + 1
    lda = rows + pad
    Apad = np.zeros((lda, cols)); Apad[:rows] = A
    dA, dd, dx, dvm = g.colmajor(Apad), g.to_dev(d), g.to_dev(xvar), g.to_dev(varmap)
    out, consts = g.empty_terms(rows * cols, g.VAT), g.empty_f64(rows)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dA), lda, rows, cols, g.ptr(dx), g.ptr(dd), -1, g.ptr(dvm), 0,
           g.ptr(out), g.ptr(consts), g.stream())
    ref = O.AffVec(rows).vecsubtract(O.AffVec(rows).matvecmul_vars(A, xvar), d)
    terms, rconsts = ref.moi(varmap)
    g.assert_terms_equal(g.terms_to_host(out, rows * cols, g.VAT), terms)
    assert g.same_bits(g.f64_to_host(consts, rows), rconsts)
    # identity varmap + row offset (second constraint block stacked below the first)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dA), lda, rows, cols, g.ptr(dx), g.ptr(dd), 1, None, 10,
           g.ptr(out), g.ptr(consts), g.stream())
    terms, rconsts = O.AffVec(rows).vecadd(O.AffVec(rows).matvecmul_vars(A, xvar), d).moi()
    terms["out"] += 10
    g.assert_terms_equal(g.terms_to_host(out, rows * cols, g.VAT), terms)
    assert g.same_bits(g.f64_to_host(consts, rows), rconsts)


def test_affine_dimension_errors():
    g = G()
    from parametron_jl_amd import DimensionMismatch, ArgumentError
    d = g.empty_f64(16)
    with pytest.raises(DimensionMismatch):
        g.call("pmt_affine_assemble_f64", g.ptr(d), 2, 4, 4, g.ptr(d), None, 0, g.ptr(d), g.ptr(d), g.stream())   # lda < rows
    with pytest.raises(ArgumentError):
        g.call("pmt_affine_assemble_f64", g.ptr(d), 4, 4, 4, g.ptr(d), None, -1, g.ptr(d), g.ptr(d), g.stream())  # sign needs b


def test_vars_addsub_bounds():
    g = G()
    n = 1000
    xvar = np.arange(1, n + 1, dtype=np.int64)[::-1].copy()
    l = O.fill_uniform(n, 9) - 0.5
    varmap = np.random.default_rng(1).permutation(n).astype(np.int64) + 1
    dx, dl, dvm = g.to_dev(xvar), g.to_dev(l), g.to_dev(varmap)
    lt, vat, consts = g.empty_terms(n, g.LT), g.empty_terms(n, g.VAT), g.empty_f64(n)
    g.call("pmt_vars_addsub_f64", g.ptr(dx), n, g.ptr(dl), -1, g.ptr(dvm), 3, g.ptr(lt), g.ptr(vat), g.ptr(consts), g.stream())
    ref = O.AffVec(n).vecsubtract(xvar, l)
    terms, _, rconsts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(lt, n, g.LT), terms)
    mt, mc = ref.moi(varmap)
    mt["out"] += 3
    g.assert_terms_equal(g.terms_to_host(vat, n, g.VAT), mt)
    assert g.same_bits(g.f64_to_host(consts, n), rconsts)


# ------------------------------------------------------------------ literal quadratic expansion
def _affvec_dev(g, ref):
    terms, row_ptr, consts = ref.flat()
    return g.to_dev(terms), g.to_dev(consts), terms, consts


@pytest.mark.parametrize("rows,n", [(2, 3), (8, 8), (5, 33), (3, 1030), (16, 128)])
@pytest.mark.parametrize("moi", [0, 1])
def test_quad_expand_residual_dot_residual(rows, n, moi):
    g = G()
    A = _rand_mat(rows, n, 11)
    b = O.fill_uniform(rows, 12)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    varmap = np.random.default_rng(2).permutation(n).astype(np.int64) + 1
    w = O.LsqWorkspace(n, rows, 1)
    w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
    # device: affine node then literal expansion
    dA, db, dx, dvm = g.colmajor(A), g.to_dev(b), g.to_dev(xvar), g.to_dev(varmap)
    res, rc = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), rows, rows, n, g.ptr(dx), g.ptr(db), -1, g.ptr(res), g.ptr(rc), g.stream())
    oq, ol, oc = g.empty_terms(rows * n * n, g.QT), g.empty_terms(2 * rows * n, g.LT), g.empty_f64(1)
    g.call("pmt_quad_expand_f64", rows, g.ptr(res), n, g.ptr(rc), g.ptr(res), n, g.ptr(rc), moi, g.ptr(dvm) if moi else None,
           g.ptr(oq), g.ptr(ol), g.ptr(oc), g.stream())
    if moi:
        at, qt, const = w.objective.moi(varmap)
    else:
        qt, at, const = w.objective.terms(), w.objective.affine.terms(), w.objective.affine.constant
    g.assert_terms_equal(g.terms_to_host(oq, rows * n * n, g.QT), qt)
    g.assert_terms_equal(g.terms_to_host(ol, 2 * rows * n, g.LT), at)
    assert g.same_bits(g.f64_to_host(oc, 1), [const])


def test_quad_expand_two_different_vectors_duplicate_vars():
    # x and y different, variables repeated across positions: doubling keys on variable identity (moi_interop.jl:58)
    g = G()
    rows, nx, ny = 4, 5, 7
    rng = np.random.default_rng(3)
    X, Y = O.AffVec(rows), O.AffVec(rows)
    for i in range(rows):
        for a in range(nx):
            X[i].push(rng.standard_normal(), int(rng.integers(1, 4)))
        for b in range(ny):
            Y[i].push(rng.standard_normal(), int(rng.integers(1, 4)))
        X[i].set_constant(rng.standard_normal()); Y[i].set_constant(rng.standard_normal())
    ref = O.Quad().vecdot_affs_affs(X, Y)
    dxt, dxc, _, _ = _affvec_dev(g, X)
    dyt, dyc, _, _ = _affvec_dev(g, Y)
    varmap = np.array([3, 1, 2], dtype=np.int64)
    dvm = g.to_dev(varmap)
    oq, ol, oc = g.empty_terms(rows * nx * ny, g.QT), g.empty_terms(rows * (nx + ny), g.LT), g.empty_f64(1)
    g.call("pmt_quad_expand_f64", rows, g.ptr(dxt), nx, g.ptr(dxc), g.ptr(dyt), ny, g.ptr(dyc), 1, g.ptr(dvm),
           g.ptr(oq), g.ptr(ol), g.ptr(oc), g.stream())
    at, qt, const = ref.moi(varmap)
    g.assert_terms_equal(g.terms_to_host(oq, rows * nx * ny, g.QT), qt)
    g.assert_terms_equal(g.terms_to_host(ol, rows * (nx + ny), g.LT), at)
    assert g.same_bits(g.f64_to_host(oc, 1), [const])


# ------------------------------------------------------------------ canonical Gram objective (f64 MFMA)
def _gram_device(g, A, b, sign, xvar, varmap, moi, lda=None):
    rows, n = A.shape
    lda = lda or rows
    Apad = np.zeros((lda, n)); Apad[:rows] = A
    dA, db, dx = g.colmajor(Apad), g.to_dev(b), g.to_dev(xvar)
    dvm = g.to_dev(varmap) if varmap is not None else None
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(max(1, g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8))
    g.call("pmt_quad_gram_f64", g.ptr(dA), lda, rows, n, g.ptr(dx), g.ptr(db), sign, moi, g.ptr(dvm), g.ptr(oq), g.ptr(ol), g.ptr(oc),
           g.ptr(ws), g.stream())
    return g.terms_to_host(oq, nq, g.QT), g.terms_to_host(ol, n, g.LT), g.f64_to_host(oc, 1)[0]


@pytest.mark.parametrize("rows,n", [(8, 8), (5, 3), (40, 24), (17, 130), (3, 1), (64, 48)])
@pytest.mark.parametrize("moi", [0, 1])
def test_quad_gram_equals_canonicalize_of_literal(rows, n, moi):
    g = G()
    A = _rand_mat(rows, n, 21)
    b = O.fill_uniform(rows, 22)
    xvar = np.arange(1, n + 1, dtype=np.int64) * 2            # strictly increasing, not contiguous
    varmap = np.random.default_rng(4).permutation(2 * n).astype(np.int64) + 1
    w = O.LsqWorkspace(n, rows, 1)
    w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
    w.objective.canonicalize()
    if moi:
        at, qt, const = w.objective.moi(varmap)
    else:
        qt, at, const = w.objective.terms(), w.objective.affine.terms(), w.objective.affine.constant
    gq, gl, gc = _gram_device(g, A, b, -1, xvar, varmap if moi else None, moi, lda=rows + (rows % 2))
    assert np.array_equal(gq["row"], qt["row"]) and np.array_equal(gq["col"], qt["col"])     # indices bit-exact
    assert np.array_equal(gl["var"], at["var"])
    np.testing.assert_allclose(gq["coeff"], qt["coeff"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(gl["coeff"], at["coeff"], rtol=1e-12, atol=0)
    order, seq = g.constant_in_the_library_order(rows, n, b)                               # tiny shapes and n > 128: the reference's sequential sum;
    assert gc == seq and (order != 0 or gc == const)                                       # (64, 48) takes the fused tall form: its own fixed order
    assert gc == pytest.approx(const, rel=1e-14)


@pytest.mark.parametrize("rows,n", [(300, 260), (1024, 512), (77, 385)])
def test_quad_gram_medium_against_numpy(rows, n):
    g = G()
    A = _rand_mat(rows, n, 31)
    b = O.fill_uniform(rows, 32)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    gq, gl, gc = _gram_device(g, A, b, -1, xvar, None, 1)
    G2 = 2 * (A.T @ A)
    iu = np.triu_indices(n)
    assert np.array_equal(gq["row"], iu[0] + 1) and np.array_equal(gq["col"], iu[1] + 1)
    np.testing.assert_allclose(gq["coeff"], G2[iu], rtol=1e-12, atol=0)
    np.testing.assert_allclose(gl["coeff"], -2 * (A.T @ b), rtol=1e-12, atol=0)
    assert np.array_equal(gl["var"], xvar)
    assert gc == pytest.approx(b @ b, rel=1e-14)


def test_quad_gram_a_is_identity_with_asymmetric_layout():
    # transpose-detecting check (cdna_hip_programming.md §3): A with distinct row/col structure
    g = G()
    rows, n = 6, 200
    A = np.zeros((rows, n))
    for i in range(rows):
        A[i, 7 * i + 3] = i + 1.0
        A[i, 150 + i] = 0.5
    b = np.zeros(rows)
    gq, gl, gc = _gram_device(g, A, b, 0, np.arange(1, n + 1, dtype=np.int64), None, 1)
    G2 = 2 * (A.T @ A)
    assert np.array_equal(gq["coeff"], G2[np.triu_indices(n)])


# ------------------------------------------------------------------ other quadratic builders
def test_bilinear_and_vecdot_terms():
    g = G()
    n = 37
    Q = _rand_mat(n, n, 41)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    varmap = np.random.default_rng(5).permutation(n).astype(np.int64) + 1
    dQ, dx, dvm = g.colmajor(Q), g.to_dev(xvar), g.to_dev(varmap)
    oq = g.empty_terms(n * n, g.QT)
    ref = O.Quad().bilinearmul(Q, xvar, xvar)
    g.call("pmt_bilinear_f64", g.ptr(dQ), n, n, n, g.ptr(dx), g.ptr(dx), 0, None, g.ptr(oq), g.stream())
    g.assert_terms_equal(g.terms_to_host(oq, n * n, g.QT), ref.terms())
    g.call("pmt_bilinear_f64", g.ptr(dQ), n, n, n, g.ptr(dx), g.ptr(dx), 1, g.ptr(dvm), g.ptr(oq), g.stream())
    # padded leading dimension of the device copy: same terms
    ldq = n + 5
    Qpad = np.zeros((ldq, n)); Qpad[:n] = Q
    dQp = g.colmajor(Qpad)
    g.call("pmt_bilinear_f64", g.ptr(dQp), ldq, n, n, g.ptr(dx), g.ptr(dx), 1, g.ptr(dvm), g.ptr(oq), g.stream())
    g.assert_terms_equal(g.terms_to_host(oq, n * n, g.QT), ref.moi(varmap)[1])
    # x . x  (Variable . Variable) and (a .* x) . x
    a = O.fill_uniform(n, 42)
    da = g.to_dev(a)
    g.call("pmt_vecdot_terms_f64", n, None, g.ptr(dx), None, g.ptr(dx), 1, g.ptr(dvm), g.ptr(oq), g.stream())
    g.assert_terms_equal(g.terms_to_host(oq, n, g.QT), O.Quad().vecdot_vars_vars(xvar, xvar).moi(varmap)[1])
    g.call("pmt_vecdot_terms_f64", n, g.ptr(da), g.ptr(dx), None, g.ptr(dx), 0, None, g.ptr(oq), g.stream())
    ref2 = O.Quad().vecdot_terms_terms(list(zip(a, xvar)), [(1.0, v) for v in xvar])
    g.assert_terms_equal(g.terms_to_host(oq, n, g.QT), ref2.terms())


def test_vecdot_affs_vars():
    g = G()
    rows, L = 9, 13
    A = _rand_mat(rows, L, 51)
    b = O.fill_uniform(rows, 52)
    xv = np.arange(1, L + 1, dtype=np.int64)
    yv = np.arange(1, rows + 1, dtype=np.int64)
    X = O.AffVec(rows).vecadd(O.AffVec(rows).matvecmul_vars(A, xv), b)
    ref = O.Quad().vecdot_affs_vars(X, yv)
    dxt, dxc, _, _ = _affvec_dev(g, X)
    dy = g.to_dev(yv)
    oq, ol = g.empty_terms(rows * L, g.QT), g.empty_terms(rows, g.LT)
    g.call("pmt_vecdot_affs_vars_f64", rows, g.ptr(dxt), L, g.ptr(dxc), g.ptr(dy), 0, None, g.ptr(oq), g.ptr(ol), g.stream())
    g.assert_terms_equal(g.terms_to_host(oq, rows * L, g.QT), ref.terms())
    g.assert_terms_equal(g.terms_to_host(ol, rows, g.LT), ref.affine.terms())
    g.call("pmt_vecdot_affs_vars_f64", rows, g.ptr(dxt), L, g.ptr(dxc), g.ptr(dy), 1, None, g.ptr(oq), g.ptr(ol), g.stream())
    at, qt, _ = ref.moi()
    g.assert_terms_equal(g.terms_to_host(oq, rows * L, g.QT), qt)


# ------------------------------------------------------------------ generic term-list builders
def test_affvec_combine_scale_matvecmul_affs():
    g = G()
    rows, L = 6, 5
    B = _rand_mat(rows, L, 61) - 0.5
    c = O.fill_uniform(rows, 62) - 0.5
    xv = np.arange(1, L + 1, dtype=np.int64)
    X = O.AffVec(rows).vecsubtract(O.AffVec(rows).matvecmul_vars(B, xv), c)
    Y = O.AffVec(rows).vecadd(O.AffVec(rows).matvecmul_vars(2 * B, xv[::-1].copy()), c)
    dxt, dxc, _, _ = _affvec_dev(g, X)
    dyt, dyc, _, _ = _affvec_dev(g, Y)
    for sb, sub in ((1, False), (-1, True)):
        ref = O.AffVec(rows).vecaddsub(X, Y, subtract=sub)
        ot, oc = g.empty_terms(rows * 2 * L, g.LT), g.empty_f64(rows)
        g.call("pmt_affvec_combine_f64", rows, g.ptr(dxt), None, L, g.ptr(dxc), g.ptr(dyt), None, L, g.ptr(dyc), sb,
               g.ptr(ot), None, 2 * L, g.ptr(oc), g.stream())
        terms, _, consts = ref.flat()
        g.assert_terms_equal(g.terms_to_host(ot, rows * 2 * L, g.LT), terms)
        assert g.same_bits(g.f64_to_host(oc, rows), consts)
    # numbers - affs  (copyto!(dest, number); subtract!(dest, aff))
    ref = O.AffVec(rows).vecsubtract(c, X)
    dc = g.to_dev(c)
    ot, oc = g.empty_terms(rows * L, g.LT), g.empty_f64(rows)
    g.call("pmt_affvec_combine_f64", rows, None, None, 0, g.ptr(dc), g.ptr(dxt), None, L, g.ptr(dxc), -1,
           g.ptr(ot), None, L, g.ptr(oc), g.stream())
    terms, _, consts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(ot, rows * L, g.LT), terms)
    assert g.same_bits(g.f64_to_host(oc, rows), consts)
    # scale!
    ref = O.AffVec(rows).scale_number_affs(-1.75, X)
    g.call("pmt_affvec_scale_f64", rows, rows * L, g.ptr(dxt), g.ptr(dxc), None, -1.75, g.ptr(ot), g.ptr(oc), g.stream())
    terms, _, consts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(ot, rows * L, g.LT), terms)
    assert g.same_bits(g.f64_to_host(oc, rows), consts)
    # matvecmul!(y, A, x::Vector{AffineFunction})
    m = 4
    A = _rand_mat(m, rows, 63) - 0.5
    ref = O.AffVec(m).matvecmul_affs(A, X)
    dA = g.colmajor(A)
    ot, oc = g.empty_terms(m * rows * L, g.LT), g.empty_f64(m)
    g.call("pmt_matvecmul_affs_f64", g.ptr(dA), m, m, rows, g.ptr(dxt), L, g.ptr(dxc), g.ptr(ot), g.ptr(oc), g.stream())
    terms, _, consts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(ot, m * rows * L, g.LT), terms)
    assert g.same_bits(g.f64_to_host(oc, m), consts)
    # dot(v, x) and dot(v, X)
    v = O.fill_uniform(rows, 64) - 0.5
    dv = g.to_dev(v)
    yv = np.arange(10, 10 + rows, dtype=np.int64)
    dyv = g.to_dev(yv)
    ot, oc1 = g.empty_terms(rows, g.LT), g.empty_f64(1)
    g.call("pmt_vecdot_numbers_vars_f64", g.ptr(dv), g.ptr(dyv), rows, g.ptr(ot), g.ptr(oc1), g.stream())
    r = O.vecdot_aff_numbers_vars(v, yv)
    g.assert_terms_equal(g.terms_to_host(ot, rows, g.LT), r.terms())
    assert g.f64_to_host(oc1, 1)[0] == 0.0
    ot = g.empty_terms(rows * L, g.LT)
    g.call("pmt_vecdot_numbers_affs_f64", g.ptr(dv), rows, g.ptr(dxt), L, g.ptr(dxc), g.ptr(ot), g.ptr(oc1), g.stream())
    r = O.vecdot_aff_numbers_affs(v, X)
    g.assert_terms_equal(g.terms_to_host(ot, rows * L, g.LT), r.terms())
    assert g.same_bits(g.f64_to_host(oc1, 1), [r.constant])


def test_ragged_combine_and_pack_vector_affine():
    # vcat!-style stacking of rows with different lengths through row_ptr
    g = G()
    rng = np.random.default_rng(7)
    rows = 7
    lens = [3, 0, 5, 1, 4, 2, 6]
    X = O.AffVec(rows)
    for i, ln in enumerate(lens):
        for _ in range(ln):
            X[i].push(rng.standard_normal(), int(rng.integers(1, 9)))
        X[i].set_constant(rng.standard_normal())
    terms, row_ptr, consts = X.flat()
    dt, dc, dp = g.to_dev(terms) if len(terms) else None, g.to_dev(consts), g.to_dev(row_ptr)
    ot, oc = g.empty_terms(len(terms), g.LT), g.empty_f64(rows)
    g.call("pmt_affvec_combine_f64", rows, g.ptr(dt), g.ptr(dp), 0, g.ptr(dc), None, None, 0, None, 1,
           g.ptr(ot), g.ptr(dp), 0, g.ptr(oc), g.stream())
    g.assert_terms_equal(g.terms_to_host(ot, len(terms), g.LT), terms)
    varmap = rng.permutation(8).astype(np.int64) + 1
    dvm = g.to_dev(varmap)
    ov = g.empty_terms(len(terms), g.VAT)
    g.call("pmt_pack_vector_affine_f64", g.ptr(dt), g.ptr(dp), rows, 0, g.ptr(dvm), 0, g.ptr(ov), g.stream())
    g.assert_terms_equal(g.terms_to_host(ov, len(terms), g.VAT), X.moi(varmap)[0])


def test_pack_scalar_functions():
    g = G()
    rng = np.random.default_rng(8)
    n = 1000
    q = O.Quad()
    for _ in range(n):
        q.add_term(rng.standard_normal(), int(rng.integers(1, 6)), int(rng.integers(1, 6)))
        q.affine.push(rng.standard_normal(), int(rng.integers(1, 6)))
    varmap = rng.permutation(5).astype(np.int64) + 1
    dq, dl, dvm = g.to_dev(q.terms()), g.to_dev(q.affine.terms()), g.to_dev(varmap)
    oq, ol = g.empty_terms(n, g.QT), g.empty_terms(n, g.LT)
    g.call("pmt_pack_scalar_quadratic_f64", g.ptr(dq), n, g.ptr(dvm), g.ptr(oq), g.stream())
    g.call("pmt_pack_scalar_affine_f64", g.ptr(dl), n, g.ptr(dvm), g.ptr(ol), g.stream())
    at, qt, _ = q.moi(varmap)
    g.assert_terms_equal(g.terms_to_host(oq, n, g.QT), qt)
    g.assert_terms_equal(g.terms_to_host(ol, n, g.LT), at)


# ------------------------------------------------------------------ sparse constraint block
def test_sparse_pack_vector_equals_dense_reference_without_structural_zeros():
    g = G()
    import scipy.sparse as sp
    m, n = 40, 70
    rng = np.random.default_rng(9)
    Cs = sp.random(m, n, density=0.05, format="csc", random_state=rng, data_rvs=lambda k: rng.random(k) + 0.1)
    colptr = Cs.indptr.astype(np.int64) + 1
    rowval = Cs.indices.astype(np.int64) + 1
    nnz = Cs.nnz
    perm, trow, tcol, row_ptr = (np.empty(nnz, np.int64), np.empty(nnz, np.int64), np.empty(nnz, np.int64), np.empty(m + 1, np.int64))
    g.call("pmt_sparse_rowmajor_order", m, n, colptr.ctypes.data_as(C.c_void_p), rowval.ctypes.data_as(C.c_void_p),
           perm.ctypes.data_as(C.c_void_p), trow.ctypes.data_as(C.c_void_p), tcol.ctypes.data_as(C.c_void_p),
           row_ptr.ctypes.data_as(C.c_void_p))
    xvar = np.arange(1, n + 1, dtype=np.int64)
    varmap = rng.permutation(n).astype(np.int64) + 1
    dnz, dperm, drow, dvar, dvm = g.to_dev(Cs.data), g.to_dev(perm), g.to_dev(trow), g.to_dev(xvar[tcol - 1]), g.to_dev(varmap)
    out = g.empty_terms(nnz, g.VAT)
    g.call("pmt_sparse_pack_vector_f64", g.ptr(dnz), g.ptr(dperm), g.ptr(drow), g.ptr(dvar), nnz, g.ptr(dvm), 0, g.ptr(out), g.stream())
    dense_terms, _ = O.AffVec(m).matvecmul_vars(Cs.toarray(), xvar).moi(varmap)
    want = dense_terms[dense_terms["coeff"] != 0.0]          # all structural values are non-zero here
    g.assert_terms_equal(g.terms_to_host(out, nnz, g.VAT), want)


# ------------------------------------------------------------------ plan: record once, replay, graph
def test_plan_record_replay_and_graph():
    g = G()
    L = g.lib()
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, None, C.byref(plan))
    try:
        rows, n = 16, 24
        def alloc(nbytes):
            p = C.c_void_p()
            g.call("pmt_plan_alloc", plan, nbytes, C.byref(p))
            return p
        dA, db, dx = alloc(rows * n * 8), alloc(rows * 8), alloc(n * 8)
        out, consts = alloc(rows * n * 16), alloc(rows * 8)
        xvar = np.arange(1, n + 1, dtype=np.int64)
        g.call("pmt_plan_upload", plan, dx, xvar.ctypes.data_as(C.c_void_p), xvar.nbytes)
        rec = C.c_void_p(L.pmt_plan_recording_stream(plan))
        g.call("pmt_plan_begin_record", plan)
        g.call("pmt_fill_uniform_f64", dA, rows * n, 1, 1.0, rec)
        g.call("pmt_fill_uniform_f64", db, rows, 2, 1.0, rec)
        g.call("pmt_affine_assemble_f64", dA, rows, rows, n, dx, db, -1, out, consts, rec)
        g.call("pmt_plan_end_record", plan)
        assert L.pmt_plan_tape_length(plan) == 3
        A = O.fill_uniform(rows * n, 1).reshape(n, rows).T
        ref = O.AffVec(rows).vecsubtract(O.AffVec(rows).matvecmul_vars(A, xvar), O.fill_uniform(rows, 2))
        terms, _, rconsts = ref.flat()
        for use_graph in (False, True):
            if use_graph:
                g.call("pmt_plan_instantiate_graph", plan)
            got = np.empty(rows * n, dtype=g.LT)
            gc = np.empty(rows)
            g.call("pmt_plan_update", plan)
            g.call("pmt_plan_fetch", plan, got.ctypes.data_as(C.c_void_p), out, got.nbytes)
            g.call("pmt_plan_fetch", plan, gc.ctypes.data_as(C.c_void_p), consts, gc.nbytes)
            g.call("pmt_plan_synchronize", plan)
            g.assert_terms_equal(got, terms)
            assert g.same_bits(gc, rconsts)
    finally:
        g.call("pmt_plan_destroy", plan)


# ------------------------------------------------------------------ BASELINE config 2 sizes: size-independent properties
def test_full_size_affine_and_gram_properties():
    g = G()
    n = r = 4096
    m = 512
    dA, db = g.empty_f64(r * n), g.empty_f64(r)
    g.call("pmt_fill_uniform_f64", g.ptr(dA), r * n, 1, 1.0, g.stream())
    g.call("pmt_fill_uniform_f64", g.ptr(db), r, 2, 1.0, g.stream())
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=g.DEV)
    out, consts = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(out), g.ptr(consts), g.stream())
    torch.cuda.synchronize()
    terms = out.view(r, n, 2)
    assert torch.equal(terms[:, :, 0].view(torch.float64), dA.view(n, r).t())          # coefficients are exact copies of A
    assert torch.equal(terms[:, :, 1], xvar.unsqueeze(0).expand(r, n))                 # variable index = column
    assert torch.equal(consts, 0.0 - db)
    # constraint block straight to MOI triplets
    dC, dd = g.empty_f64(m * n), g.empty_f64(m)
    g.call("pmt_fill_uniform_f64", g.ptr(dC), m * n, 3, 1.0, g.stream())
    g.call("pmt_fill_uniform_f64", g.ptr(dd), m, 4, 2.0, g.stream())
    vat, vc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dC), m, m, n, g.ptr(xvar), g.ptr(dd), -1, None, 0, g.ptr(vat), g.ptr(vc), g.stream())
    torch.cuda.synchronize()
    t3 = vat.view(m, n, 3)
    assert torch.equal(t3[:, :, 0], torch.arange(1, m + 1, device=g.DEV).unsqueeze(1).expand(m, n))
    assert torch.equal(t3[:, :, 1].view(torch.float64), dC.view(n, m).t())
    assert torch.equal(t3[:, :, 2], xvar.unsqueeze(0).expand(m, n))
    assert torch.equal(vc, 0.0 - dd)
    # canonical objective vs a float64 torch reference of the same contraction (tolerance from north_star: 1e-12 relative)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(max(1, g.lib().pmt_quad_gram_workspace_bytes(r, n) // 8))
    g.call("pmt_quad_gram_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    torch.cuda.synchronize()
    Am = dA.view(n, r).t()                                                             # (r, n) view of column-major A
    G2 = 2.0 * (Am.t() @ Am)
    iu = torch.triu_indices(n, n, device=g.DEV)
    q3 = oq.view(nq, 3)
    assert torch.equal(q3[:, 1], iu[0] + 1) and torch.equal(q3[:, 2], iu[1] + 1)
    got = q3[:, 0].view(torch.float64)
    want = G2[iu[0], iu[1]]
    assert float(((got - want).abs() / want.abs()).max()) <= 1e-12
    l2 = ol.view(n, 2)
    wl = -2.0 * (Am.t() @ db)
    assert float(((l2[:, 0].view(torch.float64) - wl).abs() / wl.abs()).max()) <= 1e-12
    assert torch.equal(l2[:, 1], xvar)
    assert float(oc[0]) == pytest.approx(float((db * db).sum()), rel=1e-13)


def test_scale_numbers_and_qp_bounds_entry_points():
    """two small entry points the Python host reaches only through other shapes (the Julia binding calls them directly):
    pmt_scale_numbers_f64 — scale!(dest, s, y) on number arrays, src/functions.jl:917-925, with the scalar on the device (a Parameter) and
    as an immediate; pmt_qp_bounds_f64 — one constraint block's rows `f(x) in set` turned into l <= a'x <= u for the three cones."""
    import gpu_util as g
    rng = np.random.default_rng(8)
    for n in (1, 255, 70000):
        y = rng.random(n) - 0.5
        dy, ds, out = g.to_dev(y), g.to_dev(np.array([-2.75])), g.empty_f64(n)
        g.call("pmt_scale_numbers_f64", g.ptr(dy), n, g.ptr(ds), 9.0, g.ptr(out), g.stream())         # device scalar wins over the immediate
        assert g.same_bits(g.f64_to_host(out, n), -2.75 * y)
        g.call("pmt_scale_numbers_f64", g.ptr(dy), n, None, 3.5, g.ptr(out), g.stream())
        assert g.same_bits(g.f64_to_host(out, n), 3.5 * y)
    m = 1000
    consts = rng.random(m) - 0.5
    dc = g.to_dev(consts)
    for kind, value in ((0, 0.0), (1, 0.25), (2, -1.5)):                # PMT_SET_EQUAL / GREATER / LESS
        l, u = g.empty_f64(m), g.empty_f64(m)
        g.call("pmt_qp_bounds_f64", g.ptr(dc), m, kind, value, 1e20, g.ptr(l), g.ptr(u), g.stream())
        b = value - consts
        assert g.same_bits(g.f64_to_host(l, m), np.full(m, -1e20) if kind == 2 else b)
        assert g.same_bits(g.f64_to_host(u, m), np.full(m, 1e20) if kind == 1 else b)
