"""-m gpu: the padded device layout of Parameter matrices (device.padded_lda) changes placement in HBM only — values, indices
and every MOI buffer are identical to the contiguous layout."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import Variable  # noqa: E402
from parametron_jl_amd.device import padded_lda  # noqa: E402
from oracle import oracle as O  # noqa: E402


def test_padded_lda_policy():
    assert padded_lda(4096) == 4160 and padded_lda(512) == 576 and padded_lda(8) == 8 and padded_lda(4000) == 4000 and padded_lda(1024) % 2 == 0
    assert padded_lda(4090) == 4160 and padded_lda(1000) == 1008 and padded_lda(63) == 63 and padded_lda(65) == 80      # zero rows up to a multiple of 16


def test_matrix_fill_places_the_contiguous_stream_with_a_leading_dimension():
    import gpu_util as g
    rows, cols, lda = 37, 11, 44
    d = g.empty_f64(lda * cols)
    g.call("pmt_fill_uniform_matrix_f64", g.ptr(d), rows, cols, lda, C.c_uint64(5), 2.0, g.stream())
    got = g.f64_to_host(d, lda * cols).reshape(cols, lda)[:, :rows]
    assert g.same_bits(np.ascontiguousarray(got).reshape(-1), O.fill_uniform(rows * cols, 5, 2.0))


def test_model_with_512_row_matrices_uses_padded_copies_and_matches_oracle():
    n, r, m = 24, 512, 512                              # 512 rows -> column stride 4 KiB -> padded device copies
    rng = np.random.default_rng(0)
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [Variable(model) for _ in range(n)]
    A = P.Parameter(lambda a: a.__setitem__(slice(None), rng.random(a.shape)), np.zeros((r, n)), model)
    b = P.Parameter(lambda v: v.__setitem__(slice(None), rng.random(r)), np.zeros(r), model)
    Cd = P.DeviceUniformParameter((m, n), 3, model)
    d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, Cd * x == d)
    for _ in range(2):
        P.solve(model)
        assert A._dev.lda == 576 and Cd._dev.lda == 576
        xi = np.arange(1, n + 1, dtype=np.int64)
        w = O.LsqWorkspace(n, r, m)
        w.eval_objective(np.asfortranarray(A()).reshape(-1, order="F"), b(), xi)
        w.objective.canonicalize()
        at, qt, const = w.objective.moi()
        f = model.objective.f
        assert np.array_equal(f.quadratic_terms["row"], qt["row"]) and np.array_equal(f.quadratic_terms["col"], qt["col"])
        np.testing.assert_allclose(f.quadratic_terms["coeff"], qt["coeff"], rtol=1e-12)
        np.testing.assert_allclose(f.affine_terms["coeff"], at["coeff"], rtol=1e-12)
        Ch = O.fill_uniform(m * n, 3 + 1000 * Cd.epoch).reshape(n, m).T
        assert np.array_equal(Cd(), Ch)                                    # fetched back through the pitched copy
        w.eval_constraint(np.ascontiguousarray(Ch.T).reshape(-1), d(), xi)
        ct, cc = w.constraint.moi()
        cf = list(model.constraints)[0].f
        assert np.array_equal(cf.terms.view(np.int64), ct.view(np.int64)) and np.array_equal(cf.constants, cc)
    assert [([(t.coeff, t.var.index) for t in fn.linear], fn.constant) for fn in res()] == \
        O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(A(), xi), b()).as_tuples()


@pytest.mark.parametrize("rows,cols", [(7, 5), (64, 33), (512, 9)])
def test_row_major_column_major_and_page_locked_parameter_values_upload_identically(rows, cols):
    """A Parameter matrix may live in a row-major numpy array (transposed on the device), a column-major one (copied as is) or in
    page-locked memory from Model.parameter_array: the device copy and every MOI buffer are the same bits."""
    rng = np.random.default_rng(rows)
    ref = rng.random((rows, cols)) - 0.5
    results = []
    for kind in ("row-major", "column-major", "pinned"):
        model = P.Model(P.MockOptimizer())
        x = [Variable(model) for _ in range(cols)]
        if kind == "pinned":
            val = model.parameter_array(rows, cols)
            assert val.flags.f_contiguous and val.shape == (rows, cols) and not val.any()
        else:
            val = np.zeros((rows, cols), order="C" if kind == "row-major" else "F")
        val[...] = ref
        A = P.Parameter(model, val=val)                                    # the reference's manual `val=` form (src/parameter.jl:88)
        d = P.Parameter(model, val=np.arange(rows, dtype=np.float64))
        P.constraint(model, A * x == d)
        P.solve(model)
        assert np.array_equal(A._dev.fetch(model.device()), ref)
        val[...] = 2 * ref                                                 # overwrite the buffer between solves, as README Example 2 does
        P.solve(model)
        f = list(model.constraints)[0].f
        results.append((f.terms.copy(), f.constants.copy()))
        assert np.array_equal(f.terms["coeff"].reshape(rows, cols), 2 * ref)
        model.close()
    for t, c in results[1:]:
        assert np.array_equal(t.view(np.int64), results[0][0].view(np.int64)) and np.array_equal(c, results[0][1])


@pytest.mark.parametrize("r,n", [(1000, 40), (70, 130), (4090, 256)])
def test_row_counts_that_are_not_a_multiple_of_16_use_zero_padded_copies(r, n):
    """The Gram objective is given the zero-padded row count (device.row_padded); zero rows change neither 2A'A, 2A'c nor c'c
    (c'c keeps its fixed order: adding 0.0 is the identity), so the result equals numpy on the unpadded data."""
    from parametron_jl_amd.device import row_padded
    from parametron_jl_amd.moi import _gram_rows
    rng = np.random.default_rng(r)
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [Variable(model) for _ in range(n)]
    Av, bv = rng.random((r, n)), rng.random(r)
    A = P.Parameter(model, val=np.asfortranarray(Av))
    b = P.Parameter(model, val=bv)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    for _ in range(2):
        P.solve(model)
        assert A._dev.lda >= row_padded(r) and _gram_rows(model.objective.expr.gram_candidate) == row_padded(r) != r
        f = model.objective.f
        iu = np.triu_indices(n)
        np.testing.assert_allclose(f.quadratic_terms["coeff"], (2 * Av.T @ Av)[iu], rtol=1e-12)
        np.testing.assert_allclose(f.affine_terms["coeff"], -2 * Av.T @ bv, rtol=1e-12)
        # the constant in the library's fixed order for the PADDED shape (the fused tall forms' here:
        # pmt_quad_gram_constant_order); the zero rows add 0.0 * 0.0
        import gpu_util as g
        rp = row_padded(r)
        order, seq = g.constant_in_the_library_order(rp, n, np.concatenate([bv, np.zeros(rp - r)]))
        assert f.constant == seq and order == (2 if n <= 64 else 5)          # <= 64 columns (below 32768 rows): the panel kernel; these wide shapes: the mid-size form
        Av[...] = rng.random((r, n)); A.val[...] = Av; bv[...] = rng.random(r)      # overwrite: the padding must stay zero
