"""-m gpu: edge cases of the C ABI the reference's tests also probe — empty and 1-element inputs, odd (unaligned) shapes,
repeated variables, and the error conventions of the plan layer."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402


def test_empty_inputs_are_noops_and_constants_only_blocks_work():
    import gpu_util as g
    d = g.empty_f64(8)
    canary = g.f64_to_host(d, 8).view(np.int64).copy()
    # zero rows / zero columns: nothing is written, PMT_OK
    g.call("pmt_affine_assemble_f64", g.ptr(d), 0, 0, 5, g.ptr(d), None, 0, g.ptr(d), g.ptr(d), g.stream())
    g.call("pmt_quad_expand_f64", 0, None, 3, None, None, 3, None, 0, None, None, None, g.ptr(d), g.stream())
    assert g.f64_to_host(d, 1)[0] == 0.0                                   # dot of two empty vectors: zero!(dest) -> constant 0
    g.call("pmt_pack_vector_affine_f64", None, None, 0, 0, None, 0, None, g.stream())
    g.call("pmt_sparse_pack_vector_f64", None, None, None, None, 0, None, 0, None, g.stream())
    # a matrix with zero columns still yields the constants 0.0 - b (vecsubtract! on empty affine functions)
    b = np.array([1.5, -0.0, 2.0])
    db, out = g.to_dev(b), g.empty_f64(3)
    g.call("pmt_affine_assemble_f64", None, 3, 3, 0, None, g.ptr(db), -1, None, g.ptr(out), g.stream())
    assert g.same_bits(g.f64_to_host(out, 3), 0.0 - b)
    assert np.array_equal(g.f64_to_host(d, 8).view(np.int64)[1:], canary[1:])


def test_gram_with_no_rows_and_single_variable():
    import gpu_util as g
    n = 5
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    oq, ol, oc = g.empty_terms(n * (n + 1) // 2, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(0, n) // 8)
    A = g.empty_f64(1)
    g.call("pmt_quad_gram_f64", g.ptr(A), 0, 0, n, g.ptr(xvar), None, 0, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    q = g.terms_to_host(oq, n * (n + 1) // 2, g.QT)
    assert np.all(q["coeff"] == 0.0) and q["row"].tolist() == [1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5]
    assert g.f64_to_host(oc, 1)[0] == 0.0
    # n = 1: x^2 * sum a_i^2, doubled on the diagonal in MOI form
    a = O.fill_uniform(7, 3)
    dA, db = g.to_dev(a), g.to_dev(np.zeros(7))
    g.call("pmt_quad_gram_f64", g.ptr(dA), 7, 7, 1, g.ptr(xvar), g.ptr(db), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    assert g.terms_to_host(oq, 1, g.QT)["coeff"][0] == pytest.approx(2 * a @ a, rel=1e-13)


@pytest.mark.parametrize("rows,cols", [(1, 1), (63, 65), (129, 1), (1, 257), (127, 191)])
def test_odd_shapes_take_the_unaligned_paths(rows, cols):
    import gpu_util as g
    A = O.fill_uniform(rows * cols, 9).reshape(cols, rows).T.copy() - 0.5
    b = O.fill_uniform(rows, 10)
    xvar = np.arange(cols, 0, -1, dtype=np.int64)                          # decreasing: indices are data, not positions
    dA, db, dx = g.colmajor(A), g.to_dev(b), g.to_dev(xvar)
    lt, vat, c1, c2 = g.empty_terms(rows * cols, g.LT), g.empty_terms(rows * cols, g.VAT), g.empty_f64(rows), g.empty_f64(rows)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), rows, rows, cols, g.ptr(dx), g.ptr(db), 1, g.ptr(lt), g.ptr(c1), g.stream())
    g.call("pmt_affine_pack_vector_f64", g.ptr(dA), rows, rows, cols, g.ptr(dx), g.ptr(db), 1, None, 0, g.ptr(vat), g.ptr(c2), g.stream())
    ref = O.AffVec(rows).vecadd(O.AffVec(rows).matvecmul_vars(A, xvar), b)
    terms, _, consts = ref.flat()
    g.assert_terms_equal(g.terms_to_host(lt, rows * cols, g.LT), terms)
    g.assert_terms_equal(g.terms_to_host(vat, rows * cols, g.VAT), ref.moi()[0])
    assert g.same_bits(g.f64_to_host(c1, rows), consts) and g.same_bits(g.f64_to_host(c2, rows), consts)
    # literal expansion with an odd number of terms per row exercises the 8-byte lead / tail words of the 16-byte chunk writer
    if rows * cols * cols <= 200000:
        oq, ol, oc = g.empty_terms(rows * cols * cols, g.QT), g.empty_terms(2 * rows * cols, g.LT), g.empty_f64(1)
        g.call("pmt_quad_expand_f64", rows, g.ptr(lt), cols, g.ptr(c1), g.ptr(lt), cols, g.ptr(c1), 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.stream())
        at, qt, const = O.Quad().vecdot_affs_affs(ref, ref).moi()
        g.assert_terms_equal(g.terms_to_host(oq, rows * cols * cols, g.QT), qt)
        g.assert_terms_equal(g.terms_to_host(ol, 2 * rows * cols, g.LT), at)
        assert g.same_bits(g.f64_to_host(oc, 1), [const])


def test_plan_state_errors_map_to_error_exception():
    import gpu_util as g
    from parametron_jl_amd import ErrorException, ArgumentError
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, None, C.byref(plan))
    try:
        rec = C.c_void_p(g.lib().pmt_plan_recording_stream(plan))
        d = g.empty_f64(4)
        with pytest.raises(ErrorException):
            g.call("pmt_consts_f64", g.ptr(d), 4, 1, g.ptr(d), rec)        # recording handle used outside begin/end_record
        g.call("pmt_plan_begin_record", plan)
        with pytest.raises(ErrorException):
            g.call("pmt_plan_begin_record", plan)
        with pytest.raises(ErrorException):
            g.call("pmt_plan_update", plan)                                # still recording
        g.call("pmt_plan_end_record", plan)
        with pytest.raises(ErrorException):
            g.call("pmt_plan_end_record", plan)
        with pytest.raises(ArgumentError):
            g.call("pmt_plan_update", None)
        g.call("pmt_plan_update", plan)                                    # empty tape: fine
    finally:
        g.call("pmt_plan_destroy", plan)
