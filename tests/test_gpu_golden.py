"""-m gpu: the HIP path against the committed, independently generated golden vectors of BASELINE config 1
(tests/golden/c1_readme_example1.npz, written from SURVEY.md Appendix A by tests/golden/make_c1_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_device_reproduces_the_independent_config1_golden_vectors():
    import gpu_util as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_readme_example1.npz"))
    n, r, m = int(gold["n"]), int(gold["r"]), int(gold["m"])
    dA, db, dC, dd, dQ = (g.to_dev(gold[k]) for k in ("A", "b", "C", "d", "Q"))      # column-major already
    seed_A = g.empty_f64(r * n)
    g.call("pmt_fill_uniform_f64", g.ptr(seed_A), r * n, C.c_uint64(1), 1.0, g.stream())
    assert g.same_bits(g.f64_to_host(seed_A, r * n), gold["A"])                        # the device generator is the same stream
    xvar, vm = g.to_dev(np.arange(1, n + 1, dtype=np.int64)), g.to_dev(gold["varmap"])
    # A.1 residual
    lt, cst = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(lt), g.ptr(cst), g.stream())
    t = g.terms_to_host(lt, r * n, g.LT)
    assert g.same_bits(t["coeff"].reshape(r, n), gold["residual_coeff"]) and np.array_equal(t["var"].reshape(r, n), gold["residual_var"])
    assert g.same_bits(g.f64_to_host(cst, r), gold["residual_const"])
    # A.2 literal objective, MOI copy fused
    oq, ol, oc = g.empty_terms(r * n * n, g.QT), g.empty_terms(2 * r * n, g.LT), g.empty_f64(1)
    g.call("pmt_quad_expand_f64", r, g.ptr(lt), n, g.ptr(cst), g.ptr(lt), n, g.ptr(cst), 1, g.ptr(vm), g.ptr(oq), g.ptr(ol), g.ptr(oc), g.stream())
    g.assert_terms_equal(g.terms_to_host(oq, r * n * n, g.QT), gold["literal_quad"])
    g.assert_terms_equal(g.terms_to_host(ol, 2 * r * n, g.LT), gold["literal_aff"])
    assert g.same_bits(g.f64_to_host(oc, 1), gold["const"])
    # A.3 canonical objective on the matrix cores
    nq = n * (n + 1) // 2
    cq, cl, cc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(r, n) // 8)
    g.call("pmt_quad_gram_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, 1, g.ptr(vm), g.ptr(cq), g.ptr(cl), g.ptr(cc), g.ptr(ws), g.stream())
    q = g.terms_to_host(cq, nq, g.QT)
    assert np.array_equal(q["row"], gold["canonical_quad"]["row"]) and np.array_equal(q["col"], gold["canonical_quad"]["col"])
    np.testing.assert_allclose(q["coeff"], gold["canonical_quad"]["coeff"], rtol=1e-12, atol=0)
    l = g.terms_to_host(cl, n, g.LT)
    assert np.array_equal(l["var"], gold["canonical_aff"]["var"])
    np.testing.assert_allclose(l["coeff"], gold["canonical_aff"]["coeff"], rtol=1e-12, atol=0)
    assert g.same_bits(g.f64_to_host(cc, 1), gold["const"])
    # A.4 constraint block, A.5 bounds
    vt, vc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dC), m, m, n, g.ptr(xvar), g.ptr(dd), -1, g.ptr(vm), 0, g.ptr(vt), g.ptr(vc), g.stream())
    g.assert_terms_equal(g.terms_to_host(vt, m * n, g.VAT), gold["constraint_terms"])
    assert g.same_bits(g.f64_to_host(vc, m), gold["constraint_consts"])
    bt, bc, dlows = g.empty_terms(n, g.VAT), g.empty_f64(n), g.to_dev(gold["lows"])
    g.call("pmt_vars_addsub_f64", g.ptr(xvar), n, g.ptr(dlows), -1, g.ptr(vm), 0, None, g.ptr(bt), g.ptr(bc), g.stream())
    g.assert_terms_equal(g.terms_to_host(bt, n, g.VAT), gold["bounds_terms"])
    assert g.same_bits(g.f64_to_host(bc, n), gold["bounds_consts"])
    # A.6 bilinear x'Qx
    bq = g.empty_terms(n * n, g.QT)
    g.call("pmt_bilinear_f64", g.ptr(dQ), n, n, n, g.ptr(xvar), g.ptr(xvar), 1, g.ptr(vm), g.ptr(bq), g.stream())
    g.assert_terms_equal(g.terms_to_host(bq, n * n, g.QT), gold["bilinear_quad"])
