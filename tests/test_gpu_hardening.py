"""-m gpu: robustness of the C-ABI library (round-1 review items): plans are independent — two plans driven from two host threads
produce the results of the serial run; the strictly-increasing `xvar` contract of pmt_quad_gram_f64 is checked when the call is
recorded into a plan."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _gram_plan(g, seed, n, r, reps_holder):
    """a plan with its OWN stream that rebuilds the canonical objective of a seeded r x n problem"""
    L = g.lib()
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, None, C.byref(plan))
    rec = C.c_void_p(L.pmt_plan_recording_stream(plan))
    stream = C.c_void_p(L.pmt_plan_stream(plan))
    A, b = g.empty_f64(r * n), g.empty_f64(r)
    g.call("pmt_fill_uniform_f64", g.ptr(A), r * n, seed, 1.0, stream)
    g.call("pmt_fill_uniform_f64", g.ptr(b), r, seed + 1, 1.0, stream)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=g.DEV)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(L.pmt_quad_gram_workspace_bytes(r, n) // 8)
    torch.cuda.synchronize()
    g.call("pmt_plan_begin_record", plan)
    g.call("pmt_quad_gram_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), rec)
    g.call("pmt_plan_end_record", plan)
    keep = (A, b, xvar, ws)
    return plan, (oq, ol, oc), nq, keep


@pytest.mark.parametrize("profiling", [False, True])
def test_two_plans_from_two_threads_are_independent(profiling):
    """profiling=True: every launch of both threads is bracketed by events while they run (the profiler's record list and event pool are
    shared state behind a mutex, its switch an atomic: VERDICT r2 hygiene item)"""
    import gpu_util as g
    n, r, reps = 384, 1024, 40
    plans = [_gram_plan(g, seed, n, r, None) for seed in (11, 23)]
    # serial reference results
    want = []
    for plan, (oq, ol, oc), nq, _ in plans:
        g.call("pmt_plan_update", plan)
        g.call("pmt_plan_synchronize", plan)
        want.append((g.terms_to_host(oq, nq, g.QT).copy(), g.terms_to_host(ol, n, g.LT).copy(), g.f64_to_host(oc, 1).copy()))
        oq.fill_(-7); ol.fill_(-7); oc.fill_(float("nan"))
    torch.cuda.synchronize()                         # the poison fills ran on torch's stream, the plans have their own
    errors = []

    def drive(k):
        try:
            plan = plans[k][0]
            for _ in range(reps):
                g.call("pmt_plan_update", plan)
            g.call("pmt_plan_synchronize", plan)
        except Exception as e:                       # pragma: no cover
            errors.append(e)
    if profiling:
        g.call("pmt_profile_enable", 1)
    threads = [threading.Thread(target=drive, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    if profiling:
        import parametron_jl_amd as P
        rep = P.profile_report()
        g.call("pmt_profile_enable", 0)
        assert rep["gram_mid_kernel"]["launches"] == 2 * reps, rep          # (1024 x 384: the one-launch form of wide shapes, gram_mid.hip)
    for (plan, (oq, ol, oc), nq, _), (wq, wl, wc) in zip(plans, want):
        g.assert_terms_equal(g.terms_to_host(oq, nq, g.QT), wq)
        g.assert_terms_equal(g.terms_to_host(ol, n, g.LT), wl)
        assert g.same_bits(g.f64_to_host(oc, 1), wc)
        g.call("pmt_plan_destroy", plan)


def test_recorded_gram_node_rejects_unsorted_xvar():
    import gpu_util as g
    from parametron_jl_amd import _lib
    L = g.lib()
    n, r = 8, 8
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, None, C.byref(plan))
    rec = C.c_void_p(L.pmt_plan_recording_stream(plan))
    A, b = g.empty_f64(r * n), g.empty_f64(r)
    xvar = torch.tensor([1, 2, 4, 3, 5, 6, 7, 8], dtype=torch.int64, device=g.DEV)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(max(1, L.pmt_quad_gram_workspace_bytes(r, n) // 8))
    torch.cuda.synchronize()
    g.call("pmt_plan_begin_record", plan)
    with pytest.raises(_lib.ArgumentError, match="strictly increasing"):
        g.call("pmt_quad_gram_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), rec)
    assert L.pmt_plan_tape_length(plan) == 0                     # nothing was recorded
    good = torch.arange(1, n + 1, dtype=torch.int64, device=g.DEV)
    g.call("pmt_quad_gram_f64", g.ptr(A), r, r, n, g.ptr(good), g.ptr(b), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), rec)
    g.call("pmt_plan_end_record", plan)
    assert L.pmt_plan_tape_length(plan) == 1
    g.call("pmt_plan_destroy", plan)


@pytest.mark.parametrize("use_graph", [False, True])
def test_side_lane_results_equal_the_single_stream_tape(use_graph):
    """Constraint MOI copies recorded on the plan's side lane (pmt_plan_set_lane; Model.initialize beside a canonical least-squares
    objective) must give the bytes of the single-stream tape, update after update, with the launch tape and with hipGraph replay."""
    import parametron_jl_amd as P
    from parametron_jl_amd import Variable

    def build(side_lane):
        n, r, m = 300, 520, 70
        model = P.Model(P.MockOptimizer(variable_offset=3), quadratic_mode="canonical", use_graph=use_graph, side_lane=side_lane)
        model.SMALL_MODEL_ELEMENTS = 0          # the LARGE-model path (uploads, side lane) at a size that keeps the test fast
        x = [Variable(model) for _ in range(n)]
        rng = np.random.default_rng(77)
        fill = lambda a: a.__setitem__(Ellipsis, rng.random(a.shape) - 0.25)
        A = P.Parameter(fill, np.zeros((r, n)), model)
        b = P.Parameter(fill, np.zeros(r), model)
        Cm = P.Parameter(fill, np.zeros((m, n)), model)
        d = P.Parameter(fill, np.zeros(m), model)
        lo = P.Parameter(fill, np.zeros(n), model)
        residual = A * x - b
        P.objective(model, P.Minimize, P.dot(residual, residual))
        P.constraint(model, Cm * x == d)
        P.constraint(model, x, ">=", lo)
        return model

    a, b = build(True), build(False)
    a.initialize(); b.initialize()
    lanes_a = [r for r in a._records if a._side_lane_ok(r)]
    assert len(lanes_a) == 2 and len(a._lane_records) == 2 and len(b._lane_records) == 0     # both eligible; only model `a` uses the lane
    for _ in range(3):
        a.update(); b.update()
        assert a.objective.f.quadratic_terms.tobytes() == b.objective.f.quadratic_terms.tobytes()
        assert a.objective.f.affine_terms.tobytes() == b.objective.f.affine_terms.tobytes() and a.objective.f.constant == b.objective.f.constant
        for ca, cb in zip(a.constraints, b.constraints):
            assert ca.f.terms.tobytes() == cb.f.terms.tobytes() and np.array_equal(ca.f.constants, cb.f.constants)
    a.close(); b.close()


def test_set_lane_state_and_argument_errors():
    import gpu_util as g
    import parametron_jl_amd as P
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, None, C.byref(plan))
    with pytest.raises(P.ErrorException):
        g.call("pmt_plan_set_lane", plan, 1)                            # not recording
    g.call("pmt_plan_begin_record", plan)
    with pytest.raises(P.ArgumentError):
        g.call("pmt_plan_set_lane", plan, 4)
    g.call("pmt_plan_set_lane", plan, 1)
    g.call("pmt_plan_set_lane", plan, 0)
    g.call("pmt_plan_end_record", plan)
    g.call("pmt_plan_destroy", plan)


@pytest.mark.parametrize("rows,cols,sign", [(1, 1, 0), (7, 300, -1), (130, 257, 1), (512, 1030, -1)])
def test_background_constraint_pack_writes_the_bytes_of_the_tile_kernel(rows, cols, sign):
    """pmt_affine_pack_vector_background_f64 (<= 16 VGPRs, no LDS: co-resident with the contraction on the side lane) against
    pmt_affine_pack_vector_f64 on the same padded-lda input: terms and constants byte for byte."""
    import gpu_util as g
    rng = np.random.default_rng(rows * cols)
    lda = rows + 3
    A = np.zeros((cols, lda)); A[:, :rows] = rng.random((cols, rows)) - 0.5          # column-major with padding rows
    b = rng.random(rows)
    xvar = rng.permutation(cols + 5)[:cols].astype(np.int64) + 1
    varmap = rng.permutation(cols + 5).astype(np.int64) + 1
    dA, db, dx, dv = g.to_dev(A.ravel()), g.to_dev(b), g.to_dev(xvar), g.to_dev(varmap)
    t0, t1 = g.empty_terms(rows * cols, g.VAT), g.empty_terms(rows * cols, g.VAT)
    c0, c1 = g.empty_f64(rows), g.empty_f64(rows)
    for name, t, c in (("pmt_affine_pack_vector_f64", t0, c0), ("pmt_affine_pack_vector_background_f64", t1, c1)):
        g.call(name, g.ptr(dA), lda, rows, cols, g.ptr(dx), g.ptr(db) if sign else None, sign, g.ptr(dv), 11, g.ptr(t), g.ptr(c), g.stream())
    g.assert_terms_equal(g.terms_to_host(t1, rows * cols, g.VAT), g.terms_to_host(t0, rows * cols, g.VAT))
    assert np.array_equal(g.f64_to_host(c1, rows), g.f64_to_host(c0, rows))
