"""-m gpu: the two ways results leave for the host (pmt_set_host_delivery) and the error path of a staged contraction.

The shipped library prefers the copy engine (hsadma.hip); the kernel-copy fallback (deliver.hip: courier_kernel, to_host_kernel,
to_host_2d_kernel, and the fallback branches of gram.hip) runs where the process's HSA runtime or the device's agent cannot be found.  Both
ways are driven here on the same box through the runtime switch: same tests, same bytes, and the profile says which kernels ran."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402
from gpu_util import DEV, empty_f64, lib, ptr, stream  # noqa: E402
import test_gpu_host_csc as H  # noqa: E402


@pytest.fixture
def delivery():
    def set_mode(mode):
        P.set_host_delivery(mode)
    yield set_mode
    P.set_host_delivery(0)
    _lib.call("pmt_set_fault_injection", 0)


def _profiled(fn):
    P.profile_enable(True)
    try:
        fn()
        return P.profile_report()
    finally:
        P.profile_enable(False)


def test_switch_validates_and_reports(delivery):
    with pytest.raises(P.ArgumentError):
        P.set_host_delivery(3)
    for mode in (2, 1, 0):
        delivery(mode)
        assert P.host_delivery(0)[0] == mode
    assert isinstance(P.host_delivery(0)[1], bool)


@pytest.mark.parametrize("mode", [2, 1])
def test_host_csc_solve_after_solve_in_both_modes(delivery, mode):
    """test_host_arrays_equal_device_handoff_solve_after_solve under the kernel-copy fallback and under the demanded copy engine"""
    if mode == 1 and not P.host_delivery(0)[1]:
        pytest.skip("no copy engine on this box")
    delivery(mode)
    rep = _profiled(lambda: H.test_host_arrays_equal_device_handoff_solve_after_solve(300, 520, 7, True))
    assert ("courier_kernel" in rep) == (mode == 2), sorted(rep)
    assert (("to_host_kernel" in rep) or ("to_host_2d_kernel" in rep)) == (mode == 2), sorted(rep)
    H.test_host_arrays_equal_device_handoff_solve_after_solve(1024, 2048, 64, True)
    H.test_host_arrays_equal_device_handoff_solve_after_solve(96, 80, 4, False)
    H.test_host_delivery_with_host_updated_parameters_and_several_constraint_blocks()


def test_full_size_config2_delivery_through_the_kernel_fallback(delivery):
    delivery(2)
    rep = _profiled(H.test_full_size_config2_host_delivery)
    assert rep["courier_kernel"]["launches"] >= 3


def test_moi_boundary_through_the_kernel_fallback(delivery):
    """the reference's own boundary (quadratic terms delivered row band by row band, MOI buffers as recorded fetches) with kernel copies"""
    delivery(2)
    rep = _profiled(lambda: H.test_overlapped_moi_boundary_equals_the_serial_one(700, 1100, 33))
    assert "courier_kernel" in rep and "to_host_kernel" in rep
    H.test_deliver_quadratic_terms_matches_plain_node(777, 1000, 16)
    H.test_deliver_quadratic_terms_matches_plain_node(4096, 2048, 0)
    H.test_deliver_entry_point_matches_plain_csc(2048, 1408, 0)


def test_both_modes_deliver_the_same_bytes(delivery):
    """same model, same seeds: the host arrays after three solves are identical whichever way they travelled"""
    if not P.host_delivery(0)[1]:
        pytest.skip("no copy engine on this box")
    got = {}
    for mode in (1, 2):
        delivery(mode)
        model = H.lsq_model(1024, 2048, 64, handoff="host_csc")
        for _ in range(3):
            P.solve(model)
        h = model.device_qp.host.as_dict()
        got[mode] = {k: (h[k][0].copy() if isinstance(h[k], tuple) else np.array(h[k]).copy()) for k in ("P", "A", "q", "l", "u")}
        model.close()
    for k in got[1]:
        assert np.array_equal(got[1][k], got[2][k]), k


def test_recorded_fetch_into_pageable_memory_takes_the_runtime_copy(delivery):
    """the copy engine is handed physical pages: a pageable destination (a plain numpy array) must not reach it — it takes the runtime's
    staged copy and still lands; demanding the engine for it is an error, not a fault"""
    L = lib()
    n = 1 << 16
    src = torch.arange(n, dtype=torch.float64, device=DEV)
    plan = C.c_void_p()
    _lib.call("pmt_plan_create", 0, None, C.byref(plan))
    pageable = np.full(n, -1.0)
    _lib.call("pmt_plan_begin_record", plan)
    _lib.call("pmt_plan_record_fetch", plan, pageable.ctypes.data_as(C.c_void_p), ptr(src), 8 * n)
    _lib.call("pmt_plan_end_record", plan)
    for _ in range(2):
        pageable[:] = -1.0
        _lib.call("pmt_plan_update", plan)
        _lib.call("pmt_plan_fetch_synchronize", plan)
        _lib.call("pmt_plan_synchronize", plan)
        assert np.array_equal(pageable, np.arange(n, dtype=np.float64))
    delivery(1)
    with pytest.raises(P.ErrorException):
        _lib.call("pmt_plan_update", plan)
    delivery(0)
    _lib.call("pmt_plan_destroy", plan)


@pytest.mark.parametrize("mode", [0, 2])
def test_pitched_recorded_fetch(delivery, mode):
    """pmt_plan_record_fetch_2d: the columns of a padded device matrix into a row range of every column of a taller host matrix (how the CSC
    values of a dense constraint block leave, straight out of the Parameter buffer)"""
    delivery(mode)
    rows, cols, lda, top, tall = 40, 33, 48, 7, 64                         # block rows, columns, device pitch, first row on the host, host pitch
    dev = torch.arange(lda * cols, dtype=torch.float64, device=DEV)
    hp = C.c_void_p()
    _lib.call("pmt_host_alloc", 8 * tall * cols, C.byref(hp))
    host = np.frombuffer((C.c_char * (8 * tall * cols)).from_address(hp.value), dtype=np.float64).reshape(cols, tall)
    plan = C.c_void_p()
    _lib.call("pmt_plan_create", 0, None, C.byref(plan))
    _lib.call("pmt_plan_begin_record", plan)
    _lib.call("pmt_plan_record_fetch_2d", plan, C.c_void_p(hp.value + 8 * top), 8 * tall, ptr(dev), 8 * lda, 8 * rows, cols)
    _lib.call("pmt_plan_end_record", plan)
    want = np.full((cols, tall), -1.0)
    want[:, top:top + rows] = np.arange(lda * cols, dtype=np.float64).reshape(cols, lda)[:, :rows]
    for rep in range(3):
        host[:] = -1.0
        _lib.call("pmt_plan_update", plan)
        _lib.call("pmt_plan_fetch_synchronize", plan)
        assert np.array_equal(host, want)
    _lib.call("pmt_plan_destroy", plan)
    _lib.call("pmt_host_free", hp)


@pytest.mark.parametrize("mode", [0, 2])
def test_pair_fold_timeout_is_reported_not_delivered_as_numbers(delivery, mode):
    """A second-half workgroup of a split tile whose partner never announces itself (fault injection) writes the tile as NaN AND raises the
    error word: pmt_fetch_synchronize returns PMT_HIP_ERROR instead of PMT_OK with NaNs in P.  The next delivery (injection off) is clean."""
    delivery(mode)
    L = lib()
    s = stream()
    rows = cols = 4096                                             # stages of 128 tiles split in two: the pair fold is what runs (gram_sk.hip)
    A = torch.empty(rows * cols, dtype=torch.float64, device=DEV)
    b = torch.empty(rows, dtype=torch.float64, device=DEV)
    _lib.call("pmt_fill_uniform_f64", ptr(A), rows * cols, 21, 1.0, s)
    _lib.call("pmt_fill_uniform_f64", ptr(b), rows, 22, 1.0, s)
    xvar = torch.arange(1, cols + 1, dtype=torch.int64, device=DEV)
    nnz = cols * (cols + 1) // 2
    ws = torch.empty(max(1, L.pmt_quad_gram_workspace_bytes(rows, cols) // 8), dtype=torch.float64, device=DEV)
    P1, lin, c = empty_f64(nnz), torch.empty(2 * cols, dtype=torch.int64, device=DEV), empty_f64(1)
    hp = C.c_void_p()
    _lib.call("pmt_host_alloc", 8 * nnz, C.byref(hp))
    host = np.frombuffer((C.c_char * (8 * nnz)).from_address(hp.value), dtype=np.float64)

    def deliver():
        host[:] = -1.0
        _lib.call("pmt_quad_gram_csc_deliver_f64", ptr(A), rows, rows, cols, ptr(xvar), ptr(b), -1, None, 1.0, ptr(P1), hp, 0, ptr(lin), ptr(c), ptr(ws), s)
        _lib.call("pmt_fetch_synchronize", s)

    deliver()
    torch.cuda.synchronize()
    good = host.copy()
    assert np.all(np.isfinite(good)) and np.array_equal(good, P1.cpu().numpy())
    _lib.call("pmt_set_fault_injection", 1)
    with pytest.raises(_lib.HipError, match="pair fold"):
        deliver()
    torch.cuda.synchronize()
    assert np.isnan(host).any() and np.isnan(P1.cpu().numpy()).any()           # marked, never a plausible half sum
    _lib.call("pmt_set_fault_injection", 0)
    deliver()
    torch.cuda.synchronize()
    assert np.array_equal(host, good)
    _lib.call("pmt_host_free", hp)


def test_pair_fold_timeout_surfaces_from_solve(delivery):
    from parametron_jl_amd import workloads
    model = workloads.config2(handoff="host_csc")
    P.solve(model)
    _lib.call("pmt_set_fault_injection", 1)
    with pytest.raises(_lib.HipError, match="pair fold"):
        P.solve(model)
    _lib.call("pmt_set_fault_injection", 0)
    P.solve(model)
    H.assert_host_equals_device(model)
    model.close()


@pytest.mark.parametrize("mode", [0, 2])
def test_back_to_back_immediate_deliveries_do_not_overtake_each_other(delivery, mode):
    """two unrecorded pmt_quad_gram_csc_deliver_f64 calls on one stream with no synchronisation in between, different inputs, the same device
    and host arrays: the second call waits by itself until the first delivery has read out_P_values (the header's promise — on the copy-engine
    path the signals of an immediate call are fresh, so it must wait for what earlier calls handed to the engine); afterwards the host array
    holds the SECOND result, bit for bit the device array"""
    delivery(mode)
    L = lib()
    s = stream()
    rows, cols = 2048, 1408
    nnz = cols * (cols + 1) // 2
    A1 = torch.empty(rows * cols, dtype=torch.float64, device=DEV)
    A2 = torch.empty(rows * cols, dtype=torch.float64, device=DEV)
    b = torch.empty(rows, dtype=torch.float64, device=DEV)
    _lib.call("pmt_fill_uniform_f64", ptr(A1), rows * cols, 31, 1.0, s)
    _lib.call("pmt_fill_uniform_f64", ptr(A2), rows * cols, 32, 3.0, s)
    _lib.call("pmt_fill_uniform_f64", ptr(b), rows, 33, 1.0, s)
    xvar = torch.arange(1, cols + 1, dtype=torch.int64, device=DEV)
    ws = torch.empty(max(1, L.pmt_quad_gram_workspace_bytes(rows, cols) // 8), dtype=torch.float64, device=DEV)
    Pv, lin, c = empty_f64(nnz), torch.empty(2 * cols, dtype=torch.int64, device=DEV), empty_f64(1)
    ref = empty_f64(nnz)
    _lib.call("pmt_quad_gram_csc_f64", ptr(A2), rows, rows, cols, ptr(xvar), ptr(b), -1, None, 1.0, ptr(ref), None, ptr(lin), ptr(c), ptr(ws), s)
    torch.cuda.synchronize()
    want = ref.cpu().numpy()
    hp = C.c_void_p()
    _lib.call("pmt_host_alloc", 8 * nnz, C.byref(hp))
    host = np.frombuffer((C.c_char * (8 * nnz)).from_address(hp.value), dtype=np.float64)
    for rep in range(3):
        host[:] = -1.0
        for A in (A1, A2):
            _lib.call("pmt_quad_gram_csc_deliver_f64", ptr(A), rows, rows, cols, ptr(xvar), ptr(b), -1, None, 1.0, ptr(Pv), hp, 0, ptr(lin), ptr(c), ptr(ws), s)
        _lib.call("pmt_fetch_synchronize", s)
        torch.cuda.synchronize()
        dev = Pv.cpu().numpy()
        assert np.array_equal(host, dev)
        np.testing.assert_allclose(host, want, rtol=1e-13, atol=0)          # (staged: split tiles are two half sums)
    _lib.call("pmt_host_free", hp)


@pytest.mark.parametrize("mode", [0, 2])
def test_lane_3_transfer_reads_what_the_side_stream_produced(delivery, mode):
    """C ABI: a value regenerated ON the side stream (pmt_plan_lane_stream) and a recorded pitched fetch of it at the front of the side lane
    without the fork from the plan's stream (lane 3) — while a long fill occupies the plan's stream.  Update after update the host block
    holds THIS update's values (the side stream orders callback -> transfer; the previous transfer is drained by the host before the
    next callback, as both hosts do), and the tape's main-lane entry sees the plan's stream's own work."""
    from oracle import oracle as O
    delivery(mode)
    rows, cols, lda, tall = 64, 50, 80, 96
    plan = C.c_void_p()
    _lib.call("pmt_plan_create", 0, None, C.byref(plan))
    side = C.c_void_p()
    _lib.call("pmt_plan_lane_stream", plan, 1, C.byref(side))
    main = C.c_void_p(_lib.load().pmt_plan_stream(plan))
    assert side.value and side.value != main.value
    dev = torch.zeros(lda * cols, dtype=torch.float64, device=DEV)
    big = torch.empty(1 << 24, dtype=torch.float64, device=DEV)              # 128 MB: the "objective's callback" on the plan's stream
    small = torch.empty(1000, dtype=torch.float64, device=DEV)
    out = torch.empty(1000, dtype=torch.float64, device=DEV)
    hp = C.c_void_p()
    _lib.call("pmt_host_alloc", 8 * tall * cols, C.byref(hp))
    host = np.frombuffer((C.c_char * (8 * tall * cols)).from_address(hp.value), dtype=np.float64).reshape(cols, tall)
    rec = C.c_void_p(_lib.load().pmt_plan_recording_stream(plan))
    _lib.call("pmt_plan_begin_record", plan)
    _lib.call("pmt_scale_numbers_f64", ptr(small), 1000, None, 2.0, ptr(out), rec)          # main lane: reads what the plan's stream wrote
    _lib.call("pmt_plan_set_lane", plan, 3)
    _lib.call("pmt_plan_record_fetch_2d", plan, C.c_void_p(hp.value + 8 * 5), 8 * tall, ptr(dev), 8 * lda, 8 * rows, cols)
    _lib.call("pmt_plan_set_lane", plan, 0)
    _lib.call("pmt_plan_end_record", plan)
    torch.cuda.synchronize()
    for it in range(4):
        host[:] = -1.0
        _lib.call("pmt_plan_fetch_synchronize", plan)                                                  # (the previous transfer has read `dev`)
        _lib.call("pmt_fill_uniform_f64", ptr(big), 1 << 24, C.c_uint64(100 + it), 1.0, main)        # plan's stream: long
        _lib.call("pmt_fill_uniform_f64", ptr(small), 1000, C.c_uint64(200 + it), 1.0, main)
        _lib.call("pmt_fill_uniform_matrix_f64", ptr(dev), rows, cols, lda, C.c_uint64(300 + it), 1.0, side)   # side stream: this update's value
        _lib.call("pmt_plan_update", plan)
        _lib.call("pmt_plan_fetch_synchronize", plan)
        want = O.fill_uniform(rows * cols, 300 + it).reshape(cols, rows)
        assert np.array_equal(host[:, 5:5 + rows], want) and np.all(host[:, :5] == -1.0) and np.all(host[:, 5 + rows:] == -1.0)
        _lib.call("pmt_plan_synchronize", plan)
        assert np.array_equal(out.cpu().numpy(), 2.0 * O.fill_uniform(1000, 200 + it))
    _lib.call("pmt_plan_destroy", plan)
    _lib.call("pmt_host_free", hp)
