"""What a small-plan launch costs, node by node (run under rocprofv3 --kernel-trace --stats; every case is its own plan and the kernel
trace lists them in order): python tools/small_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parametron_jl_amd import _lib
_lib.require_gpu()
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def dptr(t): return C.c_void_p(t.data_ptr())


def case(name, record, reps=200):
    plan = C.c_void_p()
    _lib.call("pmt_plan_create", 0, stream, C.byref(plan))
    rec = C.c_void_p(_lib.load().pmt_plan_recording_stream(plan))
    _lib.call("pmt_plan_begin_record", plan)
    keep = record(rec)
    _lib.call("pmt_plan_end_record", plan)
    g, n, ln = C.c_int(), C.c_int(), C.c_int64()
    _lib.call("pmt_plan_fused", plan, C.byref(g), C.byref(n), C.byref(ln))
    import time
    for _ in range(20): _lib.call("pmt_plan_update", plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): _lib.call("pmt_plan_update", plan)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    print("%-46s nodes %2d phases %2d launches %d   %.2f us per update (wall, pipelined)" % (name, n.value, _lib.load().pmt_plan_fused_phases(plan), ln.value, dt), flush=True)
    _lib.call("pmt_plan_destroy", plan)
    return keep


def consts_chain(k, dependent, n=8):
    def rec(r):
        bufs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(k + 1)]
        for i in range(k):
            src = bufs[i] if dependent else bufs[0]
            _lib.call("pmt_consts_f64", dptr(src), n, -1, dptr(bufs[i + 1]), r)
        return bufs
    return rec


def fills(k, n=64):
    def rec(r):
        bufs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(k)]
        for i, b in enumerate(bufs):
            _lib.call("pmt_fill_uniform_f64", dptr(b), n, C.c_uint64(i + 1), 1.0, r)
        return bufs
    return rec


def c1(r):
    n, rr, m = 8, 8, 2
    f64, i64 = torch.float64, torch.int64
    A, b, Cm, d = (torch.zeros(k, dtype=f64, device=dev) for k in (rr * n, rr, m * n, m))
    x = torch.arange(1, n + 1, dtype=i64, device=dev)
    res, rc = torch.zeros(rr * n * 2, dtype=i64, device=dev), torch.zeros(rr, dtype=f64, device=dev)
    oq, ol, oc = torch.zeros(rr * n * n * 3, dtype=i64, device=dev), torch.zeros(4 * rr * n, dtype=i64, device=dev), torch.zeros(1, dtype=f64, device=dev)
    vt, vc = torch.zeros(m * n * 3, dtype=i64, device=dev), torch.zeros(m, dtype=f64, device=dev)
    for buf, rows, cols, seed, sc in ((A, rr, n, 1, 1.0), (b, rr, 1, 2, 1.0), (Cm, m, n, 3, 1.0), (d, m, 1, 4, 2.0)):
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(buf), rows, cols, rows, C.c_uint64(seed), sc, r)
    _lib.call("pmt_affine_assemble_f64", dptr(A), rr, rr, n, dptr(x), dptr(b), -1, dptr(res), dptr(rc), r)
    _lib.call("pmt_quad_expand_f64", rr, dptr(res), n, dptr(rc), dptr(res), n, dptr(rc), 1, None, dptr(oq), dptr(ol), dptr(oc), r)
    _lib.call("pmt_affine_pack_vector_f64", dptr(Cm), m, m, n, dptr(x), dptr(d), -1, None, 0, dptr(vt), dptr(vc), r)
    return (A, b, Cm, d, x, res, rc, oq, ol, oc, vt, vc)


keep = []
for k in (2, 8, 32):
    keep.append(case("consts x%d independent" % k, consts_chain(k, False)))
for k in (2, 8, 32):
    keep.append(case("consts x%d dependent chain" % k, consts_chain(k, True)))
keep.append(case("fills x4 (64 each)", fills(4)))
keep.append(case("README Example 1 (7 entries)", c1))
