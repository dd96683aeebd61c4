"""Mid-size least-squares QPs (the sizes of the reference's own use: tens to hundreds of variables) through the host API: device time of
one update!(model) (tape replay, device-side callbacks) and wall time of one solve!(model) with a do-nothing optimizer (host-updated
Parameters change every solve: upload + update! + MOI fetch): python tools/mid_table.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402


def build(n, r_, m, host, **kw):
    rng = np.random.default_rng(n)
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", **kw)
    x = [P.Variable(model) for _ in range(n)]
    bufs = {}
    if host:
        bufs = {"A": np.asfortranarray(rng.random((r_, n))), "b": rng.random(r_), "C": np.asfortranarray(rng.random((m, n))), "d": rng.random(m)}
        A, b, C, d = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "C", "d"))
    else:
        A = P.DeviceUniformParameter((r_, n), 1, model); b = P.DeviceUniformParameter((r_,), 2, model)
        C = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res)); P.constraint(model, C * x, "<=", d)
    P.solve(model)
    return model, bufs, rng


def device_us(model, reps=50, warm=20):
    ctx = model.device()
    for _ in range(warm):
        model.setdirty(); model._run_tape(fetch=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.setdirty(); model._run_tape(fetch=False)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def solve_us(model, bufs, rng, reps=30, warm=8):
    pre = [{k: rng.random(a.shape) for k, a in bufs.items()} for _ in range(2)]
    total = 0.0
    for it in range(warm + reps):
        for k, a in bufs.items():
            a[...] = pre[it & 1][k]                 # (the user's own work: not timed)
        t0 = time.perf_counter()
        P.solve(model)
        if it >= warm:
            total += time.perf_counter() - t0
    return total / reps * 1e6


if __name__ == "__main__":
    shapes = [(50, 80, 10), (100, 150, 30), (128, 1000, 40), (300, 500, 60), (500, 2000, 100), (1000, 1500, 100), (2000, 3000, 200)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for n, r_, m in shapes:
        md, _, _ = build(n, r_, m, host=False, overlap_fetch=False)
        fz = md.device().fused()
        dev = device_us(md)
        md.close()
        mh, bufs, rng = build(n, r_, m, host=True)
        sv = solve_us(mh, bufs, rng)
        small = getattr(mh, "_small", False)
        mh.close()
        print("n=%-5d r=%-5d m=%-4d update! on the device %8.1f us (%d launches / copies per replay)   solve! with host Parameters %9.1f us (%s)" % (
            n, r_, m, dev, fz["exec_length"], sv, "small plan" if small else "uploads + overlapped MOI fetch"), flush=True)
