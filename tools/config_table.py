"""Device time of one update!(model) (tape replay, Parameter values resident or just uploaded, no MOI fetch: overlap_fetch=False keeps the
MOI buffers' recorded fetches out of the tape) for the BASELINE configurations, built through the host API: python tools/config_table.py"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402


def device_ms(model, reps=20, warm=25):
    ctx = model.device()
    for _ in range(warm):
        model.setdirty(); model._run_tape(fetch=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.setdirty(); model._run_tape(fetch=False)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def report(name, model, extra=""):
    P.solve(model)
    ms = device_ms(model)
    extra = extra % model.device().tape_length() if "%d" in extra else extra
    fz = model.device().fused()
    if fz["groups"]:
        extra += "  small plan: %d entries in %d fused run(s), %d launches / copies per replay" % (fz["nodes"], fz["groups"], fz["exec_length"])
    print("%-64s %9.4f ms  %10.1f re-evaluations/s  %s" % (name, ms, 1e3 / ms, extra), flush=True)


def config1(mode, graph=False):
    n, m = 8, 2
    model = P.Model(P.MockOptimizer(), quadratic_mode=mode, use_graph=graph, overlap_fetch=False)
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((n, n), 1, model); b = P.DeviceUniformParameter((n,), 2, model)
    C = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    r = A * x - b
    P.objective(model, P.Minimize, P.dot(r, r)); P.constraint(model, C * x == d)
    report("C1 README Example 1 (n=8, m=2), %s objective%s" % (mode, ", hipGraph replay" if graph else ""), model, "launch-latency bound (%d tape entries)")


def config2():
    n, r_, m = 4096, 4096, 512
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", overlap_fetch=False)
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r_, n), 1, model); b = P.DeviceUniformParameter((r_,), 2, model)
    C = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    r = A * x - b
    P.objective(model, P.Minimize, P.dot(r, r)); P.constraint(model, C * x == d)
    report("C2 dense LSQ n=4096, m=512 equalities (host API; bench.py is the C-ABI line)", model)


def config3():
    n, r_, mi = 4096, 4096, 512
    rng = np.random.default_rng(5)
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", overlap_fetch=False)
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r_, n), 1, model); b = P.DeviceUniformParameter((r_,), 2, model)
    G = P.Parameter(model, val=np.asfortranarray(rng.random((mi, n)))); h = P.Parameter(model, val=rng.random(mi))
    l = P.Parameter(model, val=-rng.random(n)); u = P.Parameter(model, val=rng.random(n))
    r = A * x - b
    P.objective(model, P.Minimize, P.dot(r, r))
    P.constraint(model, G * x, "<=", h); P.constraint(model, x, ">=", l); P.constraint(model, x, "<=", u)
    report("C3 same + 512 inequalities + bounds, val= Parameters (17 MB uploaded per update)", model)


def config5():
    m, n = 4096, 16384
    rng = np.random.default_rng(3)
    k = int(0.05 * m)
    indptr = np.arange(0, (n + 1) * k, k, dtype=np.int64)
    indices = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for _ in range(n)]).astype(np.int64)
    Cs = sp.csc_matrix((rng.random(indices.size) + 0.1, indices, indptr), shape=(m, n))
    model = P.Model(P.MockOptimizer(), overlap_fetch=False)
    x = [P.Variable(model) for _ in range(n)]
    Cp = P.Parameter(model, val=Cs); d = P.Parameter(model, val=rng.random(m))
    P.constraint(model, Cp * x == d)
    report("C5 sparse C (5 %%, %d non-zeros), n=16384, m=4096 (27 MB of nzval uploaded)" % Cs.nnz, model)


if __name__ == "__main__":
    config1("literal"); config1("literal", graph=True); config1("canonical"); config2(); config3(); config5()
