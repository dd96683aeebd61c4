"""Soak test of the host delivery: thousands of solves, the host arrays compared with the device hand-off after every one of them.
    python tools/soak_host_delivery.py [solves, default 2000]
Covers the copy engine and the kernel-copy transport, the pitched transfers out of Parameter buffers, host-side block copies with staged
uploads, the front of the side lane and the side-stream Parameter callbacks — the places where a rare ordering bug would hide."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import parametron_jl_amd as P  # noqa: E402
import test_gpu_host_csc as H  # noqa: E402
from parametron_jl_amd import workloads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0


def check(model, what, it):
    global bad
    try:
        H.assert_host_equals_device(model)
    except AssertionError as e:
        bad += 1
        print("MISMATCH %s solve %d: %s" % (what, it, str(e)[:200]), flush=True)


for mode in (0, 2):
    P.set_host_delivery(mode)
    t0 = time.perf_counter()
    model = H.lsq_model(1024, 2048, 64, handoff="host_csc")
    for it in range(N):
        P.solve(model)
        check(model, "mid-size mode %d" % mode, it)
    model.close()
    print("mode %d: %d solves of n = 1024, r = 2048, m = 64 checked in %.1f s" % (mode, N, time.perf_counter() - t0), flush=True)
P.set_host_delivery(0)

t0 = time.perf_counter()
model = workloads.config2(handoff="host_csc")
for it in range(max(50, N // 5)):
    P.solve(model)
    if it % 10 == 0:
        check(model, "config 2", it)
model.close()
print("config 2: %d solves, every tenth checked, %.1f s" % (max(50, N // 5), time.perf_counter() - t0), flush=True)

# config-3 shape: host-updated constraint data rewritten before every solve, staged uploads on alternating slots, host-side block copy
t0 = time.perf_counter()
rng = np.random.default_rng(9)
n, r, mi = 512, 768, 96
model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", handoff="host_csc")
x = [P.Variable(model) for _ in range(n)]
A = P.DeviceUniformParameter((r, n), 1, model)
b = P.DeviceUniformParameter((r,), 2, model)
res = A * x - b
P.objective(model, P.Minimize, P.dot(res, res))
bufs = {"G": model.parameter_array(mi, n), "h": model.parameter_array(mi), "l": model.parameter_array(n), "u": model.parameter_array(n)}
G, h, lo, up = (P.Parameter(model, val=bufs[k]) for k in ("G", "h", "l", "u"))
P.constraint(model, G * x, "<=", h)
P.constraint(model, x, ">=", lo)
P.constraint(model, x, "<=", up)
P.solve(model)
import scipy.sparse as sp  # noqa: E402
for it in range(N):
    for v in bufs.values():
        v[...] = rng.random(v.shape)
    if it % 3:
        model.stage_parameters()
    P.solve(model)
    check(model, "config-3 shape", it)
    if it % 50 == 0:
        host = model.device_qp.host.as_dict()
        Ad = sp.csc_matrix(host["A"], shape=(mi + 2 * n, n)).toarray()
        if not (np.array_equal(Ad[n:n + mi], bufs["G"]) and np.array_equal(host["u"][n:n + mi], bufs["h"])):
            bad += 1
            print("MISMATCH config-3 shape solve %d: A / u do not follow the host buffers" % it, flush=True)
model.close()
print("config-3 shape: %d solves checked in %.1f s" % (N, time.perf_counter() - t0), flush=True)
# a SMALL model (one-launch plan): host-updated Parameters through mailboxes, MOI buffers stored by the kernels straight into the function
# objects' page-locked arrays — every solve's results must be THIS solve's values (a stale mailbox / an early read shows up at once)
t0 = time.perf_counter()
n, r, m = 60, 90, 12
model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
x = [P.Variable(model) for _ in range(n)]
bufs = {"A": np.zeros((r, n), order="F"), "b": np.zeros(r), "G": np.zeros((m, n), order="F"), "h": np.zeros(m), "l": np.zeros(n)}
A, b, G, h, lo = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "G", "h", "l"))
res = A * x - b
P.objective(model, P.Minimize, P.dot(res, res))
P.constraint(model, G * x, "<=", h)
P.constraint(model, x, ">=", lo)
P.solve(model)
assert getattr(model, "_small", False)
for it in range(N):
    for v in bufs.values():
        v[...] = rng.random(v.shape)
    P.solve(model)
    f = model.objective.f
    cs = sorted(model.constraints, key=lambda c: len(c.f.constants))        # (G x <= h: 12 rows, x >= l: 60 rows)
    oks = (np.allclose(f.affine_terms["coeff"], -2 * bufs["A"].T @ bufs["b"], rtol=1e-12, atol=0), abs(f.constant - bufs["b"] @ bufs["b"]) <= 1e-12 * f.constant,
           np.array_equal(cs[0].f.constants, 0.0 - bufs["h"]), np.array_equal(cs[1].f.constants, 0.0 - bufs["l"]),
           np.array_equal(np.sort(cs[0].f.terms["coeff"]), np.sort(bufs["G"].reshape(-1))))
    if not all(oks):
        bad += 1
        if bad < 6:
            print("MISMATCH small model solve %d: q %s const %s -h %s -l %s G %s" % ((it,) + oks), flush=True)
model.close()
print("small model: %d solves checked in %.1f s" % (N, time.perf_counter() - t0), flush=True)
print("soak: %d mismatches" % bad, flush=True)
sys.exit(1 if bad else 0)
