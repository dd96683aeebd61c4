"""Soak of the multi-workgroup small plan (grid barrier between phases): N solves of a ~100-variable model with host-updated Parameters,
every solve's constraint data and q checked against numpy.  python tools/soak_small_plan.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import parametron_jl_amd as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
bad = 0
for (n, r, m) in ((100, 150, 30), (128, 240, 16), (60, 500, 8)):
    rng = np.random.default_rng(n)
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [P.Variable(model) for _ in range(n)]
    bufs = {"A": np.zeros((r, n), order="F"), "b": np.zeros(r), "G": np.zeros((m, n), order="F"), "h": np.zeros(m)}
    A, b, G, h = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "G", "h"))
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, G * x, "<=", h)
    P.solve(model)
    fz = model.device().fused()
    t0 = time.perf_counter()
    pre = [{k: rng.random(a.shape) for k, a in bufs.items()} for _ in range(7)]
    for it in range(N):
        for k, a in bufs.items():
            a[...] = pre[it % 7][k]
        P.solve(model)
        c = list(model.constraints)[0].f
        f = model.objective.f
        ok = (np.array_equal(c.constants, 0.0 - bufs["h"]) and np.array_equal(c.terms["coeff"].reshape(m, n), bufs["G"])
              and np.allclose(f.affine_terms["coeff"], -2 * bufs["A"].T @ bufs["b"], rtol=1e-12, atol=0))
        if not ok:
            bad += 1
            if bad < 5:
                print("MISMATCH n=%d solve %d" % (n, it), flush=True)
    print("n=%d r=%d m=%d: %d solves checked in %.1f s; fused %s" % (n, r, m, N, time.perf_counter() - t0, fz), flush=True)
    model.close()
print("soak: %d mismatches" % bad)
sys.exit(1 if bad else 0)
