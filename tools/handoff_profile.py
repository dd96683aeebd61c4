"""Per-kernel times and the steady solve! time of config 2 with the device hand-off (Model(..., handoff="device")): python tools/handoff_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parametron_jl_amd as P
n, r, m = 4096, 4096, 512
model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", handoff="device")
x = [P.Variable(model) for _ in range(n)]
A = P.DeviceUniformParameter((r, n), 1, model); b = P.DeviceUniformParameter((r,), 2, model)
Cm = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
res = A * x - b
P.objective(model, P.Minimize, P.dot(res, res)); P.constraint(model, Cm * x == d)
for _ in range(20): P.solve(model)
P.profile_enable(True)
for _ in range(10): P.solve(model)
rep = P.profile_report(); P.profile_enable(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    print("%-40s %3d launches  avg %.4f ms" % (k, v["launches"], v["avg_ms"]))
t0 = time.perf_counter()
for _ in range(20): P.solve(model)
print("solve! %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
