"""host-side cost of the staged C3 step (where does the host spend its time?): python tools/c3_hostprof.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import parametron_jl_amd as P
from parametron_jl_amd import workloads
model, bufs = workloads.config3(pinned=True, handoff="device")
P.solve(model)
ctx = model.device()
def staged():
    model.stage_parameters()
    model.update(synchronize=False)
for _ in range(20): staged()
ctx.synchronize()
ts = te = 0.0
t0 = time.perf_counter()
for _ in range(100):
    a = time.perf_counter(); model.stage_parameters(); b = time.perf_counter(); model.update(synchronize=False); c = time.perf_counter()
    ts += b - a; te += c - b
ctx.synchronize()
print("per step %.3f ms; host time in stage_parameters %.3f ms, in update %.3f ms" % ((time.perf_counter() - t0) * 10, ts * 10, te * 10))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): staged()
pr.disable(); ctx.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
