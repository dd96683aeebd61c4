"""config 5 (sparse block pack, device-resident) for a few updates — the workload of a PMC pass: python tools/c5_once.py [updates]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parametron_jl_amd as P
from parametron_jl_amd import workloads
model, Cs = workloads.config5(device_resident=True, handoff="device")
P.solve(model)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    model.update(synchronize=False)
model.device().synchronize()
model.close()
