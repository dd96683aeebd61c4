"""tools: the host-API section of bench.py alone (Model.solve! of config 2 through every hand-off + config 3 through host_csc), with the
per-kernel profile of one host_csc solve — python tools/host_api_bench.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401
import parametron_jl_amd as P  # noqa: E402
import bench_study  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = bench_study.host_api_c2(torch, P, steps)
from parametron_jl_amd import workloads  # noqa: E402
for name, build in (("c2_host_csc", lambda: workloads.config2(handoff="host_csc")), ("c3_host_csc", lambda: workloads.config3(pinned=True, handoff="host_csc")[0])):
    model = build()
    for _ in range(3):
        P.solve(model)
    P.profile_enable(True)
    P.solve(model)
    rep = P.profile_report()
    P.profile_enable(False)
    out["profile_" + name] = {k: round(v["avg_ms"] * v["launches"] * 1e3, 1) for k, v in rep.items()}
    model.close()
print(json.dumps({k: (round(v["ms_per_solve"], 4) if "ms_per_solve" in v else v) for k, v in out.items()}, indent=1))
