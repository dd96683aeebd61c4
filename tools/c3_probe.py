"""Config 3 step time, launch tape vs hipGraph replay, serial vs staged uploads: python tools/c3_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import workloads  # noqa: E402


def main():
    for use_graph in (False, True):
        model, bufs = workloads.config3(pinned=True, handoff="device", use_graph=use_graph)
        P.solve(model)
        ctx = model.device()

        def staged():
            model.stage_parameters()
            model.update(synchronize=False)

        def serial():
            model.update(synchronize=False)
        for name, fn in (("serial", serial), ("staged", staged)):
            for _ in range(15):
                fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 100
            model.wait_staged()
            print("use_graph=%s %s: %.4f ms/step" % (use_graph, name, dt * 1e3), flush=True)
        print("tape length", ctx.tape_length() if hasattr(ctx, "tape_length") else "?")
        model.close()


if __name__ == "__main__":
    main()
