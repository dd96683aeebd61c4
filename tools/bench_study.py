#!/usr/bin/env python
"""tools/bench_study.py — the longer studies that used to ride on bench.py's line: the objective node over its shape
table, solve! wall time of mid-size models, solve! through every hand-off, the constraint pack inside the step by three clocks.
    python tools/bench_study.py [tall] [mid] [host_api] [pack] [--out profiles/rNN_study.json]
Writes one JSON file (default gpurun_out/bench_study.json) and prints a compact table on stdout."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import (C2Workload, F64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, dptr, guarded, hbm_roofline, profile_report, timed_loop)  # noqa: E402
from bench_configs import config_tall, constraint_pack_microbench  # noqa: E402


MID_SHAPES = ((50, 80, 10), (100, 150, 30), (128, 240, 16), (300, 500, 60))


def config_mid(torch, P, steps=200):
    """Mid-size least-squares QPs — the sizes the reference is used at — through the host API with HOST-updated Parameters that change every
    solve (Parameter(model, val=buf), src/parameter.jl:88) and a do-nothing optimizer: wall time of one solve!(model) = mailboxes in, update!
    (one or two launches + the objective's node), the MOI buffers on the host (stored by the kernels into page-locked arrays), MOI.set calls.
    The user's own refill of the buffers is not timed."""
    import numpy as np
    out = {}
    for n, r_, m in MID_SHAPES:
        rng = np.random.default_rng(n)
        model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
        x = [P.Variable(model) for _ in range(n)]
        bufs = {"A": np.asfortranarray(rng.random((r_, n))), "b": rng.random(r_), "C": np.asfortranarray(rng.random((m, n))), "d": rng.random(m)}
        A, b, Cm, d = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "C", "d"))
        res = A * x - b
        P.objective(model, P.Minimize, P.dot(res, res)); P.constraint(model, Cm * x, "<=", d)
        P.solve(model)
        pre = [{k: rng.random(a.shape) for k, a in bufs.items()} for _ in range(2)]
        total = 0.0
        for it in range(20 + steps):
            for k, a in bufs.items():
                a[...] = pre[it & 1][k]
            t0 = time.perf_counter()
            P.solve(model)
            if it >= 20:
                total += time.perf_counter() - t0
        fz = model.device().fused()
        out["n%d_r%d_m%d" % (n, r_, m)] = {"solve_us": total / steps * 1e6, "small_model_path": bool(getattr(model, "_small", False)),
                                           "launches_per_update": fz["exec_length"], "run_workgroups": fz.get("workgroups")}
        model.close()
    return out


def host_api_c2(torch, P, steps):
    """Model.solve!() of config 2 through the host API, PCIe included where it occurs (never `value`):
      handoff_device           CSC QP data left in HBM (nothing crosses PCIe)
      handoff_host_csc         what a host OSQP's update takes (P.x 67.1 MB, A.x 16.8 MB, q, l, u = 84 MB) in page-locked host arrays,
                               shipped WHILE the re-evaluation runs: recorded fetches + band-wise delivery of P out of the contraction
      handoff_host_csc_serial  the same 84 MB fetched behind the re-evaluation
      handoff_moi              the reference's boundary: 252 MB of MOI term arrays in page-locked host buffers, shipped WHILE the re-evaluation
                               runs (recorded fetches; the objective's quadratic terms row band by row band, pmt_quad_gram_deliver_f64)
      handoff_moi_serial       the same 252 MB fetched behind the re-evaluation
    Every solve! ends with the host holding the data (synchronised); Parameters are regenerated on the device before each one."""
    from parametron_jl_amd import workloads
    out = {}
    what = {"device": "CSC QP data left in HBM", "moi": "MOI term arrays fetched to the host (252 MB over PCIe)",
            "host_csc": "CSC values of P and A, q, l, u (84 MB) delivered to page-locked host arrays while the contraction runs",
            "host_csc_serial": "the same 84 MB fetched behind the re-evaluation"}
    what["moi_serial"] = "the same 252 MB fetched behind the re-evaluation"
    what["moi"] = "MOI term arrays (252 MB) delivered to page-locked host buffers while the contraction runs: recorded fetches + the quadratic terms row band by row band"
    for name in ("device", "host_csc", "host_csc_serial", "moi", "moi_serial"):
        kw = {"handoff": "host_csc", "overlap_fetch": name == "host_csc"} if name.startswith("host_csc") else \
            ({"handoff": "moi", "overlap_fetch": name == "moi"} if name.startswith("moi") else {"handoff": name})
        model = workloads.config2(**kw)
        P.solve(model)
        for _ in range(5):
            P.solve(model)
        k = max(3, min(steps, 20))
        t0 = time.perf_counter()
        for _ in range(k):
            P.solve(model)
        dt = (time.perf_counter() - t0) / k
        out["handoff_" + name] = {"ms_per_solve": dt * 1e3, "solves_per_s": 1.0 / dt, "what": "Parameters regenerated on the device, " + what[name]}
        if name.startswith("host_csc"):
            nb = model.device_qp.host.nbytes()
            out["handoff_" + name]["bytes_to_host"] = nb
            out["handoff_" + name]["pcie_floor_ms"] = nb / 54e9 * 1e3      # 54 GB/s: the page-locked D2H rate of this box (tools/deliver_probe.hip)
        model.close()
    # config 3 end to end for a host solver: G, h, l, u rewritten by the host before every solve (17 MB up, staged), objective + three constraint
    # blocks re-evaluated, P / A / q / l / u (84 MB) delivered to the host while the contraction runs
    model, bufs = workloads.config3(pinned=True, handoff="host_csc")
    P.solve(model)

    def c3_solve():
        model.stage_parameters()
        P.solve(model)
    for _ in range(5):
        c3_solve()
    k = max(3, min(steps, 20))
    t0 = time.perf_counter()
    for _ in range(k):
        c3_solve()
    dt = (time.perf_counter() - t0) / k
    model.wait_staged()
    host = model.device_qp.host
    out["c3_host_csc"] = {"ms_per_solve": dt * 1e3, "solves_per_s": 1.0 / dt, "bytes_to_host": host.bytes_over_pcie(),
                          "bytes_copied_on_host": host.nbytes() - host.bytes_over_pcie(),
                          "what": "config 3 (inequalities + bounds) with host-updated val= Parameters (17 MB staged up) and the host_csc delivery: P, q, l, u "
                                  "(67 MB) down; A's dense block is G itself, which the host wrote: copied on the host (pmt_host_copy_2d), not shipped back"}
    model.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------------


def pack_in_step_stamps(torch, _lib, wl, step, steps):
    """affine_tile_kernel<VAT> INSIDE the step, by the device's own constant-rate clock: every workgroup of the launch reports min(start) /
    max(end) of wall_clock64 — one slot per workgroup (pmt_profile_kernel_stamps), reduced on the host — the kernel's own duration without the in-stream gap a
    HIP-event pair around an in-step launch includes.  Measured in this run, `steps` steps, one read-back per step (outside any timing)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    khz = C.c_int()
    _lib.call("pmt_device_clock_khz", torch.cuda.current_device(), C.byref(khz))
    rate_khz = khz.value
    cap = 4096
    words = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    durs, wgs = [], 0
    _lib.call("pmt_profile_kernel_stamps", dptr(words), cap)
    try:
        for _ in range(steps):
            words.zero_()
            step()
            torch.cuda.synchronize()
            w = words.cpu().numpy().view("uint64").reshape(cap, 2)
            w = w[w[:, 1] != 0]
            if len(w) == 0:
                continue
            durs.append((int(w[:, 1].max()) - int(w[:, 0].min())) / (rate_khz * 1e3))          # seconds
            wgs = len(w)
    finally:
        _lib.call("pmt_profile_kernel_stamps", None, 0)
    if not durs:
        return {"error": "no workgroup of affine_tile_kernel<VAT> reported"}
    durs.sort()
    avg = sum(durs) / len(durs)
    nbytes = 32.0 * wl.m * wl.n
    return {"avg_ms": avg * 1e3, "median_ms": durs[len(durs) // 2] * 1e3, "min_ms": durs[0] * 1e3, "max_ms": durs[-1] * 1e3, "launches": len(durs),
            "workgroups": wgs, "clock_khz": rate_khz, "achieved": nbytes / avg / 1e9, "unit": "GB/s", "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS,
            "measured_in_this_run": True,
            "source": "device clock (wall_clock64) min(start)/max(end) over the launch's workgroups, first workgroup's start to last workgroup's end"}


def rocprof_child(steps=30, warmup=5, timeout=240):
    """When rocprofv3 is on PATH: a short child run of this script's timed loop under `rocprofv3 --kernel-trace`, its kernel stamps reduced to
    the per-kernel averages of the timed launches.  Refreshes what profiles/rocprof_in_step.json holds from THIS box; None when unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="pmt_rocprof_")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps),
               "--warmup", str(warmup), "--timed-loop-only"]
        r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": "rocprofv3 child rc %d, %d trace file(s)" % (r.returncode, len(files))}
        rows = list(csv.DictReader(open(files[0])))
        first = C2Workload.SPINUP_STEPS + warmup
        out = {}
        for name, key in (("gram_sk_kernel", "gram_sk_kernel<"), ("gram_sk_fixup_kernel", "gram_sk_fixup_kernel"), ("affine_tile_kernel<VAT>", "affine_tile_kernel<1")):
            d = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in rows if key in x["Kernel_Name"]]
            if len(d) >= first + steps:
                reg = d[first:first + steps]
                out[name] = {"avg_us": sum(reg) / len(reg), "min_us": min(reg), "max_us": max(reg), "launches": len(reg)}
        out["source"] = "rocprofv3 --kernel-trace of a %d-step child of this command, run by this invocation on this box" % steps
        return out
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pack_three_clocks(torch, _lib, steps=30):
    """affine_tile_kernel<VAT> inside config 2's step by three clocks (the kernel's own device-clock stamps, HIP events around the in-step
    launch, rocprofv3 kernel-trace stamps of a child) beside its stand-alone cold / warm figures"""
    wl = C2Workload(torch, _lib, 0)
    step = wl.step_with_refresh
    for _ in range(10):
        step()
    _lib.call("pmt_profile_enable", 1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    v = profile_report(_lib).get("affine_tile_kernel<VAT>")
    _lib.call("pmt_profile_enable", 0)
    nb = 32.0 * wl.m * wl.n
    out = {"standalone": guarded(constraint_pack_microbench, torch, _lib, wl),
           "in_step": {"device_clock": guarded(pack_in_step_stamps, torch, _lib, wl, step, steps)}}
    if v:
        out["in_step"]["hip_events"] = {"avg_ms": v["avg_ms"], "frac": nb / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "includes the in-stream gap behind the fix-up pass"}
    wl.close()
    child = guarded(rocprof_child)
    ck = (child or {}).get("affine_tile_kernel<VAT>") if isinstance(child, dict) else None
    if ck:
        out["in_step"]["rocprofv3"] = {"avg_ms": ck["avg_us"] * 1e-3, "frac": nb / (ck["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "source": child.get("source")}
    out["rocprofv3_child"] = child
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sections", nargs="*", default=[], help="tall mid host_api pack (default: all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_study.json"))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--shapes", default="", help="tall: rows x columns list instead of the table, e.g. 1048576x16,4096x512")
    a = ap.parse_args()
    import torch
    import parametron_jl_amd as P
    from parametron_jl_amd import _lib
    _lib.require_gpu()
    want = a.sections or ["tall", "mid", "host_api", "pack"]
    out = {}
    if "tall" in want:
        shapes = [tuple(int(v) for v in x.split("x")) for x in a.shapes.split(",") if x] or None
        out["tall"] = guarded(config_tall, torch, _lib, a.steps, shapes)
    if "mid" in want:
        out["mid"] = guarded(config_mid, torch, P)
    if "host_api" in want:
        out["host_api"] = guarded(host_api_c2, torch, P, 10)
    if "pack" in want:
        out["pack"] = guarded(pack_three_clocks, torch, _lib)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    for k, v in (out.get("tall") or {}).items():
        if isinstance(v, dict) and "node_ms" in v:
            print("%-14s %9.1f us  frac %.3f (%s)  mfma %.3f hbm %.3f  %s" % (k, v["node_ms"] * 1e3, v["frac"], v["binding"], v["mfma_frac"], v["hbm_frac"],
                                                                              {kk.replace("_kernel", ""): round(t * 1e3, 1) for kk, t in v["kernels_ms"].items()}))
    for k, v in (out.get("mid") or {}).items():
        print(k, v)
    for k, v in (out.get("host_api") or {}).items():
        if isinstance(v, dict):
            print(k, {x: y for x, y in v.items() if x != "what"})
    print(json.dumps(out.get("pack"), indent=1)[:3000] if "pack" in out else "")


if __name__ == "__main__":
    main()
