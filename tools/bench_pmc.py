"""bench.py's counter side: HBM-side traffic of the step's kernels from two short `rocprofv3 --kernel-trace --pmc` children of the bench
run itself (FETCH_SIZE and WRITE_SIZE in separate passes), and the committed replay bench.py falls back to when rocprofv3 is absent."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
DOMINANT = "gram_mid_kernel"


def pmc_replay(prefix):
    """HBM-side bytes per launch of the kernel whose name starts with `prefix` from the COMMITTED rocprofv3 PMC passes
    (profiles/pmc_traffic.json) — the fallback when this run cannot measure them (no rocprofv3 on PATH, N > 1)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            data = json.load(fh)
    except Exception:
        return None
    for name, v in data.items():
        if name.startswith(prefix) and isinstance(v, dict):
            return {"read": v["read_bytes"], "write": v["write_bytes"]}
    return None




PMC_KERNELS = {"gram_mid_kernel": "gram_mid_kernel<", "gram_sk_kernel": "gram_sk_kernel<", "gram_sk_fixup_kernel": "gram_sk_fixup_kernel", "gram_linear_kernel": "gram_linear_kernel",
               "affine_tile_kernel<VAT>": "affine_tile_kernel<1"}


def reduce_counter_csv(path, steps):
    """rocprofv3's counter_collection.csv -> {kernel: mean Counter_Value over its last `steps` launches} for the step's kernels"""
    import csv
    rows = list(csv.DictReader(open(path)))
    out = {}
    for name, key in PMC_KERNELS.items():
        v = [float(r["Counter_Value"]) for r in rows if key in r["Kernel_Name"]]
        if v:
            out[name] = sum(v[-steps:]) / len(v[-steps:])
    return out


def reduce_trace_csv(path, steps):
    """rocprofv3's kernel_trace.csv -> {kernel: mean duration in us over its last `steps` launches} for the step's kernels (the profiler's
    own clock; under --pmc the dispatches are serialised, a kernel's own begin-to-end time is what it is in the step)"""
    import csv
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    for name, key in PMC_KERNELS.items():
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
        if d:
            out[name] = sum(d[-steps:]) / len(d[-steps:])
    return out


def pmc_children(steps=5, timeout=180):
    """Two short children of this command under `rocprofv3 --kernel-trace --pmc <counter>` — FETCH_SIZE and WRITE_SIZE in SEPARATE passes
    (they do not fit the TCC's counter slots together), the timed loop only — reduced to bytes per launch as MI355X_MICROARCH.md's HBM
    section prescribes: both counters in KiB, FETCH_SIZE doubled on gfx950.  None when rocprofv3 is not on PATH; {"error": ...} on failure."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = {}
    # a third child WITHOUT counters (a --pmc pass serialises and stretches the dispatches: the pack reads 14.3 us there, 12.7 plain): the
    # profiler's own begin / end stamps of the step's kernels over 20 timed steps
    for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0), (None, 0.0)):
        tmp = tempfile.mkdtemp(prefix="pmt_pmc_")
        try:
            nsteps = steps if counter else 20
            cmd = [exe, "--kernel-trace"] + (["--pmc", counter] if counter else []) + ["--output-format", "csv", "-d", tmp, "--", sys.executable, BENCH,
                   "--steps", str(nsteps), "--warmup", "5" if not counter else "1", "--timed-loop-only"]
            r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if not counter:
                traces = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
                if r.returncode == 0 and traces:
                    res["durations"] = reduce_trace_csv(traces[0], nsteps)
                continue
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": "rocprofv3 --pmc %s child: rc %d, %d counter file(s): %s" % (counter, r.returncode, len(files), r.stderr.decode(errors="replace")[-300:])}
            res[counter] = {k: v * scale for k, v in reduce_counter_csv(files[0], steps).items()}
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out = {k: {"read": res["FETCH_SIZE"][k], "write": res["WRITE_SIZE"].get(k, 0.0)} for k in res["FETCH_SIZE"]}
    for k, us in res.get("durations", {}).items():
        out.setdefault(k, {})["rocprofv3_us"] = us
    out["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate %d-step children of this invocation on this box; "
                     "KiB -> bytes, FETCH_SIZE x2 (gfx950); rocprofv3_us: a third child with --kernel-trace only, 20 timed steps" % steps)
    return out


def attach_traffic(roof, measured, prefix):
    """roofline.traffic = HBM-side bytes per launch of the dominant kernel: measured by this run's children, else the committed replay"""
    m = measured.get(DOMINANT) if isinstance(measured, dict) else None
    if m:
        roof.update(traffic=m["read"] + m["write"], traffic_read=m["read"], traffic_write=m["write"], measured_in_this_run=True,
                    traffic_source="rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE children of this run (separate passes, x1024, FETCH x2)")
        if m.get("rocprofv3_us"):           # the same kernel by the profiler's own clock (a third child, --kernel-trace only)
            roof["avg_ms_rocprofv3"] = m["rocprofv3_us"] * 1e-3
        return roof
    rp = pmc_replay(prefix)
    why = "rocprofv3 not on PATH" if measured is None else (measured.get("error", "no launches seen") if isinstance(measured, dict) else "not run")
    roof.update(traffic=(rp["read"] + rp["write"]) if rp else None, traffic_read=rp and rp["read"], traffic_write=rp and rp["write"],
                measured_in_this_run=False, traffic_source="replay of profiles/pmc_traffic.json (%s)" % why[:160])
    return roof


