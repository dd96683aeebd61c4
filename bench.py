#!/usr/bin/env python
"""bench.py — QP re-evaluations/s of the Parametron hot path on MI355X.

A "step" is one update!(model) (src/model.jl:132-143) of BASELINE config 2 — dense least-squares QP, n = 4096 variables, A 4096x4096,
b 4096, C 512x4096, d 512, fp64 — i.e. setdirty!, the four Parameter callbacks (device-side rand! of A, b, C, d) and one rebuild of every
MOI coefficient buffer from the Parameter values resident in HBM:
    objective   residual . residual, residual = A*x - b   -> canonical MOI.ScalarQuadraticFunction (pmt_quad_gram_f64)
    constraint  C*x - d in Zeros(m)                       -> MOI.VectorAffineFunction       (pmt_affine_pack_vector_f64)
(the literal objective the reference would emit is 1.65 TB at this size; the canonical form = canonicalize!(literal)).
`value` is the DEVICE-RESIDENT rate: inputs in HBM when the timed region starts, outputs left in HBM, nothing crosses PCIe.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU under torch.distributed.run; started here if RANK is unset)

Prints ONE JSON line of at most 4 KB on stdout (rank 0): the contract's fields, `roofline` of the dominant kernel (HIP events in this run;
`traffic` from rocprofv3 --pmc children of this run, tools/bench_pmc.py), `cpu_baseline` (oracle/ on one host core, bounded sample) and
`summary`: one number per other BASELINE config (C1, C3, C4, C5: tools/bench_configs.py) and `C4_sharded` — config 4 sharded by instance
over the N ranks with the slab exchange behind the C ABI, with `ranks_seen`.  Full objects: bench_detail.json (--detail PATH); the longer
studies: tools/bench_study.py.  N > 1 runs N independent config-2 instances (replicas, weak scaling); --workload batch runs config 4 alone."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.modules.setdefault("bench", sys.modules[__name__])          # tools/bench_configs.py imports this module by name, also under __main__

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# f64 MFMA: the local guides list no peak.  AMD's datasheet figure for MI355X FP64 matrix is 78.6 TFLOP/s; tools/f64_coissue.hip
# measures 75.6 TFLOP/s with two MFMA waves per SIMD on the box (profiles/r02_fp64_coissue.txt).
F64_MFMA_PEAK_TFLOPS = 78.6
LINE_LIMIT = 4096              # bytes of the stdout line (the driver stopped parsing a 20.8 KB line in round 5)
DOMINANT = "gram_mid_kernel"


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="c2", choices=["c2", "batch", "launch-check"],
                   help="launch-check: no GPU work at all — the launcher / rank accounting alone over gloo (the CPU test of --gpus N)")
    p.add_argument("--dry-launch", action="store_true", help="print the torch.distributed.run command --gpus N would start (JSON) and exit")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C4 / C5 sections and the sharded config 4")
    p.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc children (roofline.traffic is then the committed replay)")
    p.add_argument("--timed-loop-only", action="store_true", help="spin-up, W warm-up and K timed steps, then exit without a line (the rocprofv3 children)")
    p.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (no per-kernel events)")
    p.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full objects go (the line carries one number each)")
    return p.parse_args()


def dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def profile_report(_lib):
    L = _lib.load()
    n = L.pmt_profile_report(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    L.pmt_profile_report(buf, int(n) + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot, mn, mx = line.split("\t")
        out[name] = {"launches": int(cnt), "avg_ms": float(tot) / max(1, int(cnt)), "min_ms": float(mn), "max_ms": float(mx)}
    return out


def hbm_roofline(kernel, avg_ms, nbytes, **extra):
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": None, "avg_ms": avg_ms, "algorithmic_bytes": nbytes}
    out.update(extra)
    return out


def mfma_roofline(kernel, avg_ms, flops, **extra):
    tf = flops / (avg_ms * 1e-3) / 1e12
    out = {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TFLOPS,
           "traffic": None, "avg_ms": avg_ms, "algorithmic_flops": flops, "peak_source": "datasheet FP64 matrix (75.6 measured with bare MFMAs)"}
    out.update(extra)
    return out


class C2Workload:
    """BASELINE config 2: n = r = 4096, m = 512, through the C ABI (a recorded plan)."""

    n, r, m = 4096, 4096, 512
    SPINUP_STEPS = 15
    name = "C2 dense least-squares QP: n=4096 vars, A 4096x4096, m=512 equality rows, fp64; canonical Q,q,const + C,d MOI triplets"
    units_per_step = 1

    def __init__(self, torch, _lib, rank):
        self.torch, self._lib = torch, _lib
        n, r, m = self.n, self.r, self.m
        dev = torch.device("cuda", torch.cuda.current_device())
        f64, i64 = torch.float64, torch.int64
        # device copies of the Parameter matrices use the package's padded leading dimension
        from parametron_jl_amd.device import padded_lda
        self.lda, self.ldc = padded_lda(r), padded_lda(m)
        self.A = torch.empty(self.lda * n, dtype=f64, device=dev)
        self.b = torch.empty(r, dtype=f64, device=dev)
        self.Cm = torch.empty(self.ldc * n, dtype=f64, device=dev)
        self.d = torch.empty(m, dtype=f64, device=dev)
        self.xvar = torch.arange(1, n + 1, dtype=i64, device=dev)
        self.varmap = torch.arange(1, n + 1, dtype=i64, device=dev)            # model_var_to_optimizer (src/model.jl:100-107)
        self.nq = n * (n + 1) // 2
        self.Q = torch.empty(self.nq * 3, dtype=i64, device=dev)               # MOI.ScalarQuadraticTerm[]
        self.q = torch.empty(n * 2, dtype=i64, device=dev)                     # MOI.ScalarAffineTerm[]
        self.const = torch.empty(1, dtype=f64, device=dev)
        self.Ct = torch.empty(m * n * 3, dtype=i64, device=dev)                # MOI.VectorAffineTerm[]
        self.Cc = torch.empty(m, dtype=f64, device=dev)
        ws_bytes = _lib.load().pmt_quad_gram_workspace_bytes(r, n)
        self.ws = torch.empty(max(1, ws_bytes // 8), dtype=f64, device=dev)
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.seed_offset = 1000 * rank
        self.epoch = 0
        self.refresh()                                                         # device-side Parameter callbacks (README.md:36-43 rand!)
        self.plan = C.c_void_p()
        _lib.call("pmt_plan_create", torch.cuda.current_device(), self.stream, C.byref(self.plan))
        rec = C.c_void_p(_lib.load().pmt_plan_recording_stream(self.plan))
        _lib.call("pmt_plan_begin_record", self.plan)
        _lib.call("pmt_quad_gram_f64", dptr(self.A), self.lda, r, n, dptr(self.xvar), dptr(self.b), -1, 1, dptr(self.varmap),
                  dptr(self.Q), dptr(self.q), dptr(self.const), dptr(self.ws), rec)
        _lib.call("pmt_affine_pack_vector_f64", dptr(self.Cm), self.ldc, m, n, dptr(self.xvar), dptr(self.d), -1, dptr(self.varmap), 0,
                  dptr(self.Ct), dptr(self.Cc), rec)
        _lib.call("pmt_plan_end_record", self.plan)
        # setup, not measurement: first touch of every output buffer and code object, and the power-state ramp of the GPU (the first
        # ~20 ms of fp64 matrix work after an idle period run ~10 % slower).  The W warm-up steps and the K timed steps follow unchanged.
        for _ in range(self.SPINUP_STEPS):
            _lib.call("pmt_plan_update", self.plan)
        torch.cuda.synchronize()

    def refresh(self):
        """the four Parameter callbacks on the device: A, b, C, d ~ U[0,1) (d scaled by 2), seeds A:1 b:2 C:3 d:4 + 1000 per epoch
        (SURVEY.md §8d), distinct per instance"""
        _lib, s, e = self._lib, self.seed_offset, 1000 * self.epoch
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(self.A), self.r, self.n, self.lda, 1 + s + e, 1.0, self.stream)
        _lib.call("pmt_fill_uniform_f64", dptr(self.b), self.r, 2 + s + e, 1.0, self.stream)
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(self.Cm), self.m, self.n, self.ldc, 3 + s + e, 1.0, self.stream)
        _lib.call("pmt_fill_uniform_f64", dptr(self.d), self.m, 4 + s + e, 2.0, self.stream)
        self.epoch += 1

    def step(self):
        self._lib.call("pmt_plan_update", self.plan)

    def step_with_refresh(self):
        """setdirty! + the Parameter callbacks (device-side rand!) + the re-evaluation: update!(model) end to end when the callbacks
        live on the device (src/model.jl:132-133, src/parameter.jl:93-102)"""
        self.refresh()
        self._lib.call("pmt_plan_update", self.plan)

    def close(self):
        if self.plan:
            self.torch.cuda.synchronize()
            self._lib.call("pmt_plan_destroy", self.plan)
            self.plan = None

    # algorithmic work per step
    def gram_flops(self):
        return float(self.r) * self.n * (self.n + 1)

    def gram_bytes(self):
        return 8.0 * self.r * self.n + 24.0 * self.nq

    def step_bytes(self):
        n, r, m = self.n, self.r, self.m
        return 8.0 * (r * n + r + m * n + m) + 24.0 * self.nq + 16.0 * n + 8 + 24.0 * m * n + 8.0 * m


def timed_loop(torch, fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------------------------------
# HBM-side traffic of the step's kernels: rocprofv3 --pmc children of THIS run (tools/bench_pmc.py)

def _pmc():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_pmc
    return bench_pmc


pmc_children = lambda steps=5, timeout=180: _pmc().pmc_children(steps, timeout)
reduce_counter_csv = lambda path, steps: _pmc().reduce_counter_csv(path, steps)
attach_traffic = lambda roof, measured, prefix: _pmc().attach_traffic(roof, measured, prefix)

# ---------------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1)

def cpu_baseline(wl):
    """`cpu_baseline` of the line: the reference's literal CPU path restated in C (oracle/), one thread like the reference, a bounded sample
    (~5 s) timed on this box's host cores — tools/bench_configs.py:cpu_baseline"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    return bench_configs.cpu_baseline(wl)

# ---------------------------------------------------------------------------------------------------------------------------
# stdout, launcher

def guarded(fn, *a):
    try:
        return fn(*a)
    except Exception as e:                       # a failing side section must not take the headline line with it
        return {"error": "%s: %s" % (type(e).__name__, e)}


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too (RCCL prints a version banner through C stdio when
    its first communicator is built — flushed at exit, i.e. BEHIND the JSON line): file descriptor 1 is pointed at stderr for the life of
    the process and the line goes to a duplicate of the original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(args, argv, port=None):
    """the command `--gpus N` (N > 1) re-executes itself as when no launcher has set RANK: the driver's own shape, one rank per GPU"""
    own = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + own


def visible_gpus():
    """GPUs this process could give a rank each (the library's own count: hipGetDeviceCount; 0 without a driver)"""
    from parametron_jl_amd import _lib
    try:
        return int(_lib.load().pmt_device_count())
    except Exception:
        return 0


def fail(message, code=2, **extra):
    """a command that cannot measure what it was asked to says so — one JSON object with "error", exit code != 0 — and never prints a line
    that looks like a result for fewer GPUs"""
    emit_line(dict({"error": message}, **extra))
    sys.stdout.flush()
    os._exit(code)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks here (torch.distributed.run, one process per
    GPU over RCCL) and relay rank 0's line.  Never falls back to fewer ranks."""
    cmd = launch_command(args, argv)
    if args.dry_launch:
        emit_line({"launch": cmd, "n_gpus": args.gpus})
        return 0
    if args.workload != "launch-check":
        have = visible_gpus()
        if have < args.gpus:
            fail("--gpus %d asked for, %d GPU(s) visible on this node: not launched (no line for fewer GPUs is printed)" % (args.gpus, have),
                 gpus_requested=args.gpus, gpus_visible=have)
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env, stdout=_REAL_STDOUT if _REAL_STDOUT is not None else None)
    if rc != 0:
        fail("the %d-rank launch exited with code %d (see stderr)" % (args.gpus, rc), code=rc if 0 < rc < 256 else 1, launch=cmd)
    return 0


def count_ranks(torch, dist, device):
    """ranks that take part in THIS job: one all-reduce of a 1 from every rank"""
    if dist is None:
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one)
    return int(one.item())


def launch_check(args, rank, world):
    """--workload launch-check: the launcher path without a GPU (gloo) — every rank joins, the ranks are counted, rank 0 prints the count.
    Not a measurement: no metric/value."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = count_ranks(torch, dist, "cpu")
    dist.barrier()
    if rank == 0:
        if seen != args.gpus:
            fail("ranks_seen %d != --gpus %d" % (seen, args.gpus), ranks_seen=seen)
        emit_line({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_requested": args.gpus})
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------
# the line

def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short(x, digits=6):
    """floats to `digits` significant figures, recursively (the line is a summary; bench_detail.json keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def summary_of(out):
    """one number per section beside the headline (full objects: bench_detail.json)"""
    c = out.get("configs") or {}
    s = {
        "value_inputs_resident": _get(out, "config_detail", "value_inputs_resident"), "value_200_steps": _get(out, "value_200_steps", "value"),
        "C1_update_us": _get(c, "C1", "update_us"), "C1_solve_us": _get(c, "C1", "solve_us_python_host_mock_optimizer"),
        "C1_model_update_us": _get(c, "C1", "model_update_us_c_entry"),
        "C3_ms": _get(c, "C3", "ms_per_step"), "C4_ms": _get(c, "C4", "ms_per_step"), "C4_frac": _get(c, "C4", "roofline", "frac"),
        "C5_ms": _get(c, "C5", "ms_per_step"), "C5_frac": _get(c, "C5", "roofline", "frac"),
        "C4_sharded_ms": _get(c, "C4_sharded", "ms_per_step"), "C4_sharded_value": _get(c, "C4_sharded", "value"),
        "rccl_calls_made": _get(c, "C4_sharded", "rccl_calls_made"),
        "affine_warm_frac": _get(out, "roofline_affine", "frac"), "affine_cold_frac": _get(out, "roofline_affine", "cold", "frac"),
        "pack_cold_frac": _get(out, "roofline_constraint_pack", "frac"), "pack_in_step_frac_rocprofv3": _get(out, "pack_in_step_rocprofv3", "frac"),
        "gram_shape_frac": {k: round(v["frac"], 3) for k, v in (c.get("shapes") or {}).items() if isinstance(v, dict) and "frac" in v} or None,
        "gram_shape_us": {k: round(v["node_ms"] * 1e3, 1) for k, v in (c.get("shapes") or {}).items() if isinstance(v, dict) and "node_ms" in v} or None,
        "cpu_blas_all_cores": _get(out, "cpu_canonical_blas", "value"), "ranks_seen": out.get("ranks_seen"),
        "errors": [k for k, v in list(c.items()) + [(k, out.get(k)) for k in ("roofline_affine", "roofline_constraint_pack", "cpu_baseline", "sections", "pmc_traffic")]
                   if isinstance(v, dict) and "error" in v] or None,
    }
    return {k: v for k, v in s.items() if v is not None}


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "ranks_seen", "step_algorithmic_bytes", "step_flops")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "avg_ms_rocprofv3", "launches", "traffic", "traffic_read", "traffic_write",
                 "measured_in_this_run", "traffic_source", "algorithmic_flops", "algorithmic_bytes", "peak_source")
CPU_KEYS = ("value", "unit", "cores", "host_cores", "kind", "sample", "seconds_per_reevaluation_extrapolated")


def compact_line(out, detail_path=None):
    """The stdout line: the contract's fields, `roofline`, `cpu_baseline`, `summary` — at most LINE_LIMIT bytes whatever the sections hold
    (strings are cut, then summary entries dropped, before the contract objects would be touched)."""
    line = {k: out[k] for k in LINE_KEYS if k in out}
    roof, cpu = out.get("roofline"), out.get("cpu_baseline")
    line["roofline"] = {k: roof[k] for k in ROOFLINE_KEYS if k in roof} if isinstance(roof, dict) else None
    line["cpu_baseline"] = ({k: cpu[k] for k in CPU_KEYS if k in cpu} if "error" not in cpu else {"error": str(cpu["error"])[:200]}) if isinstance(cpu, dict) else None
    line["summary"] = summary_of(out)
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    line = _short(line)

    def cut(o, limit):
        if isinstance(o, str):
            return o if len(o) <= limit else o[:limit - 3] + "..."
        if isinstance(o, dict):
            return {k: cut(v, limit) for k, v in o.items()}
        return o
    line = cut(line, 400)
    for limit in (240, 120, 60):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line = cut(line, limit)
    while len(json.dumps(line)) > LINE_LIMIT and line["summary"]:
        line["summary"].pop(next(reversed(line["summary"])))
    return line


def finish(out, args):
    path = args.detail
    try:
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
    except Exception as e:
        sys.stderr.write("bench.py: could not write %s (%s)\n" % (path, e))
        path = None
    emit_line(compact_line(out, path))


def sections(out, args, torch, P, _lib, wl, BC):
    """N = 1: the micro-benchmarks on config 2's buffers, the PMC children, the other configurations, the CPU baselines — every one guarded"""
    out["roofline_affine"] = guarded(BC.affine_microbench, torch, _lib, wl)
    out["roofline_constraint_pack"] = guarded(BC.constraint_pack_microbench, torch, _lib, wl)
    wl.close()          # the plan and its side stream go before the other configurations create theirs (streams share hardware queues)
    if out["roofline"] is not None:
        measured = "not run (--no-pmc)" if args.no_pmc else guarded(pmc_children)
        out["pmc_traffic"] = measured
        attach_traffic(out["roofline"], measured, "pmt::gram_mid_kernel<true>")
        us = _get(measured, "affine_tile_kernel<VAT>", "rocprofv3_us")
        if us:      # the constraint pack INSIDE the step (cold input behind the contraction) by the profiler's own clock, this run
            out["pack_in_step_rocprofv3"] = hbm_roofline("affine_tile_kernel<VAT>", us * 1e-3, 32.0 * wl.m * wl.n, source="rocprofv3 --kernel-trace child of this run (no counters), 20 timed steps")
    if not args.no_configs:
        ksteps = max(50, min(args.steps, 100))      # (at least 50 steps: 20 steps of a 1.3 ms configuration are too short to average out a hiccup)
        out["configs"] = {"C1": guarded(BC.config_c1, torch, P, _lib), "C3": guarded(BC.config_c3, torch, P, _lib, ksteps),
                          "C4": guarded(BC.config_c4, torch, _lib, ksteps), "C5": guarded(BC.config_c5, torch, P, _lib, ksteps),
                          "shapes": guarded(BC.config_tall, torch, _lib, 20, BC.LINE_SHAPES)}
    out["cpu_baseline"] = None if args.no_cpu_baseline else guarded(cpu_baseline, wl)
    out["cpu_canonical_blas"] = None if args.no_cpu_baseline else guarded(BC.cpu_canonical_blas, wl)


def main():
    args = parse_args()
    claim_stdout()
    args.gpus >= 1 or fail("--gpus must be >= 1")
    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_launch):
        return self_launch(args, sys.argv[1:])
    rank, world, local_rank = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if world != args.gpus:
        # a launcher started a different number of ranks than the command line names: the line would carry the wrong n_gpus either way
        if rank == 0:
            fail("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), gpus_requested=args.gpus, world_size=world)
        os._exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.workload == "launch-check":
        return launch_check(args, rank, world)
    import torch
    import parametron_jl_amd as P
    from parametron_jl_amd import _lib
    _lib.require_gpu()
    if torch.cuda.device_count() <= local_rank:
        if rank == 0 or world == 1:
            fail("rank with LOCAL_RANK=%d has no GPU: %d visible, --gpus %d" % (local_rank, torch.cuda.device_count(), args.gpus),
                 gpus_requested=args.gpus, gpus_visible=torch.cuda.device_count())
        os._exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # the FIRST collective builds the RCCL communicator (hundreds of ms with the GPU idle): pay for that here, in front of the workload's
        # spin-up and warm-up steps, not in the barrier that brackets the timed region
        dist.barrier()
        torch.cuda.synchronize()
    if args.workload == "batch":
        from parametron_jl_amd import batch
        return batch.bench(args, torch, dist, _lib, rank, world, emit=emit_line)

    ranks_seen = count_ranks(torch, dist, "cuda")
    if ranks_seen != args.gpus:
        if rank == 0:
            fail("ranks_seen %d != --gpus %d" % (ranks_seen, args.gpus), ranks_seen=ranks_seen)
        os._exit(2)
    wl = C2Workload(torch, _lib, rank)
    if args.graph:
        _lib.call("pmt_plan_instantiate_graph", wl.plan)
    step = wl.step_with_refresh           # update!(model): setdirty! + the Parameter callbacks (151 MB regenerated in HBM) + the rebuild

    def bracketed(fn, steps):
        """K steps between barrier + synchronize on both sides; this rank's time, then the MAX over ranks"""
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0          # this rank's K steps are complete; the MAX over ranks below is the job's time
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step()
    _lib.call("pmt_profile_enable", 0)
    elapsed = bracketed(step, args.steps)          # THE timed region: no HIP events, no profiling hooks inside
    if args.timed_loop_only:
        wl.close()
        if dist:
            dist.destroy_process_group()
        return 0
    ms_per_step = elapsed / args.steps * 1e3
    value = world * wl.units_per_step * args.steps / elapsed
    elapsed_resident = bracketed(wl.step, args.steps)          # the rebuild alone (callbacks outside): the same K steps, bracketed the same way
    kernels = {}
    if not args.graph:
        # the DOMINANT kernel's roofline: a second pass of K identical steps with a HIP-event pair (on the stream the kernel is launched on)
        # around its launches only — every event pair costs queue time, so it stays out of the timed region above
        _lib.call("pmt_profile_filter", DOMINANT.encode())
        _lib.call("pmt_profile_enable", 1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        kernels = profile_report(_lib)
        _lib.call("pmt_profile_enable", 0)
        _lib.call("pmt_profile_filter", None)
        for v in kernels.values():
            v["measured"] = "HIP events around this kernel only, %d steps right behind the timed region" % args.steps
        _lib.call("pmt_profile_enable", 1)                   # all kernels of the step, 20 more steps
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        for k, v in profile_report(_lib).items():
            kernels.setdefault(k, dict(v, measured="HIP events around every kernel, separate 20-step pass"))
        _lib.call("pmt_profile_enable", 0)

    out = None
    if rank == 0:
        out = {
            "metric": "QP re-evaluations/sec (Q,q,C,d rebuild) at n=4096",
            "value": value, "unit": "re-evaluations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name, "n": wl.n, "r": wl.r, "m": wl.m, "objective_mode": "canonical", "instances_per_gpu": 1,
                       "parallelism": "replicas (independent QP instances, no collective)" if world > 1 else "single GPU",
                       "replay": "hipGraph" if args.graph else "tape",
                       "step": "update!(model): setdirty! + 4 device-side Parameter callbacks (rand! of A,b,C,d) + rebuild of Q,q,const,C,d; all in HBM"},
            "config_detail": {"setup_spinup_steps": wl.SPINUP_STEPS, "device_lda": [wl.lda, wl.ldc],
                              "value_inputs_resident": world * wl.units_per_step * args.steps / elapsed_resident,
                              "ms_per_step_inputs_resident": elapsed_resident / args.steps * 1e3,
                              "boundary": "device-resident hand-off: the Parameter callbacks write HBM, MOI buffers left in HBM; nothing crosses "
                                          "PCIe (PCIe-inclusive rates: tools/bench_study.py host_api).  value_inputs_resident = the rebuild alone"},
            "step_algorithmic_bytes": wl.step_bytes(), "step_flops": wl.gram_flops(), "ranks_seen": ranks_seen, "kernels": kernels,
        }
        g = kernels.get(DOMINANT)
        out["roofline"] = mfma_roofline(DOMINANT, g["avg_ms"], wl.gram_flops(), launches=g["launches"], algorithmic_bytes=wl.gram_bytes()) if g else None
    if world == 1 and not args.graph:
        if args.steps < 200:
            t = timed_loop(torch, step, 200)
            out["value_200_steps"] = {"value": 200 / t, "ms_per_step": t / 200 * 1e3, "steps": 200,
                                      "what": "the same step timed over 200 steps (> 0.2 s of device time) right after the K-step measurement"}
        try:                                    # nothing behind the headline may take the line with it
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs as BC
            sections(out, args, torch, P, _lib, wl, BC)
        except Exception as e:
            out["sections"] = {"error": "%s: %s" % (type(e).__name__, e)}
            wl.close()
    else:
        wl.close()
        if rank == 0 and out["roofline"] is not None:
            attach_traffic(out["roofline"], "not run (N > 1 or --graph)", "pmt::gram_mid_kernel<true>")
    if not args.no_configs:
        # every rank: BASELINE config 4 sharded by instance over the N ranks (N = 1: the same code path, single-rank communicator)
        from parametron_jl_amd import batch
        ksteps = max(50, min(args.steps, 100))
        # This section has only ever run with one rank on hardware.  Should a collective of it hang with N > 1 (one rank failing alone), the
        # headline — complete at this point — must still come out: after 150 s rank 0 prints the line without the section and every rank exits.
        import threading

        def give_up():
            if rank == 0:
                out.setdefault("configs", {})["C4_sharded"] = {"error": "timeout: the sharded section did not finish within 150 s"}
                finish(out, args)
            os._exit(0)
        watchdog = threading.Timer(150.0, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            sharded = batch.measure(torch, dist, rank, world, ksteps, args.warmup)
        except Exception as e:                  # (a failing side section must not take the headline with it; every rank fails alike)
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}
        watchdog.cancel()
        if rank == 0:
            out.setdefault("configs", {})["C4_sharded"] = sharded
            if isinstance(sharded, dict) and sharded.get("ranks_seen") not in (None, ranks_seen):
                out["ranks_seen"] = min(ranks_seen, sharded["ranks_seen"])
    if rank == 0:
        finish(out, args)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
