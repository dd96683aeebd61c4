#!/usr/bin/env python
"""bench.py — QP re-evaluations/s of the Parametron hot path on MI355X.

A "step" is one update!(model) (src/model.jl:132-143) of BASELINE config 2 — dense least-squares QP, n = 4096
variables, A 4096x4096, b 4096, C 512x4096, d 512, fp64 — i.e. one rebuild of every MOI coefficient buffer
(Q, q, constant, constraint triplets, constraint constants) from the Parameter values resident in HBM:
    objective   residual . residual, residual = A*x - b   -> canonical MOI.ScalarQuadraticFunction (pmt_quad_gram_f64)
    constraint  C*x - d in Zeros(m)                       -> MOI.VectorAffineFunction       (pmt_affine_pack_vector_f64)
The literal (uncombined) objective the reference would emit is 1.65 TB at this size; the canonical
form = canonicalize!(literal) is what is rebuilt here, see DESIGN.md.  `value` is the DEVICE-RESIDENT hand-off rate: inputs
in HBM when the timed region starts, outputs left in HBM (the PCIe-inclusive rates of the host API are reported next to
it under "host_api", never as `value`).

python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)
The headline (`value`) is config 2 on every GPU: N > 1 runs N independent QP instances (one per GPU, no data-path collective —
a single QP's rebuild does not shard): weak scaling.  The SAME line carries `configs.C4_sharded`: BASELINE config 4 (8192 independent
n = 128 QPs) sharded by instance over the N ranks with the exchange of the coefficient slabs behind the C ABI (RCCL send/recv per
peer) — compute only, computation + one monolithic all-gather, and the overlapped step, with `ranks_seen`, the bytes every rank puts
on the wire and the per-link-bound cost of them: north_star's multi-GPU configuration under the driver's own command.  At N = 1 it
runs the same code path with the single-rank communicator.
--workload batch runs config 4 alone.  Prints ONE JSON line on rank 0.  At N = 1 the line also carries, under "configs", the other
BASELINE configurations (C3, C4 on one GPU, C5) measured in the same process with their dominant kernel's roofline.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# f64 MFMA: the local guides list no peak.  AMD's datasheet figure for MI355X FP64 matrix is 78.6 TFLOP/s;
# tools/f64_coissue.hip measures 75.6 TFLOP/s with two MFMA waves per SIMD on the box (profiles/r02_fp64_coissue.txt).
F64_MFMA_PEAK_TFLOPS = 78.6
TRAFFIC_SOURCE = "profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not measured in this run)"


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernel whose name starts with `prefix`, from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE/WRITE_SIZE, separate passes, gfx950 x2 read correction).  PMC counters cannot be
    collected from inside this process; None when no measurement is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            data = json.load(fh)
    except Exception:
        return None
    for name, v in data.items():
        if name.startswith(prefix):
            return v["read_bytes"] + v["write_bytes"]
    return None


def rocprof_in_step(prefix):
    """average duration (us) of the kernel whose name starts with `prefix` over the launches of the timed region, from the committed
    rocprofv3 --kernel-trace run of this command (profiles/rocprof_in_step.json, written by tools/collect_profiles.sh) — the kernel's own
    begin/end stamps, without the in-stream gap the HIP events of an in-step launch include.  None when no record is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", "rocprof_in_step.json")) as fh:
            data = json.load(fh)
    except Exception:
        return None
    for name, v in data.items():
        if name.startswith(prefix):
            return v
    return None


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="c2", choices=["c2", "batch", "launch-check"],
                   help="launch-check: no GPU work at all — the launcher / rank accounting alone over gloo (the CPU test of --gpus N)")
    p.add_argument("--dry-launch", action="store_true", help="print the torch.distributed.run command --gpus N would start (JSON) and exit")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 / host-API sections")
    p.add_argument("--timed-loop-only", action="store_true", help="spin-up, W warm-up and K timed steps, then exit without a line (the rocprofv3 child)")
    p.add_argument("--no-rocprof-child", action="store_true", help="do not run the short rocprofv3 --kernel-trace child even when rocprofv3 is on PATH")
    p.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (no per-kernel events)")
    p.add_argument("--pack-first", action="store_true", help="A/B: record the constraint pack in FRONT of the contraction (same stream)")
    p.add_argument("--side-lane", default="off", choices=["off", "tile", "background"],
                   help="A/B: record config 2's constraint pack on the plan's side lane (Model.initialize does so only when a model has "
                        "several such entries, e.g. config 3 or the device hand-off; for one kernel it does not pay, DESIGN.md section 4)")
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


def hbm_roofline(kernel, avg_ms, nbytes, traffic_prefix=None, **extra):
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": pmc_traffic(traffic_prefix) if traffic_prefix else None, "traffic_source": TRAFFIC_SOURCE if traffic_prefix else None,
           "avg_ms": avg_ms, "algorithmic_bytes": nbytes}
    out.update(extra)
    return out


def mfma_roofline(kernel, avg_ms, flops, traffic_prefix=None):
    tf = flops / (avg_ms * 1e-3) / 1e12
    return {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TFLOPS,
            "traffic": pmc_traffic(traffic_prefix) if traffic_prefix else None, "traffic_source": TRAFFIC_SOURCE if traffic_prefix else None,
            "avg_ms": avg_ms, "algorithmic_flops": flops,
            "peak_source": "MI355X datasheet FP64 matrix 78.6 TFLOP/s (not in the local guides; 75.6 measured with bare MFMAs); see DESIGN.md",
            # context, not the contract's frac: under this kernel the chip holds 2.19 GHz (GRBM_GUI_ACTIVE / duration, profiles/r01e_gram_sk_pmc_sq.txt),
            # where 1024 SIMDs x 32 FLOP/clk deliver 71.8 TFLOP/s
            "frac_of_peak_at_sustained_clock": tf / 71.8}


class C2Workload:
    """BASELINE config 2: n = r = 4096, m = 512, through the C ABI (a recorded plan)."""

    n, r, m = 4096, 4096, 512

    def __init__(self, torch, _lib, rank, side_lane=False, background=False, pack_first=False):
        self.torch, self._lib = torch, _lib
        n, r, m = self.n, self.r, self.m
        dev = torch.device("cuda", torch.cuda.current_device())
        f64, i64 = torch.float64, torch.int64
        # device copies of the Parameter matrices use the package's padded leading dimension: same values,
        # column stride 4096+64 / 512+64 doubles instead of a multiple of 4 KiB
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
        if pack_first:
            _lib.call("pmt_affine_pack_vector_f64", dptr(self.Cm), self.ldc, m, n, dptr(self.xvar), dptr(self.d), -1, dptr(self.varmap), 0, dptr(self.Ct), dptr(self.Cc), rec)
        _lib.call("pmt_quad_gram_f64", dptr(self.A), self.lda, r, n, dptr(self.xvar), dptr(self.b), -1, 1, dptr(self.varmap),
                  dptr(self.Q), dptr(self.q), dptr(self.const), dptr(self.ws), rec)
        # (A/B only: with ONE constraint copy and no hand-off Model.initialize() keeps it on the plan's stream)
        if side_lane:
            _lib.call("pmt_plan_set_lane", self.plan, 1)
        if not pack_first:
            _lib.call("pmt_affine_pack_vector_background_f64" if (side_lane and background) else "pmt_affine_pack_vector_f64", dptr(self.Cm), self.ldc, m, n, dptr(self.xvar),
                      dptr(self.d), -1, dptr(self.varmap), 0, dptr(self.Ct), dptr(self.Cc), rec)
        _lib.call("pmt_plan_end_record", self.plan)
        # setup, not measurement: first touch of every output buffer and code object, and the power-state ramp of the GPU —
        # the first ~20 ms of fp64 matrix work after an idle period run ~10 % slower (profiles/r01c_lda_padding.txt shows the
        # effect on back-to-back identical launches).  The W warmup steps and the K timed steps follow unchanged.
        for _ in range(self.SPINUP_STEPS):
            _lib.call("pmt_plan_update", self.plan)
        torch.cuda.synchronize()

    SPINUP_STEPS = 15

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

    def close(self):
        if self.plan:
            self.torch.cuda.synchronize()
            self._lib.call("pmt_plan_destroy", self.plan)
            self.plan = None

    def step_with_refresh(self):
        """setdirty! + the Parameter callbacks (device-side rand!) + the re-evaluation: what update!(model) does end to end when the
        callbacks live on the device (src/model.jl:132-133, src/parameter.jl:93-102)"""
        self.refresh()
        self._lib.call("pmt_plan_update", self.plan)

    name = "C2 dense least-squares QP: n=4096 vars, A 4096x4096, m=512 equality rows, fp64; canonical Q,q,const + C,d MOI triplets"
    units_per_step = 1

    # algorithmic work of the dominant kernel per launch
    def gram_flops(self):
        return float(self.r) * self.n * (self.n + 1)

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


def constraint_pack_microbench(torch, _lib, wl, reps=20):
    """affine_tile_kernel<VAT> on config 2's constraint block (C 512 x 4096 -> MOI.VectorAffineTerms), launched alone on the stream.
    In the step this kernel runs behind the contraction, which has streamed ~1 GB through the Infinity Cache since C was last touched: its
    input is always COLD there (and the kernel reads a large block with the nontemporal policy for that reason, affine.hip).  The stand-alone
    figure is therefore taken over SIX (C, output) pairs visited in turn (403 MB > the 256 MiB cache); `warm` — the same pair every launch,
    what rounds 1-3 reported — is kept beside it."""
    m, n = wl.m, wl.n
    Cs = [wl.Cm] + [torch.empty_like(wl.Cm) for _ in range(5)]
    outs = [wl.Ct] + [torch.empty_like(wl.Ct) for _ in range(5)]
    for c in Cs[1:]:
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(c), m, n, wl.ldc, 78, 1.0, wl.stream)

    def run(i):
        _lib.call("pmt_affine_pack_vector_f64", dptr(Cs[i % 6]), wl.ldc, m, n, dptr(wl.xvar), dptr(wl.d), -1, dptr(wl.varmap), 0, dptr(outs[i % 6]), dptr(wl.Cc), wl.stream)

    def timed(which):
        for i in range(6):
            run(which(i))
        torch.cuda.synchronize()
        _lib.call("pmt_profile_enable", 1)
        for i in range(reps + reps // 2):
            run(which(i))
        torch.cuda.synchronize()
        rep = profile_report(_lib)
        _lib.call("pmt_profile_enable", 0)
        return rep.get("affine_tile_kernel<VAT>")
    k = timed(lambda i: i)
    kw = timed(lambda i: 0)
    if not k:
        return None
    out = hbm_roofline("affine_tile_kernel<VAT>", k["avg_ms"], 32.0 * m * n, "pmt::affine_tile_kernel<1",
                       note="stand-alone launches of the tile kernel over six (C, output) pairs visited in turn (cold inputs, as in the step)")
    if kw:
        out["warm"] = {"avg_ms": kw["avg_ms"], "achieved": 32.0 * m * n / (kw["avg_ms"] * 1e-3) / 1e9, "frac": 32.0 * m * n / (kw["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "what": "the same (C, output) pair every launch: C comes out of the Infinity Cache, where the nontemporal read policy of the "
                               "large-block form costs; not how the step runs it"}
    return out


def affine_microbench(torch, _lib, wl, reps=20):
    """The affine-assembly kernel on the 4096x4096 residual block (matvecmul! + vecsubtract!, LinearTerm output):
    24*r*n algorithmic bytes per launch (8 read + 16 written), north_star's >= 60 % HBM target.  A MICROBENCHMARK: this
    kernel is not part of the timed step (the canonical objective reads A directly; the constraint block uses the VAT form
    of the same kernel, reported as roofline_constraint_pack)."""
    n, r = wl.n, wl.r
    out = torch.empty(r * n * 2, dtype=torch.int64, device=wl.A.device)
    consts = torch.empty(r, dtype=torch.float64, device=wl.A.device)
    for _ in range(3):
        _lib.call("pmt_affine_assemble_f64", dptr(wl.A), wl.lda, r, n, dptr(wl.xvar), dptr(wl.b), -1, dptr(out), dptr(consts), wl.stream)
    torch.cuda.synchronize()
    _lib.call("pmt_profile_enable", 1)
    for _ in range(reps):
        _lib.call("pmt_affine_assemble_f64", dptr(wl.A), wl.lda, r, n, dptr(wl.xvar), dptr(wl.b), -1, dptr(out), dptr(consts), wl.stream)
    torch.cuda.synchronize()
    rep = profile_report(_lib)
    _lib.call("pmt_profile_enable", 0)
    k = rep.get("affine_tile_kernel<LT>")
    if not k:
        return None
    warm = hbm_roofline("affine_tile_kernel<LT>", k["avg_ms"], 24.0 * r * n, "pmt::affine_tile_kernel<0", shape="A 4096x4096 -> 16.8M LinearTerms",
                        note="microbenchmark of the affine-assembly kernel; not a kernel of the timed step.  WARM: the same 134 MB A is read "
                             "by every launch and fits the 256 MiB Infinity Cache; `cold` rotates three (A, output) pairs, 1.2 GB in all")
    # cold inputs: three distinct A buffers (403 MB > 256 MiB of Infinity Cache) and three output buffers, visited in turn
    As = [wl.A] + [torch.empty_like(wl.A) for _ in range(2)]
    outs = [out] + [torch.empty_like(out) for _ in range(2)]
    for a in As[1:]:
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(a), r, n, wl.lda, 77, 1.0, wl.stream)
    for i in range(6):
        _lib.call("pmt_affine_assemble_f64", dptr(As[i % 3]), wl.lda, r, n, dptr(wl.xvar), dptr(wl.b), -1, dptr(outs[i % 3]), dptr(consts), wl.stream)
    torch.cuda.synchronize()
    _lib.call("pmt_profile_enable", 1)
    for i in range(reps + reps // 2):
        _lib.call("pmt_affine_assemble_f64", dptr(As[i % 3]), wl.lda, r, n, dptr(wl.xvar), dptr(wl.b), -1, dptr(outs[i % 3]), dptr(consts), wl.stream)
    torch.cuda.synchronize()
    rep = profile_report(_lib)
    _lib.call("pmt_profile_enable", 0)
    kc = rep.get("affine_tile_kernel<LT>")
    if kc:
        warm["cold"] = {"avg_ms": kc["avg_ms"], "achieved": 24.0 * r * n / (kc["avg_ms"] * 1e-3) / 1e9, "frac": 24.0 * r * n / (kc["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "what": "three (A, output) pairs visited in turn: every launch reads an A that left the Infinity Cache two launches ago"}
    return warm


# ---------------------------------------------------------------------------------------------------------------------------
# the other BASELINE configurations, measured in the same process (N = 1)

def config_c3(torch, P, _lib, steps):
    """C3 through the host API: C2's objective + 512 inequality rows + bounds, the constraint Parameters in the reference's `val=`
    form, rewritten by the host before every update.  Serial: update!() uploads them on the plan's stream.  Staged: the values of
    update k+1 travel on the copy stream while update k runs (Model.stage_parameters)."""
    from parametron_jl_amd import workloads
    model, bufs = workloads.config3(pinned=True, handoff="device")
    P.solve(model)
    ctx = model.device()

    def serial():
        model.update(synchronize=False)

    def staged():
        model.stage_parameters()
        model.update(synchronize=False)
    out = {"workload": "C3: C2 objective + G*x <= h (512 rows) + x >= l, x <= u; G,h,l,u host-updated val= Parameters (17 MB per update)"}
    for name, fn in (("serial_upload", serial), ("staged_upload", staged)):
        for _ in range(10):
            fn()
        ctx.synchronize()
        t = timed_loop(torch, fn, steps)
        model.wait_staged()
        out[name] = {"ms_per_step": t / steps * 1e3, "re_evaluations_per_s": steps / t}
    _lib.call("pmt_profile_enable", 1)
    for _ in range(10):
        staged()
    ctx.synchronize()
    kern = profile_report(_lib)
    _lib.call("pmt_profile_enable", 0)
    out["ms_per_step"] = out["staged_upload"]["ms_per_step"]
    out["kernels"] = kern
    g = kern.get("gram_sk_kernel")
    if g:
        out["roofline"] = mfma_roofline("gram_sk_kernel", g["avg_ms"], 4096.0 * 4096 * 4097)
    v = kern.get("affine_tile_kernel<VAT>")
    if v:
        out["roofline_inequality_pack"] = hbm_roofline("affine_tile_kernel<VAT>", v["avg_ms"], 32.0 * 512 * 4096)
    model.close()
    return out


def config_c1(torch, P, _lib, steps=400):
    """BASELINE config 1 — README Example 1 (n = 8, m = 2, literal objective), the one configuration the reference publishes a number for
    (README.md:132-136: solve! 51.863 us including OSQP; the update! share is ~15 us on one CPU core).  Through the host API
    with device-side Parameter callbacks; Model.initialize records callbacks + tape and the library replays them as ONE launch (small plan,
    csrc/small.hip).  update_us = host wall time per update!(model) in a pipelined loop (setdirty! + seeds + one launch, no MOI fetch);
    kernel_us = HIP-event time of that launch; the same plan replayed as recorded (9 launches) beside it."""
    model = P.Model(P.MockOptimizer(), quadratic_mode="literal")
    n, m = 8, 2
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((n, n), 1, model); b = P.DeviceUniformParameter((n,), 2, model)
    Cm = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    r = A * x - b
    P.objective(model, P.Minimize, P.dot(r, r)); P.constraint(model, Cm * x == d)
    P.solve(model)
    ctx = model.device()

    def upd():
        model.setdirty(); model._run_tape(fetch=False)

    def wall(fn, k):
        for _ in range(50):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / k * 1e6

    def kernel_us():
        _lib.call("pmt_profile_enable", 1)
        for _ in range(50):
            upd()
        ctx.synchronize()
        rep = profile_report(_lib)
        _lib.call("pmt_profile_enable", 0)
        return {k: v["avg_ms"] * 1e3 for k, v in rep.items()}
    fz = ctx.fused()
    out = {"workload": "C1 README Example 1: n=8 variables, A 8x8, m=2 equality rows, literal objective, device-side rand! callbacks",
           "fused": fz, "tape_entries": ctx.tape_length()}
    out["update_us"] = wall(upd, steps)
    out["plan_update_us"] = wall(ctx.replay, steps)            # the C-ABI call alone (pmt_plan_update), seeds unchanged
    out["kernel_us"] = kernel_us()
    ctx.synchronize()
    ctx.set_fusion(False)
    out["unfused"] = {"update_us": wall(upd, steps), "plan_update_us": wall(ctx.replay, steps), "kernel_us": kernel_us(), "launches": ctx.fused()["exec_length"]}
    ctx.set_fusion(True)
    # the whole solve!(model) through the Python host with a do-nothing optimizer and the results ON THE HOST at the end (callbacks, one
    # launch whose kernels store the MOI buffers into the function objects' page-locked arrays, one synchronisation, MOI.set calls)
    def solve_wall(k):
        for _ in range(30):
            P.solve(model)
        t0 = time.perf_counter()
        for _ in range(k):
            P.solve(model)
        return (time.perf_counter() - t0) / k * 1e6
    out["solve_us_python_host_mock_optimizer"] = solve_wall(steps)
    out["reference"] = {"solve_us_incl_osqp": 51.863, "update_us_estimate": 15.0,
                        "source": "README.md:132-136 (BenchmarkTools median of solve!, other hardware); the update! share per BASELINE.md section 1"}
    model.close()
    return out


TALL_SHAPES = [(1 << 20, 128), (262144, 512), (65536, 1024), (8192, 128), (1 << 20, 16), (1 << 20, 64), (4096, 512), (100, 100)]


def config_tall(torch, _lib, steps=20):
    """The Gram node (pmt_quad_gram_f64: Q, q, constant) on the shapes beside config 2 — tall (rows >> columns, the usual shape of
    README.md:34-38 with real data), narrow, mid-size and the reference's own sizes.  Per shape: node time, the algorithmic flops r n (n + 1) against the f64 MFMA peak and the algorithmic bytes
    (8 r n read + 24 n (n + 1) / 2 written) against HBM; `binding` names the roofline whose algorithmic time is longer."""
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    from parametron_jl_amd.device import padded_lda
    out = {}
    for r, n in TALL_SHAPES:
        lda = padded_lda(r)
        A = torch.empty(lda * n, dtype=torch.float64, device=dev); b = torch.empty(r, dtype=torch.float64, device=dev)
        _lib.call("pmt_fill_uniform_matrix_f64", dptr(A), r, n, lda, 1, 1.0, stream); _lib.call("pmt_fill_uniform_f64", dptr(b), r, 2, 1.0, stream)
        xvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
        nq = n * (n + 1) // 2
        Q = torch.empty(nq * 3, dtype=torch.int64, device=dev); q = torch.empty(n * 2, dtype=torch.int64, device=dev)
        c = torch.empty(1, dtype=torch.float64, device=dev)
        ws = torch.empty(max(1, _lib.load().pmt_quad_gram_workspace_bytes(r, n) // 8), dtype=torch.float64, device=dev)

        def run():
            _lib.call("pmt_quad_gram_f64", dptr(A), lda, r, n, dptr(xvar), dptr(b), -1, 1, dptr(xvar), dptr(Q), dptr(q), dptr(c), dptr(ws), stream)
        # (at least 10 calls and 40 ms: the first shape of a process measured after 4 ms of warm-up read 10 % slow — the clock settles in ~30 ms)
        t0 = time.perf_counter()
        k = 0
        while k < 10 or time.perf_counter() - t0 < 0.04:
            run(); k += 1
            if k % 10 == 0:
                torch.cuda.synchronize()
        t = timed_loop(torch, run, steps) / steps
        _lib.call("pmt_profile_enable", 1)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        kern = {k: v["avg_ms"] for k, v in profile_report(_lib).items()}
        _lib.call("pmt_profile_enable", 0)
        flops, nbytes = float(r) * n * (n + 1), 8.0 * r * n + 24.0 * nq
        t_mfma, t_hbm = flops / (F64_MFMA_PEAK_TFLOPS * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
        # (a node whose algorithmic time on BOTH rooflines is under 5 us is bound by its launches: two or four kernels, ~6 us each in-stream)
        binding = "launch latency" if max(t_mfma, t_hbm) < 5e-6 else ("mfma" if t_mfma >= t_hbm else "hbm")
        out["%dx%d" % (r, n)] = {"node_ms": t * 1e3, "mfma_frac": t_mfma / t, "hbm_frac": t_hbm / t, "binding": binding, "launches": len(kern),
                                 "frac": max(t_mfma, t_hbm) / t, "tflops": flops / t / 1e12, "A_TBps": 8.0 * r * n / t / 1e12, "kernels_ms": kern}
        del A, b, Q, q, ws
    return out


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


def config_c4(torch, _lib, steps):
    from parametron_jl_amd import batch
    total, n, r, m = 8192, 128, 128, 16
    wl = batch.BatchLSQ(torch, total, n, r, m)
    for _ in range(60):                                   # ~30 ms: the clock settles
        wl.compute()
    t = timed_loop(torch, wl.compute, steps)
    _lib.call("pmt_profile_enable", 1)
    for _ in range(10):
        wl.compute()
    torch.cuda.synchronize()
    kern = profile_report(_lib)
    _lib.call("pmt_profile_enable", 0)
    off, L = batch.slab_layout(n, m)
    nbytes = (8.0 * (r * n + r + m * n + m) + 8.0 * L) * total
    k = kern.get("batch_small_kernel")
    out = {"workload": "C4 on one GPU: 8192 independent QPs n=r=128, m=16; one coefficient slab per instance (no collective at N=1)",
           "ms_per_step": t / steps * 1e3, "re_evaluations_per_s": total * steps / t, "kernels": kern}
    if k:
        out["roofline"] = hbm_roofline("batch_small_kernel", k["avg_ms"], nbytes, "pmt::batch_small_kernel",
                                       mfma_frac=(total * 128.0 * 128 * 129 / (k["avg_ms"] * 1e-3) / 1e12) / F64_MFMA_PEAK_TFLOPS,
                                       note="both bounds are ~0.3 ms for this step (1.9 GB of HBM traffic; 8192 x 2.1 MFLOP on the f64 matrix pipe)")
    return out


def config_c5(torch, P, _lib, steps):
    """C5 at two boundaries.  `ms_per_step` is DEVICE-RESIDENT like the headline: nzval and d are regenerated on the device (the fills are
    part of the step), the MOI triplets stay in HBM.  Beside it the host-updated form (`val=` Parameters: 27 MB cross PCIe per update),
    serial and staged — BOTH reported: the copy is 0.5 ms against 0.04 ms of kernels, so there is nothing for a staged copy to hide behind
    and the two differ by the cost of their host calls only."""
    from parametron_jl_amd import workloads
    out = {}
    model, Cs = workloads.config5(device_resident=True, handoff="device")
    P.solve(model)
    ctx = model.device()
    out["workload"] = "C5: sparse C (5 %%, %d non-zeros, fixed pattern), n=16384, m=4096; device-resident: nzval and d regenerated on the device each step" % Cs.nnz

    def resident():
        model.update(synchronize=False)
    # (at least 40 ms of warm-up: 20 calls of this 41 us step are 0.8 ms, and the chip's clocks take ~30 ms of load to settle)
    t0 = time.perf_counter()
    k = 0
    while k < 20 or time.perf_counter() - t0 < 0.04:
        resident(); k += 1
        if k % 50 == 0:
            ctx.synchronize()
    ctx.synchronize()
    t = timed_loop(torch, resident, max(steps, 500))
    steps = max(steps, 500)
    out["ms_per_step"] = t / steps * 1e3
    out["re_evaluations_per_s"] = steps / t
    out["boundary"] = "device-resident (Parameter values made in HBM, MOI buffers left in HBM), as the headline"
    _lib.call("pmt_profile_enable", 1)
    for _ in range(20):
        resident()
    ctx.synchronize()
    kern = profile_report(_lib)
    _lib.call("pmt_profile_enable", 0)
    out["kernels"] = kern
    name = next((k for k in kern if k.startswith("sparse_")), None)
    if name:
        # per non-zero: coefficient read (8) + static index streams read (4 + 4) + VectorAffineTerm written (24); + row pointers
        out["roofline"] = hbm_roofline(name, kern[name]["avg_ms"], 40.0 * Cs.nnz, "pmt::sparse_")
    model.close()

    model, Cs = workloads.config5(pinned=True, handoff="device")
    P.solve(model)
    ctx = model.device()

    def serial():
        model.update(synchronize=False)

    def staged():
        model.stage_parameters()
        model.update(synchronize=False)
    hu = {"what": "nzval and d host-updated val= Parameters: 27 MB over PCIe per update (0.5 ms at 54 GB/s) in front of 0.04 ms of kernels"}
    for name, fn in (("serial_upload", serial), ("staged_upload", staged)):
        for _ in range(10):
            fn()
        ctx.synchronize()
        t = timed_loop(torch, fn, steps)
        model.wait_staged()
        hu[name] = {"ms_per_step": t / steps * 1e3, "re_evaluations_per_s": steps / t}
    out["host_updated"] = hu
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
# CPU baselines (rank 0, N = 1)

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
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--steps", str(steps),
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


def cpu_baseline(wl):
    """The reference's literal CPU path restated in C (oracle/, single thread like the reference), timed on this box's host cores.
    BASELINE.md §2: affine nodes and the constraint MOI copy at full size; the literal quadratic expansion + MOI copy at
    n = r in {64, 128, 256, 512}, fitted as c*n^3 and extrapolated to n = 4096 (the full literal objective is 1.65 TB and cannot be
    materialised); cross-check: 8 of the 4096 residual rows at full width, extrapolated x512.  About 5 s of single-core work in all."""
    import numpy as np
    from oracle import oracle as O
    n, r, m = wl.n, wl.r, wl.m
    A = O.fill_uniform(r * n, 1)
    b = O.fill_uniform(r, 2)
    Cm = O.fill_uniform(m * n, 3)
    d = O.fill_uniform(m, 4, 2.0)
    xvar = np.arange(1, n + 1, dtype=np.int64)
    w = O.LsqWorkspace(n, r, m)
    rows = 8                                 # 8 x 4096^2 literal terms = 3.2 GB (+ the same again as MOI terms)

    def affine_part():
        w.eval_residual(A, b, xvar)
        w.eval_residual(A, b, xvar)          # the reference evaluates `residual` twice (no memoisation, lazyexpression.jl:53-61)
        w.eval_constraint(Cm, d, xvar)
        return w.constraint.moi(xvar)

    def quad_part():
        w.eval_vecdot(rows)
        return w.objective.moi(xvar)

    affine_part(); quad_part()               # first touch: the reference's first solve! allocates, later ones do not
    t0 = time.perf_counter(); affine_part(); t_aff = time.perf_counter() - t0
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        quad_part()
    t_quad_rows = (time.perf_counter() - t0) / reps
    cross_check = t_aff + t_quad_rows * (r / rows)
    del w
    # c * n^3 fit of the literal quadratic node (n^3 terms at r = n)
    sizes, times = [64, 128, 256, 512], []
    for k in sizes:
        Ak, bk = O.fill_uniform(k * k, 1), O.fill_uniform(k, 2)
        xk = np.arange(1, k + 1, dtype=np.int64)
        wk = O.LsqWorkspace(k, k, 1)
        wk.eval_residual(Ak, bk, xk)
        wk.eval_vecdot(-1); wk.objective.moi(xk)                 # first touch
        best = float("inf")
        for _ in range(3):                                        # best of 3: the host is shared with the driver's own processes
            t0 = time.perf_counter()
            wk.eval_vecdot(-1); wk.objective.moi(xk)
            best = min(best, time.perf_counter() - t0)
        times.append(best)
        del wk
    n3 = np.array([float(k) ** 3 for k in sizes])
    c = float(np.dot(n3, times) / np.dot(n3, n3))                 # least squares through the origin
    t_quad_fit = c * float(n) ** 3
    total = t_aff + t_quad_fit
    return {"value": 1.0 / total, "unit": "re-evaluations/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
            "sample": "C restatement of the reference loops (oracle/), 1 thread: matvecmul!+vecsubtract! x2, constraint C*x-d and its MOI copy at full "
                      "size (%.3f s); literal _vecdot!/muladd! expansion + MOI copy at n = r = 64, 128, 256, 512 (%s s), fitted c*n^3 with c = %.3e s and "
                      "EXTRAPOLATED to n = 4096 (%.1f s) — the full literal objective is 1.65 TB and cannot be materialised"
                      % (t_aff, ", ".join("%.4f" % t for t in times), c, t_quad_fit),
            "seconds_per_reevaluation_extrapolated": total,
            "cross_check_rows": {"seconds_per_reevaluation_extrapolated": cross_check, "re_evaluations_per_s": 1.0 / cross_check,
                                     "sample": "%d of %d residual rows at full width (%.3f s), x%d" % (rows, r, t_quad_rows, r // rows)}}


def cpu_canonical_blas(wl):
    """'Best CPU' line of BASELINE.md §2 (CPU-canonical): the CANONICAL result (what the GPU path produces) computed on all host
    cores with multithreaded BLAS (numpy: 2*A'A as one dgemm, -2A'b, and vectorised packing into the MOI term arrays).  Not the
    reference's algorithm — the reference has no threading and never combines terms — reported next to the literal port so that
    the GPU/CPU ratio can be read against a strong CPU implementation too."""
    import numpy as np
    from oracle import oracle as O
    n, r, m = wl.n, wl.r, wl.m
    A = O.fill_uniform(r * n, 1).reshape(n, r).T          # column-major view (r, n)
    b = O.fill_uniform(r, 2)
    Cm = O.fill_uniform(m * n, 3).reshape(n, m).T
    d = O.fill_uniform(m, 4, 2.0)
    iu = np.triu_indices(n)
    QT = np.dtype([("coeff", "<f8"), ("row", "<i8"), ("col", "<i8")])
    VAT = np.dtype([("out", "<i8"), ("coeff", "<f8"), ("var", "<i8")])
    q = np.empty(len(iu[0]), dtype=QT); q["row"] = iu[0] + 1; q["col"] = iu[1] + 1
    v = np.empty(m * n, dtype=VAT); v["out"] = np.repeat(np.arange(1, m + 1), n); v["var"] = np.tile(np.arange(1, n + 1), m)

    def once():
        G = A.T @ A
        q["coeff"] = 2.0 * G[iu]
        lin = -2.0 * (A.T @ b)
        v["coeff"] = np.ascontiguousarray(Cm).reshape(-1)
        return lin, float(b @ b), 0.0 - d
    once()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    return {"value": 1.0 / dt, "unit": "re-evaluations/s", "cores": os.cpu_count(), "kind": "numpy/BLAS canonical (not the reference algorithm)",
            "seconds_per_reevaluation": dt}


def run_batch(args, torch, dist, _lib, rank, world):
    from parametron_jl_amd import batch
    return batch.bench(args, torch, dist, _lib, rank, world, emit=emit_line)


def guarded(fn, *a):
    try:
        return fn(*a)
    except Exception as e:                       # a failing side section must not take the headline line with it
        return {"error": "%s: %s" % (type(e).__name__, e)}


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too (RCCL prints a five-line version banner through C
    stdio when its first communicator is built — flushed at exit, i.e. BEHIND the JSON line): file descriptor 1 is pointed at stderr for
    the life of the process and the line goes to a duplicate of the original descriptor."""
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
    """the command `--gpus N` (N > 1) re-executes itself as when no launcher has set RANK: the driver's own shape, one rank per GPU of this node"""
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


def main():
    args = parse_args()
    claim_stdout()
    if args.gpus < 1:
        fail("--gpus must be >= 1")
    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_launch):
        return self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        # spin-up and warm-up steps, not in the barrier that brackets the timed region — behind it the first timed steps ran 8 % slow while
        # the clocks came back (20 steps under torchrun: 746/s against 810/s without)
        dist.barrier()
        torch.cuda.synchronize()
    if args.workload == "batch":
        return run_batch(args, torch, dist, _lib, rank, world)

    ranks_seen = count_ranks(torch, dist, "cuda")
    if ranks_seen != args.gpus:
        if rank == 0:
            fail("ranks_seen %d != --gpus %d" % (ranks_seen, args.gpus), ranks_seen=ranks_seen)
        os._exit(2)
    wl = C2Workload(torch, _lib, rank, side_lane=args.side_lane != "off", background=args.side_lane == "background", pack_first=args.pack_first)
    if args.graph:
        _lib.call("pmt_plan_instantiate_graph", wl.plan)
    # A step = update!(model) of the reference (src/model.jl:132-143): setdirty! + the Parameter callbacks (src/parameter.jl:93-102; here the
    # device-side rand! of A, b, C, d — 151 MB regenerated in HBM) + the rebuild of every MOI buffer.  Nothing crosses PCIe in the step.
    step = wl.step_with_refresh

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
    # the rebuild alone (Parameter values already in HBM, callbacks outside): the same K steps, bracketed the same way
    elapsed_resident = bracketed(wl.step, args.steps)
    kernels = {}
    stamps = None
    if not args.graph:
        # the DOMINANT kernel's roofline: a second pass of K identical steps with a HIP-event pair (on the stream the kernel is launched on)
        # around its launches only — every event pair costs queue time, so it stays out of the timed region above
        _lib.call("pmt_profile_filter", b"gram_sk_kernel")
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
        others = profile_report(_lib)
        _lib.call("pmt_profile_enable", 0)
        for k, v in others.items():
            if k not in kernels:
                kernels[k] = dict(v, measured="HIP events around every kernel, separate 20-step pass")
        if rank == 0:
            stamps = guarded(pack_in_step_stamps, torch, _lib, wl, step, max(20, min(args.steps, 100)))

    if rank == 0:
        out = {
            "metric": "QP re-evaluations/sec (Q,q,C,d rebuild) at n=4096",
            "value": value, "unit": "re-evaluations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name, "n": wl.n, "r": wl.r, "m": wl.m, "objective_mode": "canonical",
                       "instances_per_gpu": 1, "parallelism": "replicas (independent QP instances, no collective)" if world > 1 else "single GPU",
                       "replay": "hipGraph" if args.graph else "tape", "setup_spinup_steps": wl.SPINUP_STEPS,
                       "device_lda": [wl.lda, wl.ldc],
                       "step": "update!(model): setdirty! + the four Parameter callbacks (device-side rand! of A, b, C, d) + the rebuild of Q, q, const, C, d",
                       "value_inputs_resident": world * wl.units_per_step * args.steps / elapsed_resident,
                       "ms_per_step_inputs_resident": elapsed_resident / args.steps * 1e3,
                       "boundary": "device-resident hand-off: the Parameter callbacks write HBM, MOI buffers left in HBM; nothing crosses PCIe "
                                   "(PCIe-inclusive rates: host_api).  value_inputs_resident = the rebuild alone, callbacks outside the timed loop"},
            "step_algorithmic_bytes": wl.step_bytes(), "step_flops": wl.gram_flops(), "ranks_seen": ranks_seen,
        }
        g = kernels.get("gram_sk_kernel")
        out["roofline"] = mfma_roofline("gram_sk_kernel", g["avg_ms"], wl.gram_flops(), "pmt::gram_sk_kernel<2, 16, 2, 0, false>") if g else None
        v = kernels.get("affine_tile_kernel<VAT>")
        bg = kernels.get("affine_pack_background_kernel")
        if v:
            # Back-to-back stand-alone launches of the step's own kernel on the step's own buffers.  Between the events of an in-step launch lies
            # the in-stream gap behind the fix-up pass as well (the start event completes when the previous kernel does): that figure is kept
            # beside it; rocprofv3 (profiles/) times the in-step kernel itself at 13-14 us.
            rc = guarded(constraint_pack_microbench, torch, _lib, wl)
            if isinstance(rc, dict) and "avg_ms" in rc:
                nb = rc.get("algorithmic_bytes") or 32.0 * wl.m * wl.n
                # the same kernel where it runs — between the contraction's fix-up pass and the end of the step.  Three clocks, all labelled:
                #   device_clock  the kernel's own first-start .. last-end, measured in THIS run (pmt_profile_kernel_stamps)
                #   hip_events    HIP events around the in-step launch, THIS run; they include the in-stream gap in front of it (lower bound)
                #   rocprofv3     kernel-trace stamps: of a child of this invocation when rocprofv3 is on PATH, else `rocprofv3_replayed`
                #                 from the committed profiles/rocprof_in_step.json (NOT measured in this run)
                ins = {"hip_events": {"avg_ms": v["avg_ms"], "achieved": nb / (v["avg_ms"] * 1e-3) / 1e9, "unit": "GB/s",
                                      "frac": nb / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "measured_in_this_run": True,
                                      "note": "includes the in-stream gap behind the fix-up pass"}}
                if isinstance(stamps, dict):
                    ins["device_clock"] = stamps
                child = None if (args.no_rocprof_child or world > 1) else guarded(rocprof_child)
                ck = (child or {}).get("affine_tile_kernel<VAT>") if isinstance(child, dict) else None
                if ck:
                    ins["rocprofv3"] = {"avg_ms": ck["avg_us"] * 1e-3, "achieved": nb / (ck["avg_us"] * 1e-6) / 1e9, "unit": "GB/s",
                                        "frac": nb / (ck["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "measured_in_this_run": True, "source": child.get("source")}
                    out["rocprofv3_child"] = child
                elif isinstance(child, dict) and "error" in child:
                    out["rocprofv3_child"] = child
                rp = rocprof_in_step("affine_tile_kernel<VAT>")
                if rp:
                    ins["rocprofv3_replayed"] = {"avg_ms": rp["avg_us"] * 1e-3, "achieved": nb / (rp["avg_us"] * 1e-6) / 1e9, "unit": "GB/s",
                                                 "frac": nb / (rp["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "measured_in_this_run": False,
                                                 "source": "replayed: profiles/rocprof_in_step.json (%s)" % rp.get("source", "rocprofv3 --kernel-trace")}
                rc["in_step"] = ins
                rc["note"] = "stand-alone launches of the step's constraint-pack kernel; in_step = the launch inside the step, by three clocks"
            out["roofline_constraint_pack"] = rc
        elif bg:
            # side lane: the constraint block is packed by the <= 16-VGPR background kernel INSIDE the contraction; its duration is time spent
            # beside the Gram kernel, not on the step's critical path, so a bandwidth fraction of it would mean nothing
            out["constraint_pack"] = {"kernel": "affine_pack_background_kernel", "avg_ms": bg["avg_ms"], "algorithmic_bytes": 32.0 * wl.m * wl.n,
                                      "placement": "side lane, co-resident with gram_sk_kernel (pmt_plan_set_lane); the stand-alone tile kernel "
                                                   "of the same node is measured in roofline_constraint_pack"}
            out["roofline_constraint_pack"] = guarded(constraint_pack_microbench, torch, _lib, wl)
        out["kernels"] = kernels
    if world == 1 and not args.graph:
        if args.steps < 200:
            t = timed_loop(torch, step, 200)
            out["value_200_steps"] = {"value": 200 / t, "ms_per_step": t / 200 * 1e3, "steps": 200,
                                      "what": "the same step timed over 200 steps (> 0.2 s of device time) right after the K-step measurement"}
        out["roofline_affine"] = guarded(affine_microbench, torch, _lib, wl)
        wl.close()          # the plan and its side stream go before the other configurations create theirs (streams share hardware queues)
        if not args.no_configs:
            ksteps = max(50, min(args.steps, 100))      # (at least 50 steps: 20 steps of a 1.3 ms configuration are 26 ms, too short to average out a box hiccup)
            out["configs"] = {"C1": guarded(config_c1, torch, P, _lib), "C3": guarded(config_c3, torch, P, _lib, ksteps),
                              "C4": guarded(config_c4, torch, _lib, ksteps), "C5": guarded(config_c5, torch, P, _lib, ksteps),
                              "tall": guarded(config_tall, torch, _lib), "mid": guarded(config_mid, torch, P)}
            out["host_api"] = guarded(host_api_c2, torch, P, 10)
        out["cpu_baseline"] = None if args.no_cpu_baseline else guarded(cpu_baseline, wl)
        out["cpu_canonical_blas"] = None if args.no_cpu_baseline else guarded(cpu_canonical_blas, wl)
    else:
        wl.close()
    if not args.no_configs:
        # every rank: BASELINE config 4 sharded by instance over the N ranks (N = 1: the same code path, single-rank communicator)
        from parametron_jl_amd import batch
        ksteps = max(50, min(args.steps, 100))      # (at least 50 steps: 20 steps of a 1.3 ms configuration are 26 ms, too short to average out a box hiccup)
        # This section has only ever run with one rank on hardware.  Should a collective of it hang with N > 1 (one rank failing alone), the
        # headline — complete at this point — must still come out: after 150 s rank 0 prints the line without the section and every rank exits.
        import threading

        def give_up():
            if rank == 0:
                out.setdefault("configs", {})["C4_sharded"] = {"error": "timeout: the sharded section did not finish within 150 s"}
                emit_line(ordered_for_the_tail(out))
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
        emit_line(ordered_for_the_tail(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def summary_of(out):
    """the numbers a reader of the LAST characters of the line needs (a log that keeps only the tail of stdout still shows them)"""
    c = out.get("configs") or {}
    h = out.get("host_api") or {}
    pack = out.get("roofline_constraint_pack") or {}
    return {
        "value": out.get("value"), "ms_per_step": out.get("ms_per_step"), "gram_frac": _get(out, "roofline", "frac"),
        "gram_avg_ms": _get(out, "roofline", "avg_ms"),
        "C3_ms": _get(c, "C3", "ms_per_step"), "C4_ms": _get(c, "C4", "ms_per_step"), "C4_frac": _get(c, "C4", "roofline", "frac"),
        "C5_ms": _get(c, "C5", "ms_per_step"), "C5_frac": _get(c, "C5", "roofline", "frac"),
        "C4_sharded_ms": _get(c, "C4_sharded", "ms_per_step"), "rccl_calls_made": _get(c, "C4_sharded", "rccl_calls_made"),
        "device_ms": _get(h, "handoff_device", "ms_per_solve"), "host_csc_ms": _get(h, "handoff_host_csc", "ms_per_solve"),
        "moi_ms": _get(h, "handoff_moi", "ms_per_solve"), "c3_host_csc_ms": _get(h, "c3_host_csc", "ms_per_solve"),
        "pack_standalone_frac": pack.get("frac") if isinstance(pack, dict) else None,
        # measured in this run only: the device-clock figure, else the HIP-event lower bound; the replay of a committed profile has its own key
        "pack_in_step_frac": _get(pack, "in_step", "device_clock", "frac") or _get(pack, "in_step", "rocprofv3", "frac") or _get(pack, "in_step", "hip_events", "frac"),
        "pack_in_step_frac_rocprof_replayed": _get(pack, "in_step", "rocprofv3_replayed", "frac"),
        "value_inputs_resident": _get(out, "config", "value_inputs_resident"), "mid_solve_us": {k: round(v["solve_us"], 1) for k, v in (c.get("mid") or {}).items() if isinstance(v, dict) and "solve_us" in v} or None,
        "C1_us": _get(c, "C1", "update_us"), "C1_solve_us": _get(c, "C1", "solve_us_python_host_mock_optimizer"), "C1_kernel_us": _get(c, "C1", "kernel_us", "small_plan_kernel"),
        "tall_frac": {k: round(v["frac"], 3) for k, v in (c.get("tall") or {}).items()
                      if isinstance(v, dict) and "frac" in v and v.get("binding") != "launch latency"} or None,
        "node_us_launch_bound": {k: round(v["node_ms"] * 1e3, 1) for k, v in (c.get("tall") or {}).items()
                                 if isinstance(v, dict) and v.get("binding") == "launch latency"} or None,
        "affine_warm_frac": _get(out, "roofline_affine", "frac"), "affine_cold_frac": _get(out, "roofline_affine", "cold", "frac"),
        "cpu_baseline_value": _get(out, "cpu_baseline", "value"), "ranks_seen": out.get("ranks_seen"),
    }


def ordered_for_the_tail(out):
    """Same object, keys re-ordered: the bulky sections (per-kernel tables, the other configurations with their prose) first, then the
    contract's own fields, the CPU-baseline and roofline objects, and a compact `summary` LAST."""
    bulky = ("kernels", "configs", "host_api", "cpu_canonical_blas", "roofline_affine", "roofline_constraint_pack", "constraint_pack",
             "rocprofv3_child", "value_200_steps")
    last = ("config", "cpu_baseline", "roofline")        # (the driver's text tail also carries a few lines of stderr: the shortest objects go last)
    res = {}
    for k in bulky:
        if k in out:
            res[k] = out[k]
    for k, v in out.items():
        if k not in bulky and k not in last:
            res[k] = v
    for k in last:
        if k in out:
            res[k] = out[k]
    res["summary"] = summary_of(out)
    return res


if __name__ == "__main__":
    main()
