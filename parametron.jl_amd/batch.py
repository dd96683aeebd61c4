"""Batched independent QPs sharded by instance across GPUs (BASELINE config 4).

In the reference a batch is many independent `Model`s (src/model.jl:1-22).  All instances share one structure, so a
re-evaluation produces one coefficient slab per instance (pmt_batch_lsq_coeffs_f64); rank g owns the contiguous instance
range shard_range(B, g, world) and ONE all-gather (RCCL over xGMI via torch.distributed) assembles every slab on every rank.
There is no other collective: the instances never interact.
"""
import ctypes as C
import json
import time

import numpy as np

from . import _lib
from ._lib import ArgumentError


def shard_range(total, rank, world):
    """Contiguous instance range [lo, hi) owned by `rank` (pmt_batch_shard): equal shards, or DimensionMismatch — the exchange lays the
    gathered buffer out as world * per_rank slabs, an uneven split would corrupt it silently."""
    per, first = C.c_int64(), C.c_int64()
    _lib.call("pmt_batch_shard", total, world, rank, C.byref(per), C.byref(first))
    return first.value, first.value + per.value


def slab_layout(n, m):
    """Offsets (in doubles) of the sections of one instance's slab and the slab length."""
    nq = n * (n + 1) // 2
    off = {"Q": 0, "q": nq, "const": nq + n, "C": nq + n + 1, "dconst": nq + n + 1 + m * n}
    return off, nq + n + 1 + m * n + m


def gather_slabs(dist, local, gathered):
    """all-gather of equally sized per-rank slab blocks into `gathered` ([world * B_local, L]); rank order = instance order."""
    if dist is None:
        gathered.copy_(local)
        return
    dist.all_gather_into_tensor(gathered.view(-1), local.contiguous().view(-1))


def chunk_schedule(per_rank, chunk):
    """The chunk schedule of pmt_batch_step_f64 / pmt_batch_allgather_f64 (the library's own host arithmetic, include/parametron_hip.h):
    [(lo, hi)] local instance ranges in exchange order; chunk <= 0 means one chunk."""
    L = _lib.load()
    out = []
    for c in range(int(L.pmt_batch_num_chunks(per_rank, chunk))):
        lo, hi = C.c_int64(), C.c_int64()
        _lib.call("pmt_batch_chunk_range", per_rank, chunk, c, C.byref(lo), C.byref(hi))
        out.append((lo.value, hi.value))
    return out


def gathered_offset(rank, per_rank, local_instance, stride):
    """position (in doubles) of a rank's local instance in the gathered buffer (global instance order)"""
    return int(_lib.load().pmt_batch_gathered_offset(rank, per_rank, local_instance, stride))


def exchange_chunks(dist, local, gathered, rank, world, chunk):
    """The library's exchange schedule driven through a torch.distributed backend that has no RCCL (the 2-rank gloo test on CPU): for
    every chunk, in order, one send / receive pair per peer (peers staggered as in comm.hip) and the local copy; `local` is
    [per_rank, stride], `gathered` [world * per_rank, stride]."""
    per_rank, stride = local.shape
    flat_l, flat_g = local.reshape(-1), gathered.reshape(-1)
    for lo, hi in chunk_schedule(per_rank, chunk):
        cnt = (hi - lo) * stride
        mine = gathered_offset(rank, per_rank, lo, stride)
        flat_g[mine:mine + cnt] = flat_l[lo * stride:lo * stride + cnt]
        if dist is None or world == 1:
            continue
        ops = []
        for k in range(1, world):
            to, frm = (rank + k) % world, (rank - k + world) % world
            off = gathered_offset(frm, per_rank, lo, stride)
            ops.append(dist.P2POp(dist.isend, flat_l[lo * stride:lo * stride + cnt].contiguous(), to))
            ops.append(dist.P2POp(dist.irecv, flat_g[off:off + cnt], frm))
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class Communicator:
    """pmt_comm_*: the library's own RCCL communicator.  The launcher (torch.distributed here) only carries rank 0's unique id."""

    def __init__(self, torch, dist, rank, world, device_index, rccl_single=False):
        """rccl_single: a single rank still builds a real (one-rank) RCCL communicator and exchanges with itself — the N-rank code path on
        one GPU; False: a single rank loads no RCCL at all"""
        self.rank, self.world = rank, world
        self.handle = C.c_void_p()
        if world > 1:
            ident = (C.c_char * 128)()
            if rank == 0:
                _lib.call("pmt_comm_unique_id", C.cast(ident, C.c_void_p))
            t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device="cuda")
            dist.broadcast(t, src=0)
            ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
            _lib.call("pmt_comm_init_rank", world, rank, C.cast(ident, C.c_void_p), device_index, C.byref(self.handle))
        elif rccl_single:
            ident = (C.c_char * 128)()
            _lib.call("pmt_comm_unique_id", C.cast(ident, C.c_void_p))
            _lib.call("pmt_comm_init_rank", 1, 0, C.cast(ident, C.c_void_p), device_index, C.byref(self.handle))
        else:
            _lib.call("pmt_comm_init_rank", 1, 0, None, device_index, C.byref(self.handle))

    def rccl_calls(self):
        return int(_lib.load().pmt_comm_rccl_calls(self.handle))

    def close(self):
        if self.handle:
            _lib.call("pmt_comm_destroy", self.handle)
            self.handle = C.c_void_p()


class BatchLSQ:
    """Device-resident batch: A[B][r x n], b[B][r], C[B][m x n], d[B][m] generated by the counter-based stream so that
    instance k holds the same data whatever the sharding (seed streams are indexed by GLOBAL element index)."""

    SEEDS = {"A": 101, "b": 102, "C": 103, "d": 104}

    def __init__(self, torch, total, n, r, m, rank=0, world=1, device=None):
        _lib.require_gpu()
        self.torch, self.total, self.n, self.r, self.m, self.rank, self.world = torch, total, n, r, m, rank, world
        self.lo, self.hi = shard_range(total, rank, world)
        self.B = self.hi - self.lo
        dev = device or torch.device("cuda", torch.cuda.current_device())
        f64 = torch.float64
        self.A = torch.empty(self.B * r * n, dtype=f64, device=dev)
        self.b = torch.empty(self.B * r, dtype=f64, device=dev)
        self.Cm = torch.empty(self.B * m * n, dtype=f64, device=dev)
        self.d = torch.empty(self.B * m, dtype=f64, device=dev)
        self.offsets, self.L = slab_layout(n, m)
        self.local = torch.empty((self.B, self.L), dtype=f64, device=dev)
        self.gathered = torch.empty((total, self.L), dtype=f64, device=dev) if world > 1 else self.local
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.regenerate(0)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def regenerate(self, epoch):
        """device-side Parameter callbacks (↔ rand! in README.md:36-43), one stream per Parameter and epoch"""
        n, r, m = self.n, self.r, self.m
        for name, buf, per, scale in (("A", self.A, r * n, 1.0), ("b", self.b, r, 1.0), ("C", self.Cm, m * n, 1.0), ("d", self.d, m, 2.0)):
            _lib.call("pmt_fill_uniform_offset_f64", self._p(buf), self.B * per, C.c_uint64(self.SEEDS[name] + 1000 * epoch),
                      C.c_uint64(self.lo * per), scale, self.stream)

    def compute(self):
        _lib.call("pmt_batch_lsq_coeffs_f64", self._p(self.A), self._p(self.b), self._p(self.Cm), self._p(self.d), self.B, self.n, self.r,
                  self.m, -1, -1, self._p(self.local), self.L, self.stream)

    def gather(self, dist):
        if self.world > 1:
            gather_slabs(dist, self.local, self.gathered)

    def step(self, dist):
        self.compute()
        self.gather(dist)

    def step_pipelined(self, comm, chunk):
        """pmt_batch_step_f64: compute chunk by chunk, chunk c on the wire (the library's RCCL communicator, its own stream) while
        chunk c + 1 is computed"""
        _lib.call("pmt_batch_step_f64", comm.handle, self._p(self.A), self._p(self.b), self._p(self.Cm), self._p(self.d), self.B, self.n, self.r,
                  self.m, -1, -1, self._p(self.local), self._p(self.gathered), self.L, int(chunk), self.stream)


TOTAL, N, R, M = 8192, 128, 128, 16        # BASELINE config 4 (m is not stated there: 16)


def sharded_report(world, ranks_seen, steps, warmup, t_pipe, t_mono, t_compute, nchunks, total=TOTAL, n=N, r=R, m=M):
    """The JSON object of the sharded C4 step (pure arithmetic on the three measured times: testable without a GPU).
    value = the whole job's instances per second with the exchange overlapped; beside it compute-only and computation + one monolithic
    all-gather, the bytes every rank puts on the wire and what a per-link-bound exchange of them would cost."""
    off, L = slab_layout(n, m)
    per_inst_bytes = 8.0 * (r * n + r + m * n + m) + 8.0 * L
    per_gpu = per_inst_bytes * (total // world)
    exch = 8.0 * L * (total // world)
    return {
        "metric": "QP re-evaluations/sec (Q,q,C,d rebuild), batched n=128", "value": total * steps / t_pipe,
        "unit": "re-evaluations/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": steps, "warmup": warmup,
        "ms_per_step": t_pipe / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C4 batch of 8192 independent QPs n=r=128 m=16, sharded by instance; value = computation + chunked "
                               "exchange of the coefficient slabs overlapped (pmt_batch_step_f64: RCCL send/recv per peer behind the C ABI)",
                   "instances": total, "instances_per_gpu": total // world, "slab_doubles": L, "exchange_chunks_per_rank": nchunks,
                   "parallelism": "instance sharding + direct (per-peer) exchange over xGMI" if world > 1 else "single GPU (single-rank communicator, same code path)"},
        "compute_only": {"value": total * steps / t_compute, "ms_per_step": t_compute / steps * 1e3},
        "compute_then_allgather": {"value": total * steps / t_mono, "ms_per_step": t_mono / steps * 1e3,
                                   "what": "one torch.distributed all_gather_into_tensor behind the computation (round 1's path)"},
        "compute_overlapped_exchange": {"value": total * steps / t_pipe, "ms_per_step": t_pipe / steps * 1e3},
        "exchange_bytes_per_rank": exch, "step_algorithmic_bytes_per_gpu": per_gpu,
        "expected_exchange_ms": {"direct_per_link": exch / 153e9 * 1e3 if world > 1 else 0.0,
                                 "ring": (world - 1) * exch / 153e9 * 1e3 if world > 1 else 0.0,
                                 "note": "xGMI ~153 GB/s per link, 7 links per GPU (MI355X_MICROARCH.md): direct = every peer over its own link"},
        "roofline": {"kernel": "batch_small_kernel", "bound": "hbm", "achieved": per_gpu / (t_compute / steps) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": per_gpu / (t_compute / steps) / 1e9 / 8000.0, "traffic": None,
                     "note": "compute-only step (one batch_small_kernel launch per step); both bounds of the step are ~0.3 ms at 1 GPU"},
        "cpu_baseline": None,
    }


def measure(torch, dist, rank, world, steps, warmup):
    """The sharded C4 step on `world` ranks (every rank calls this; world == 1: the same code path with the single-rank communicator).
    Times K steps three ways — exchange overlapped (pmt_batch_step_f64), monolithic all-gather behind the computation, computation only —
    each bracketed by barrier + synchronize, MAX over ranks.  Returns the report on rank 0, None elsewhere."""
    wl = BatchLSQ(torch, TOTAL, N, R, M, rank, world)
    comm = Communicator(torch, dist, rank, world, torch.cuda.current_device())
    # one rank: nothing to exchange, one launch.  Several ranks: the first exchange starts after 1/8 of the computation, but a chunk never
    # has fewer instances than the GPU has CUs (batch_small_kernel is one persistent workgroup per CU)
    chunk = wl.B if world == 1 else max(256, wl.B // 8)

    def timed(fn):
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

    for _ in range(max(warmup, 60)):                       # >= 30 ms of work: the clock settles before anything is timed
        wl.compute()
    for _ in range(3):
        wl.step(dist)
        wl.step_pipelined(comm, chunk)
    t_pipe = timed(lambda: wl.step_pipelined(comm, chunk))
    t_mono = timed(lambda: wl.step(dist))
    t_compute = timed(wl.compute)
    ranks_seen = world
    if dist:
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    nchunks = len(chunk_schedule(wl.B, chunk))
    rccl_calls = comm.rccl_calls()
    comm.close()
    rep = sharded_report(world, ranks_seen, steps, warmup, t_pipe, t_mono, t_compute, nchunks) if rank == 0 else None
    if world == 1:
        # One GPU: the value above loads no RCCL (a single rank has nothing to exchange).  The library's RCCL binding is exercised all the
        # same: a real ONE-RANK communicator (ncclGetUniqueId / ncclCommInitRank) whose chunks travel to itself through grouped ncclSend /
        # ncclRecv on the communication stream — the N-rank code path on this GPU, checked against the plain result and timed beside it.
        try:
            want = wl.local.clone()
            c1 = Communicator(torch, None, 0, 1, torch.cuda.current_device(), rccl_single=True)
            wl.gathered = torch.full_like(wl.local, float("nan"))
            ch = max(256, wl.B // 8)
            for _ in range(3):
                wl.step_pipelined(c1, ch)
            torch.cuda.synchronize()
            ok = bool(torch.equal(wl.gathered, want) and torch.equal(wl.local, want))
            t_self = timed(lambda: wl.step_pipelined(c1, ch))
            rccl_calls = c1.rccl_calls()
            c1.close()
            rep["rccl_self_exchange"] = {"ms_per_step": t_self / steps * 1e3, "chunks": len(chunk_schedule(wl.B, ch)), "gathered_equals_local": ok,
                                         "what": "one-rank RCCL communicator: every chunk sent to / received from this rank through RCCL while the next is computed"}
        except Exception as e:
            rep["rccl_self_exchange"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rep is not None:
        rep["rccl_calls_made"] = rccl_calls > 0
        rep["rccl_calls"] = rccl_calls
    return rep


def bench(args, torch, dist, lib_mod, rank, world, emit=None):
    """bench.py --workload batch: 8192 x (n = r = 128, m = 16) instances, strong scaling over ranks.  Reports the re-evaluation rate
    compute-only, with one monolithic all-gather behind the computation (torch.distributed), and with the library's chunked exchange
    overlapped with the computation (pmt_batch_step_f64) — BASELINE.md §3 asks for the gather included and excluded."""
    out = measure(torch, dist, rank, world, args.steps, args.warmup)
    if rank == 0:
        (emit or (lambda o: print(json.dumps(o), flush=True)))(out)        # (bench.py passes the writer that owns the real stdout)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
