"""CPU, world_size 2 over gloo: the N > 1 path of the batched workload — instance sharding and the single all-gather that
assembles the coefficient slabs.  The per-instance slabs are produced by the CPU oracle here (there is no GPU in this
container and the product has no CPU compute path); what is under test is batch.shard_range / batch.gather_slabs / slab_layout."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def oracle_slab(inst, n, r, m):
    from oracle import oracle as O
    from parametron_jl_amd import batch
    A = O.fill_uniform((inst + 1) * r * n, 101)[inst * r * n:].reshape(n, r).T
    b = O.fill_uniform((inst + 1) * r, 102)[inst * r:]
    Cm = O.fill_uniform((inst + 1) * m * n, 103)[inst * m * n:].reshape(n, m).T
    d = O.fill_uniform((inst + 1) * m, 104, 2.0)[inst * m:]
    xvar = np.arange(1, n + 1, dtype=np.int64)
    w = O.LsqWorkspace(n, r, m)
    w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
    w.eval_constraint(np.ascontiguousarray(Cm.T).reshape(-1), d, xvar)
    w.objective.canonicalize()
    at, qt, const = w.objective.moi()
    ct, cc = w.constraint.moi()
    off, L = batch.slab_layout(n, m)
    slab = np.empty(L)
    slab[:off["q"]] = qt["coeff"]
    slab[off["q"]:off["const"]] = at["coeff"]
    slab[off["const"]] = const
    slab[off["C"]:off["dconst"]] = ct["coeff"]
    slab[off["dconst"]:] = cc
    return slab


def _worker(rank, world, port, total, n, r, m, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import parametron_jl_amd  # noqa: F401
    from parametron_jl_amd import batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        lo, hi = batch.shard_range(total, rank, world)
        off, L = batch.slab_layout(n, m)
        local = torch.from_numpy(np.stack([oracle_slab(i, n, r, m) for i in range(lo, hi)]))
        gathered = torch.empty((total, L), dtype=torch.float64)
        batch.gather_slabs(dist, local, gathered)
        full = np.stack([oracle_slab(i, n, r, m) for i in range(total)])
        q.put((rank, lo, hi, bool(np.array_equal(gathered.numpy(), full))))
    finally:
        dist.destroy_process_group()


def _worker_chunked(rank, world, port, total, n, r, m, chunk, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import parametron_jl_amd  # noqa: F401
    from parametron_jl_amd import batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        lo, hi = batch.shard_range(total, rank, world)
        off, L = batch.slab_layout(n, m)
        local = torch.from_numpy(np.stack([oracle_slab(i, n, r, m) for i in range(lo, hi)]))
        gathered = torch.full((total, L), float("nan"), dtype=torch.float64)
        batch.exchange_chunks(dist, local, gathered, rank, world, chunk)           # the library's chunk schedule, chunk by chunk
        full = np.stack([oracle_slab(i, n, r, m) for i in range(total)])
        q.put((rank, batch.chunk_schedule(hi - lo, chunk), bool(np.array_equal(gathered.numpy(), full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,chunk,want", [(10, 2, [(0, 2), (2, 4), (4, 5)]), (10, 5, [(0, 5)]), (10, 0, [(0, 5)]), (14, 3, [(0, 3), (3, 6), (6, 7)])])
def test_two_rank_chunked_exchange_follows_the_library_schedule(total, chunk, want):
    """The N > 1 path of pmt_batch_step_f64 on CPU: the chunk schedule and the gathered offsets are the LIBRARY's own host functions
    (pmt_batch_num_chunks / pmt_batch_chunk_range / pmt_batch_gathered_offset, the same calls comm.hip makes), driven here over gloo with
    one send / receive pair per peer and chunk — chunk order, offsets, and an uneven last chunk."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n, r, m = 5, 4, 2
    procs = [ctx.Process(target=_worker_chunked, args=(rank, 2, port, total, n, r, m, chunk, q)) for rank in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r_[1] == want for r_ in results)                               # both ranks walk the same chunks in the same order
    assert all(r_[2] for r_ in results)                                       # every rank holds every slab, in global instance order


def test_chunk_schedule_edge_cases():
    from parametron_jl_amd import batch
    assert batch.chunk_schedule(0, 4) == []
    assert batch.chunk_schedule(1024, 128) == [(k * 128, (k + 1) * 128) for k in range(8)]
    assert batch.chunk_schedule(1000, 128)[-1] == (896, 1000)
    assert batch.chunk_schedule(5, 100) == [(0, 5)]
    assert batch.gathered_offset(3, 1024, 17, 10449) == (3 * 1024 + 17) * 10449


def test_two_rank_instance_sharding_and_allgather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, n, r, m = 6, 5, 4, 2
    procs = [ctx.Process(target=_worker, args=(rank, 2, port, total, n, r, m, q)) for rank in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r_[1], r_[2]) for r_ in results] == [(0, 3), (3, 6)]          # contiguous, disjoint, covering
    assert all(r_[3] for r_ in results)                                      # every rank holds every slab in instance order


def test_shard_range_partitions_any_world():
    """equal contiguous shards (pmt_batch_shard); a batch that does not divide over the ranks is a DimensionMismatch, never an uneven split
    (comm.hip lays the gathered buffer out as world * per_rank slabs)"""
    import pytest
    from parametron_jl_amd import batch, _lib
    for total in (0, 6, 24, 8192):
        for world in (1, 2, 3, 8):
            if total % world:
                with pytest.raises(_lib.DimensionMismatch):
                    batch.shard_range(total, 0, world)
                continue
            spans = [batch.shard_range(total, k, world) for k in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1))
            assert len({hi - lo for lo, hi in spans}) == 1
    for total, world in ((7, 2), (8192, 3), (1, 8)):
        with pytest.raises(_lib.DimensionMismatch):
            batch.shard_range(total, world - 1, world)
    with pytest.raises(_lib.ArgumentError):
        batch.shard_range(8, 2, 2)
    off, L = batch.slab_layout(128, 16)
    assert L == 8256 + 128 + 1 + 2048 + 16 and off["C"] == 8385
