"""-m gpu: batched instances (BASELINE config 4) — every instance's slab expands to the buffers the single-instance entry
points produce for that instance: indices, Q coefficients (same MFMA lane mapping and k order), the constant and the constraint
block bit for bit; q within 1e-12 (the fused small-instance kernel sums a column in a different order); and to numpy within 1e-12."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402


def test_batch_slabs_match_single_instance_path_and_oracle():
    import gpu_util as g
    from parametron_jl_amd import batch
    total, n, r, m = 12, 128, 96, 16
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.compute()
    torch.cuda.synchronize()
    off, L = batch.slab_layout(n, m)
    assert L == g.lib().pmt_batch_lsq_slab_doubles(n, m)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=g.DEV)
    varmap = torch.from_numpy(np.random.default_rng(0).permutation(n).astype(np.int64) + 1).to(g.DEV)
    nq = n * (n + 1) // 2
    for inst in (0, 5, total - 1):
        A = wl.A[inst * r * n:(inst + 1) * r * n]
        b = wl.b[inst * r:(inst + 1) * r]
        Cm = wl.Cm[inst * m * n:(inst + 1) * m * n]
        d = wl.d[inst * m:(inst + 1) * m]
        # the data of instance k does not depend on the sharding: global element index drives the stream
        assert g.same_bits(A.cpu().numpy(), O.fill_uniform((inst + 1) * r * n, 101)[inst * r * n:])
        oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
        ov, ovc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
        g.call("pmt_batch_expand_f64", g.ptr(wl.local[inst]), n, m, g.ptr(xvar), g.ptr(varmap), g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ov),
               g.ptr(ovc), g.stream())
        sq, sl, sc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
        ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(r, n) // 8)
        g.call("pmt_quad_gram_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, 1, g.ptr(varmap), g.ptr(sq), g.ptr(sl), g.ptr(sc), g.ptr(ws), g.stream())
        sv, svc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
        g.call("pmt_affine_pack_vector_f64", g.ptr(Cm), m, m, n, g.ptr(xvar), g.ptr(d), -1, g.ptr(varmap), 0, g.ptr(sv), g.ptr(svc), g.stream())
        # (the single-instance node of a one-tile shape is the fused tall form, gram_tall.hip: its own fixed summation order — the batch
        # kernel's sums agree with it to rounding, indices exactly)
        bq, sq_h = g.terms_to_host(oq, nq, g.QT), g.terms_to_host(sq, nq, g.QT)
        assert np.array_equal(bq["row"], sq_h["row"]) and np.array_equal(bq["col"], sq_h["col"])
        np.testing.assert_allclose(bq["coeff"], sq_h["coeff"], rtol=1e-12, atol=0)
        bl, sl_h = g.terms_to_host(ol, n, g.LT), g.terms_to_host(sl, n, g.LT)
        assert np.array_equal(bl["var"], sl_h["var"])
        np.testing.assert_allclose(bl["coeff"], sl_h["coeff"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(g.f64_to_host(oc, 1), g.f64_to_host(sc, 1), rtol=1e-14, atol=0)
        g.assert_terms_equal(g.terms_to_host(ov, m * n, g.VAT), g.terms_to_host(sv, m * n, g.VAT))
        assert g.same_bits(g.f64_to_host(ovc, m), g.f64_to_host(svc, m))
        # and against numpy on the host copy (tolerance of north_star)
        Ah = A.cpu().numpy().reshape(n, r).T
        bh = b.cpu().numpy()
        slab = wl.local[inst].cpu().numpy()
        np.testing.assert_allclose(slab[:nq], (2 * Ah.T @ Ah)[np.triu_indices(n)], rtol=1e-12)
        np.testing.assert_allclose(slab[off["q"]:off["q"] + n], -2 * Ah.T @ bh, rtol=1e-12)
        assert slab[off["const"]] == pytest.approx(bh @ bh, rel=1e-14)
        assert np.array_equal(slab[off["C"]:off["C"] + m * n].reshape(m, n), Cm.cpu().numpy().reshape(n, m).T)
        assert np.array_equal(slab[off["dconst"]:], 0.0 - d.cpu().numpy())


def test_batch_sharding_is_data_independent():
    from parametron_jl_amd import batch
    total, n, r, m = 8, 128, 128, 16
    whole = batch.BatchLSQ(torch, total, n, r, m)
    whole.compute()
    parts = []
    for rank in range(4):
        p = batch.BatchLSQ(torch, total, n, r, m, rank=rank, world=4)
        p.compute()
        parts.append(p.local.clone())
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), whole.local)


@pytest.mark.parametrize("n,r,m", [(128, 128, 16), (128, 70, 3), (100, 33, 5), (16, 1, 1), (128, 200, 2), (200, 64, 4), (128, 64, 40), (64, 50, 40), (127, 31, 17)])
def test_batch_small_and_general_paths_agree_with_numpy(n, r, m):
    """Ragged shapes: fewer than 128 columns (zero padded in LDS), row counts that are not a multiple of the 32-row chunk (1 to 7
    chunks per instance), odd leading dimensions (unaligned loads), constraint blocks too large for the register prefetch (m*n > 2048)
    or for the LDS staging buffer (written straight to HBM), and n > 128 (general tiled path)."""
    import gpu_util as g
    from parametron_jl_amd import batch
    total = 5
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.compute()
    torch.cuda.synchronize()
    off, L = batch.slab_layout(n, m)
    nq = n * (n + 1) // 2
    for inst in range(total):
        Ah = wl.A[inst * r * n:(inst + 1) * r * n].cpu().numpy().reshape(n, r).T
        bh = wl.b[inst * r:(inst + 1) * r].cpu().numpy()
        slab = wl.local[inst].cpu().numpy()
        np.testing.assert_allclose(slab[:nq], (2 * Ah.T @ Ah)[np.triu_indices(n)], rtol=1e-12)
        np.testing.assert_allclose(slab[off["q"]:off["q"] + n], -2 * Ah.T @ bh, rtol=1e-12)
        seq = 0.0
        for v in 0.0 - bh:
            seq = seq + v * v
        assert slab[off["const"]] == seq                                   # left-to-right sum, bit for bit
        Ch = wl.Cm[inst * m * n:(inst + 1) * m * n].cpu().numpy().reshape(n, m).T
        assert np.array_equal(slab[off["C"]:off["C"] + m * n].reshape(m, n), Ch)


def test_config4_full_size_checksums():
    """BASELINE config 4 at full size (8192 instances of n = r = 128, m = 16): per-instance coefficient checksums in closed form
    (sum of Q = ||A 1||^2 + ||A||_F^2, sum of q = -2 (A 1).b, const = b.b), the constraint block bit for bit — vectorised in torch."""
    from parametron_jl_amd import batch
    total, n, r, m = 8192, 128, 128, 16
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.compute()
    torch.cuda.synchronize()
    off, L = batch.slab_layout(n, m)
    nq = n * (n + 1) // 2
    A = wl.A.view(total, n, r).transpose(1, 2)                             # (B, r, n): column-major per instance
    b = wl.b.view(total, r)
    row1 = A.sum(dim=2)
    want_q = (row1 * row1).sum(dim=1) + (A * A).sum(dim=(1, 2))
    got = wl.local
    torch.testing.assert_close(got[:, :nq].sum(dim=1), want_q, rtol=1e-11, atol=0)
    torch.testing.assert_close(got[:, off["q"]:off["q"] + n].sum(dim=1), -2 * (row1 * b).sum(dim=1), rtol=1e-11, atol=0)
    torch.testing.assert_close(got[:, off["const"]], (b * b).sum(dim=1), rtol=1e-13, atol=0)
    Cm = wl.Cm.view(total, n, m).transpose(1, 2)                           # (B, m, n)
    assert torch.equal(got[:, off["C"]:off["C"] + m * n].view(total, m, n), Cm)
    assert torch.equal(got[:, off["dconst"]:], 0.0 - wl.d.view(total, m))


def test_batch_many_instances_per_workgroup():
    """More than 64 instances per persistent workgroup (one per CU): the c'c chains are computed 64 instances at a time by the lanes of
    one wave; every instance's slab must still be its own (const bit for bit, Q/q within 1e-12, C exactly)."""
    from parametron_jl_amd import batch
    total, n, r, m = 256 * 64 + 300, 8, 5, 1
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.compute()
    torch.cuda.synchronize()
    off, L = batch.slab_layout(n, m)
    nq = n * (n + 1) // 2
    A = wl.A.view(total, n, r).transpose(1, 2).cpu().numpy()              # (B, r, n)
    b = wl.b.view(total, r).cpu().numpy()
    got = wl.local.cpu().numpy()
    iu = np.triu_indices(n)
    G = 2.0 * np.einsum("bri,brj->bij", A, A)
    np.testing.assert_allclose(got[:, :nq], G[:, iu[0], iu[1]], rtol=1e-12)
    np.testing.assert_allclose(got[:, off["q"]:off["q"] + n], -2.0 * np.einsum("bri,br->bi", A, b), rtol=1e-12)
    seq = np.zeros(total)
    for i in range(r):                                                     # left to right, vectorised over instances
        c = 0.0 - b[:, i]
        seq = seq + c * c
    assert np.array_equal(got[:, off["const"]], seq)
    assert np.array_equal(got[:, off["C"]:off["C"] + m * n].reshape(total, m, n), wl.Cm.view(total, n, m).transpose(1, 2).cpu().numpy())
    assert np.array_equal(got[:, off["dconst"]:], 0.0 - wl.d.view(total, m).cpu().numpy())


def test_batch_step_through_the_library_communicator_single_rank():
    """pmt_batch_step_f64 (computation in chunks + exchange on the communicator's stream) with one rank: the gathered buffer equals the
    plain computation, for an uneven last chunk and for a single chunk; steps can be issued back to back."""
    from parametron_jl_amd import batch
    total, n, r, m = 37, 128, 64, 16
    ref = batch.BatchLSQ(torch, total, n, r, m)
    ref.compute()
    torch.cuda.synchronize()
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.gathered = torch.full_like(wl.local, float("nan"))                  # a separate gathered buffer, as with world > 1
    comm = batch.Communicator(torch, None, 0, 1, torch.cuda.current_device())
    try:
        for chunk in (8, 0, 37, 100):
            wl.local.fill_(float("nan")); wl.gathered.fill_(float("nan"))
            torch.cuda.synchronize()
            wl.step_pipelined(comm, chunk)
            wl.step_pipelined(comm, chunk)
            torch.cuda.synchronize()
            assert torch.equal(wl.gathered, ref.local) and torch.equal(wl.local, ref.local)
    finally:
        comm.close()


def test_one_rank_rccl_communicator_exchanges_through_rccl():
    """The library's RCCL binding (hand-declared prototypes resolved from librccl, ncclGetUniqueId, ncclCommInitRank, grouped ncclSend /
    ncclRecv with ncclDouble, the communication stream and both events) on ONE GPU: a one-rank communicator built from a unique id sends
    every chunk to itself.  pmt_batch_step_f64 with 4 uneven chunks: `gathered` is filled by RCCL alone, equals `local` bit for bit and
    the single-launch slabs, and the instances' coefficient blocks equal the oracle's."""
    import gpu_util as g
    from parametron_jl_amd import batch, _lib
    total, n, r, m = 103, 128, 96, 16                                  # chunks of 30, 30, 30, 13 instances
    wl = batch.BatchLSQ(torch, total, n, r, m)
    wl.compute()
    torch.cuda.synchronize()
    want = wl.local.clone()
    comm = batch.Communicator(torch, None, 0, 1, torch.cuda.current_device(), rccl_single=True)
    assert comm.rccl_calls() == 0
    wl.gathered = torch.full((total, wl.L), float("nan"), dtype=torch.float64, device=g.DEV)       # a buffer of its own: only RCCL writes it
    chunk = 30
    assert batch.chunk_schedule(total, chunk) == [(0, 30), (30, 60), (60, 90), (90, 103)]
    for epoch in range(2):
        wl.local.fill_(float("nan"))
        wl.step_pipelined(comm, chunk)
        torch.cuda.synchronize()
        assert torch.equal(wl.gathered, wl.local) and torch.equal(wl.local, want)
        assert comm.rccl_calls() == 8 * (epoch + 1)                     # one send + one receive per chunk
    # exchange of slabs that already exist (pmt_batch_allgather_f64), default stream, one chunk and many
    for chunk in (0, 7):
        wl.gathered.fill_(float("nan"))
        before = comm.rccl_calls()
        _lib.call("pmt_batch_allgather_f64", comm.handle, g.ptr(wl.local), g.ptr(wl.gathered), total, wl.L, chunk, g.stream())
        torch.cuda.synchronize()
        assert torch.equal(wl.gathered, want)
        assert comm.rccl_calls() - before == 2 * len(batch.chunk_schedule(total, chunk))
    # in place (gathered is local, the one-GPU bench layout): RCCL's self copy leaves the block as it is
    _lib.call("pmt_batch_allgather_f64", comm.handle, g.ptr(wl.local), g.ptr(wl.local), total, wl.L, 30, g.stream())
    torch.cuda.synchronize()
    assert torch.equal(wl.local, want)
    comm.close()
    # and the gathered slabs against the oracle's builders for a few instances (constraint block and constants bit for bit, Q / q to 1e-12)
    off, L = batch.slab_layout(n, m)
    nq = n * (n + 1) // 2
    got = wl.gathered.cpu().numpy()
    for inst in (0, 29, 30, 102):
        A = O.fill_uniform((inst + 1) * r * n, 101)[inst * r * n:].reshape(n, r).T
        b = O.fill_uniform((inst + 1) * r, 102)[inst * r:]
        Cm = O.fill_uniform((inst + 1) * m * n, 103)[inst * m * n:].reshape(n, m).T
        d = O.fill_uniform((inst + 1) * m, 104, 2.0)[inst * m:]
        w = O.LsqWorkspace(n, r, m)
        xvar = np.arange(1, n + 1, dtype=np.int64)
        w.eval_objective(np.ascontiguousarray(A.T).reshape(-1), b, xvar)
        w.eval_constraint(np.ascontiguousarray(Cm.T).reshape(-1), d, xvar)
        w.objective.canonicalize()
        at, qt, const = w.objective.moi()
        ct, cc = w.constraint.moi()
        np.testing.assert_allclose(got[inst, :nq], qt["coeff"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(got[inst, off["q"]:off["q"] + n], at["coeff"], rtol=1e-12, atol=0)
        assert got[inst, off["const"]] == const
        assert np.array_equal(got[inst, off["C"]:off["C"] + m * n], ct["coeff"])
        assert np.array_equal(got[inst, off["dconst"]:], cc)
