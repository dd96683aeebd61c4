"""-m gpu: the one-launch form of the canonical objective node for wide shapes of up to 4096 columns (csrc/gram_mid.hip: 64 x 64 tiles, the
contraction index split among the waves and over row chunks, last-arriver fold in chunk order, c'c as one more workgroup) — every plan
regime (one chunk per tile, split tiles in one round, several rounds, the XCD-aware order beyond 4 MB, fewer than 32 rows), every load path,
both outputs; repeated and concurrent launches give the same BITS (the fold's order does not depend on who arrives last; the per-tile
counts re-arm themselves).  Reference semantics: canonicalize(_vecdot!(residual, residual)) -> update!(::MOI.ScalarQuadraticFunction),
/root/reference/src/functions.jl:702-709,548-576,381-386 and src/moi_interop.jl:45-62."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _order(rows, n):
    import gpu_util as g
    o = C.c_int()
    g.call("pmt_quad_gram_constant_order", rows, n, C.byref(o), None, None)
    return o.value


# (rows, columns, pad of the leading dimension, 8-byte shift of the base)
SHAPES = [
    (1, 129, 0, 0), (7, 200, 1, 0), (15, 300, 0, 1), (31, 640, 0, 0),                      # fewer than four 8-row groups: idle waves, one chunk
    (33, 129, 0, 0), (64, 130, 0, 0), (130, 130, 2, 0), (40, 520, 0, 0), (100, 1000, 1, 1),  # one chunk per tile (no fold)
    (300, 300, 0, 0), (517, 391, 0, 1), (1024, 512, 0, 0), (1027, 515, 3, 0),                # split tiles, one round; ragged rows / columns
    (4096, 512, 0, 0), (4099, 448, 0, 0), (5000, 129, 1, 0),                               # the masked last 8-row group beside unmasked ones
    (2048, 1024, 0, 0), (4096, 1024, 0, 0), (2048, 1536, 0, 0), (1000, 2048, 2, 0),         # several rounds of workgroups; > 4 MB: XCD-aware order
    (128, 2048, 0, 0), (20000, 200, 0, 0), (8192, 512, 0, 1), (65536, 256, 0, 0), (30001, 130, 1, 0),          # many chunks per tile: the two-level fold
    # 2049 .. 4096 columns (round 6c, config 2's regime): unsplit tiles in several rounds walked by persistent workgroups in super-tile order,
    # the diagonal tiles and the partial round split; few rows, ragged rows / columns, odd pitch, shifted base (the masked load path)
    (5, 2100, 0, 0), (100, 2500, 1, 0), (31, 4096, 0, 1), (1000, 3000, 0, 0), (2048, 2304, 0, 0), (1029, 2049, 3, 1), (1500, 4095, 0, 0),
    (3000, 2100, 1, 0), (2000, 2560, 0, 0),
]


@pytest.mark.parametrize("rows,n,pad,shift", SHAPES)
def test_one_launch_wide_node_against_numpy_on_every_load_path(rows, n, pad, shift):
    from test_gpu_fuzz import _check_gram_node
    assert _order(rows, n) == 5, "the one-launch form did not take this shape"
    _check_gram_node(rows, n, rows + pad, shift, np.random.default_rng(rows * 13 + n))


def _node(g, dA, lda, rows, n, xvar, db, sign, ws, stream, out=None):
    nq = n * (n + 1) // 2
    oq, ol, oc = out or (g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1))
    g.call("pmt_quad_gram_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), g.ptr(db), sign, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), stream)
    return oq, ol, oc


@pytest.mark.parametrize("rows,n", [(300, 300), (4096, 512), (4096, 1024), (20000, 200), (1024, 2560)])      # (20000 x 200: groups of 8 chunks folded first, then the groups; the last: persistent workgroups, split tails)
def test_repeated_launches_give_the_same_bits(rows, n):
    """the sums of a split tile are added in chunk order by whichever workgroup arrives last, and that workgroup re-arms the tile's count:
    thirty launches in a row, every one bit-identical to the first"""
    import gpu_util as g
    assert _order(rows, n) == 5
    rng = np.random.default_rng(rows + n)
    dA, db = g.colmajor(rng.random((rows, n)) - 0.5), g.to_dev(rng.random(rows))
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8)
    nq = n * (n + 1) // 2
    first = None
    for _ in range(30):
        oq, ol, oc = _node(g, dA, rows, rows, n, xvar, db, -1, ws, g.stream())
        got = (g.terms_to_host(oq, nq, g.QT).tobytes(), g.terms_to_host(ol, n, g.LT).tobytes(), g.f64_to_host(oc, 1).tobytes())
        if first is None:
            first = got
        assert got == first


@pytest.mark.parametrize("rows,n", [(2048, 512), (1024, 2304)])      # (the second: two sets of 256 persistent workgroups share the CUs; tickets per stream)
def test_two_streams_run_the_node_at_once_without_sharing_counts(rows, n):
    """the per-tile arrival counts (and the persistent form's tickets) belong to the CALLING STREAM (gram.hip: SideStream): two streams
    launching the node back to back, each on its own matrix and workspace, get what each gets alone"""
    import gpu_util as g
    assert _order(rows, n) == 5
    rng = np.random.default_rng(5)
    nq = n * (n + 1) // 2
    data, alone = [], []
    for k in range(2):
        dA, db = g.colmajor(rng.random((rows, n)) - 0.5), g.to_dev(rng.random(rows))
        xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
        ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8)
        data.append((dA, db, xvar, ws))
        oq, ol, oc = _node(g, dA, rows, rows, n, xvar, db, -1, ws, g.stream())
        alone.append((g.terms_to_host(oq, nq, g.QT).tobytes(), g.terms_to_host(ol, n, g.LT).tobytes(), g.f64_to_host(oc, 1).tobytes()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    # (the poisoned output buffers are filled on torch's current stream: all of them exist, and the fills are done, before the two streams start)
    outs = [[(g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)) for _ in range(12)] for k in range(2)]
    torch.cuda.synchronize()
    for i in range(12):
        for k in range(2):
            dA, db, xvar, ws = data[k]
            _node(g, dA, rows, rows, n, xvar, db, -1, ws, C.c_void_p(streams[k].cuda_stream), outs[k][i])
    torch.cuda.synchronize()
    for k in range(2):
        for oq, ol, oc in outs[k]:
            got = (g.terms_to_host(oq, nq, g.QT).tobytes(), g.terms_to_host(ol, n, g.LT).tobytes(), g.f64_to_host(oc, 1).tobytes())
            assert got == alone[k]


def test_workspace_covers_every_workgroups_partial():
    """pmt_quad_gram_workspace_bytes is what the node may write: a guard band behind it stays untouched"""
    import gpu_util as g
    rng = np.random.default_rng(9)
    for rows, n in ((4096, 1024), (1024, 512), (128, 2048), (1024, 2304)):
        nbytes = g.lib().pmt_quad_gram_workspace_bytes(rows, n)
        guard = 4096
        buf = torch.full((nbytes // 8 + guard,), 7.25, dtype=torch.float64, device="cuda")
        dA, db = g.colmajor(rng.random((rows, n))), g.to_dev(rng.random(rows))
        xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
        _node(g, dA, rows, rows, n, xvar, db, 1, buf, g.stream())
        torch.cuda.synchronize()
        assert bool(torch.all(buf[nbytes // 8:] == 7.25))
