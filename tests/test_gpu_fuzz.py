"""-m gpu: randomised shapes through the Gram node (stream-K split, whole-tile and bounds-checked paths, padded / odd leading
dimensions, 8-byte-misaligned bases) and the dense affine nodes, against numpy — a fixed seed, so failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _shapes(seed, count, rmax, nmax):
    rng = np.random.default_rng(seed)
    special = [(1, 1), (16, 128), (256, 128), (257, 129), (512, 256), (300, 127), (15, 300), (272, 384)]
    rnd = [(int(rng.integers(1, rmax)), int(rng.integers(1, nmax))) for _ in range(count - len(special))]
    return special + rnd


@pytest.mark.parametrize("rows,n", _shapes(11, 28, 700, 420))
def test_gram_node_random_shapes(rows, n):
    rng = np.random.default_rng(rows * 1000 + n)
    # odd / even / padded leading dimensions; an 8-byte misaligned base: the 16-byte load path must not be taken
    _check_gram_node(rows, n, rows + int(rng.integers(0, 5)), int(rng.integers(0, 2)), rng)


# round 6: the stream form of the narrow panels (from 32768 rows; 16 / 32 / 48 / 64-column groups), the tall kernel per block-column count
# (65 .. 128 columns) and the mid-size form, each on EVERY load path: whole aligned panels (unmasked 16-byte loads), ragged rows / columns
# (the masked tail iteration, clamped columns), an odd pitch or an 8-byte-shifted base (8-byte loads throughout)
@pytest.mark.parametrize("rows,n,pad,shift", [
    (32768, 16, 0, 0), (32800, 13, 0, 0), (33001, 20, 1, 0), (32769, 32, 0, 1), (40000, 48, 2, 0), (32784, 41, 0, 0), (32784, 60, 3, 1), (65536, 64, 0, 0),
    (4000, 70, 0, 0), (3001, 90, 1, 1), (2999, 100, 0, 1), (4096, 112, 0, 0), (3000, 128, 2, 0), (40000, 80, 0, 0), (33000, 97, 2, 0),
    (1000, 300, 1, 1), (4096, 512, 0, 0), (2040, 1000, 2, 0)])
def test_gram_node_forms_of_round_6_on_every_load_path(rows, n, pad, shift):
    _check_gram_node(rows, n, rows + pad, shift, np.random.default_rng(rows * 7 + n))


def _check_gram_node(rows, n, lda, shift, rng):
    import gpu_util as g
    A = rng.random((rows, n)) - 0.4
    b = rng.random(rows) - 0.5
    buf = np.zeros(shift + lda * n)
    buf[shift:].reshape(n, lda)[:, :rows] = A.T
    dbuf = g.to_dev(buf)
    dA = dbuf[shift:]
    db, xvar = g.to_dev(b), g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    vm_h = np.arange(1, n + 1, dtype=np.int64) + 3
    vm = g.to_dev(vm_h)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8)
    sign = int(rng.choice([-1, 1]))
    g.call("pmt_quad_gram_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), g.ptr(db), sign, 1, g.ptr(vm), g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    q = g.terms_to_host(oq, nq, g.QT)
    iu = np.triu_indices(n)
    G = 2 * A.T @ A
    scale = 2 * np.abs(A).T @ np.abs(A)                                    # signed data: tolerance relative to sum |a||b|
    assert np.array_equal(q["row"], vm_h[iu[0]]) and np.array_equal(q["col"], vm_h[iu[1]])
    assert np.all(np.abs(q["coeff"] - G[iu]) <= 1e-12 * scale[iu] + 1e-300)
    l = g.terms_to_host(ol, n, g.LT)
    c = 0.0 + sign * b
    assert np.array_equal(l["var"], vm_h)
    assert np.all(np.abs(l["coeff"] - 2 * A.T @ c) <= 1e-12 * (2 * np.abs(A).T @ np.abs(c)) + 1e-300)
    order, seq = g.constant_in_the_library_order(rows, n, b, sign)          # the library's fixed order for this shape, restated bit for bit
    assert g.f64_to_host(oc, 1)[0] == seq
    assert seq == pytest.approx(float(c @ c), rel=1e-14)
    # the CSC epilogue writes the same coefficients
    px = g.empty_f64(nq)
    g.call("pmt_quad_gram_csc_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), g.ptr(db), sign, g.ptr(vm), 1.0, g.ptr(px), None, g.ptr(ol), g.ptr(oc),
           g.ptr(ws), g.stream())
    want = np.empty(nq)
    want[iu[1] * (iu[1] + 1) // 2 + iu[0]] = q["coeff"]
    assert g.same_bits(g.f64_to_host(px, nq), want)


@pytest.mark.parametrize("rows,n", [(1, 1), (3, 16), (70, 7), (700, 9), (2049, 16), (100, 17), (1000, 32), (515, 33), (4097, 64), (31, 65), (640, 128),
                                    (33, 129), (2000, 200), (130, 600), (64, 2048), (50, 2049)])
@pytest.mark.parametrize("moi", [0, 1])
def test_gram_node_without_affine_part_and_in_native_form(rows, n, moi):
    """b = null / sign = 0 (the objective (A x) . (A x): q = 0, constant = 0) and the NATIVE canonical form (moi = 0: diagonal coefficient
    (A'A)_jj, off-diagonal 2 (A'A)_jk, model variable indices) across the node's forms — tiny, the 16 / 32 / 64-column panels, one tile,
    diagonal tiles + strict stream-K up to 2048 columns, the stream-K node beyond — with 8-byte misaligned and odd-pitched A"""
    import gpu_util as g
    rng = np.random.default_rng(rows * 31 + n + moi)
    lda = rows + int(rng.integers(0, 4))
    shift = int(rng.integers(0, 2))
    A = rng.random((rows, n)) - 0.3
    buf = np.zeros(shift + lda * n)
    buf[shift:].reshape(n, lda)[:, :rows] = A.T
    dbuf = g.to_dev(buf)
    dA = dbuf[shift:]
    xv = np.cumsum(rng.integers(1, 4, size=n)).astype(np.int64)             # strictly increasing, with gaps
    vm_h = rng.permutation(int(xv[-1])).astype(np.int64) + 1
    xvar, vm = g.to_dev(xv), g.to_dev(vm_h)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(max(1, g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8))
    g.call("pmt_quad_gram_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), None, 0, moi, g.ptr(vm) if moi else None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws),
           g.stream())
    q, l = g.terms_to_host(oq, nq, g.QT), g.terms_to_host(ol, n, g.LT)
    iu = np.triu_indices(n)
    G = A.T @ A
    want = 2 * G[iu]
    if not moi:
        want = np.where(iu[0] == iu[1], G[iu], want)
    idx = (lambda v: vm_h[v - 1]) if moi else (lambda v: v)
    assert np.array_equal(q["row"], idx(xv[iu[0]])) and np.array_equal(q["col"], idx(xv[iu[1]]))
    scale = 2 * (np.abs(A).T @ np.abs(A))[iu]
    assert np.all(np.abs(q["coeff"] - want) <= 1e-12 * scale + 1e-300)
    assert np.array_equal(l["var"], idx(xv)) and np.all(l["coeff"] == 0.0)
    assert g.f64_to_host(oc, 1)[0] == 0.0


@pytest.mark.parametrize("rows,n", _shapes(12, 20, 300, 300))
def test_affine_nodes_random_shapes(rows, n):
    import gpu_util as g
    rng = np.random.default_rng(rows * 977 + n)
    lda = rows + int(rng.integers(0, 4))
    A = rng.random((rows, n)) - 0.5
    b = rng.random(rows)
    buf = np.zeros(lda * n); buf.reshape(n, lda)[:, :rows] = A.T
    dA, db = g.to_dev(buf), g.to_dev(b)
    xv = rng.permutation(n).astype(np.int64) + 1
    vm_h = rng.permutation(n).astype(np.int64) + 1
    xvar, vm = g.to_dev(xv), g.to_dev(vm_h)
    lt, c1 = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)
    vat, c2 = g.empty_terms(rows * n, g.VAT), g.empty_f64(rows)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), g.ptr(db), 1, g.ptr(lt), g.ptr(c1), g.stream())
    g.call("pmt_affine_pack_vector_f64", g.ptr(dA), lda, rows, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(vm), 5, g.ptr(vat), g.ptr(c2), g.stream())
    t = g.terms_to_host(lt, rows * n, g.LT)
    assert g.same_bits(t["coeff"].reshape(rows, n), A) and np.array_equal(t["var"].reshape(rows, n), np.tile(xv, (rows, 1)))
    assert g.same_bits(g.f64_to_host(c1, rows), 0.0 + b)
    v = g.terms_to_host(vat, rows * n, g.VAT)
    assert g.same_bits(v["coeff"].reshape(rows, n), A) and np.array_equal(v["var"].reshape(rows, n), np.tile(vm_h[xv - 1], (rows, 1)))
    assert np.array_equal(v["out"].reshape(rows, n), np.tile(np.arange(6, rows + 6)[:, None], (1, n)))
    assert g.same_bits(g.f64_to_host(c2, rows), 0.0 - b)


def test_config2_full_size_checksums():
    """BASELINE config 2 at full size (n = r = 4096, m = 512) through size-independent properties: the coefficient checksum of the
    canonical objective is ||A 1||^2 + ||A||_F^2 (sum over j <= k of 2 (A'A)_jk), that of q is -2 (A 1).b, index checksums are closed
    forms, and the constraint block carries C's entries row-major.  Reference sums are computed by torch in fp64 on the device."""
    import ctypes as C
    import gpu_util as g
    n = r = 4096
    m = 512
    s = g.stream()
    A, b, Cm, d = g.empty_f64(r * n), g.empty_f64(r), g.empty_f64(m * n), g.empty_f64(m)
    for buf, cnt, seed, sc in ((A, r * n, 1, 1.0), (b, r, 2, 1.0), (Cm, m * n, 3, 1.0), (d, m, 4, 2.0)):
        g.call("pmt_fill_uniform_f64", g.ptr(buf), cnt, C.c_uint64(seed), sc, s)
    xvar = torch.arange(1, n + 1, dtype=torch.int64, device=g.DEV)
    vm = xvar + 7
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(r, n) // 8)
    g.call("pmt_quad_gram_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, 1, g.ptr(vm), g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), s)
    vt, vc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
    g.call("pmt_affine_pack_vector_f64", g.ptr(Cm), m, m, n, g.ptr(xvar), g.ptr(d), -1, g.ptr(vm), 0, g.ptr(vt), g.ptr(vc), s)
    torch.cuda.synchronize()
    Am = A.view(n, r).t()                                                  # (r, n) view of the column-major buffer
    q3 = oq.view(nq, 3)
    coeff = q3[:, 0].view(torch.float64)
    row1 = Am.sum(dim=1)
    want = float((row1 * row1).sum() + (Am * Am).sum())
    assert float(coeff.sum()) == pytest.approx(want, rel=1e-11)
    assert int(q3[:, 1].sum()) == sum((7 + j + 1) * (n - j) for j in range(n))            # row index vm[j] appears n - j times
    assert int(q3[:, 2].sum()) == sum((7 + k + 1) * (k + 1) for k in range(n))            # col index vm[k] appears k + 1 times
    l2 = ol.view(n, 2)
    assert float(l2[:, 0].view(torch.float64).sum()) == pytest.approx(float(-2 * (row1 * b).sum()), rel=1e-11)
    assert torch.equal(l2[:, 1], vm)
    assert float(oc.item()) == pytest.approx(float((b * b).sum()), rel=1e-13)
    v3 = vt.view(m * n, 3)
    assert torch.equal(v3[:, 1].view(torch.float64).view(m, n), Cm.view(n, m).t())        # bit for bit, row-major
    assert torch.equal(v3[:, 0].view(m, n), torch.arange(1, m + 1, device=g.DEV).view(m, 1).expand(m, n))
    assert torch.equal(v3[:, 2].view(m, n), vm.view(1, n).expand(m, n))
    assert torch.equal(vc, 0.0 - d)
