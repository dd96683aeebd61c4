"""Random differential sweep of the canonical objective node (pmt_quad_gram_f64 / _csc_f64) against float64 numpy over the node's dispatch
boundaries: tiny / 16-32-64-column panels / one tile / diagonal tiles + strict stream-K / stream-K node, whole and ragged panels, aligned and
8-byte-shifted A, odd pitches, sign in {-1, 0, +1}, MOI / native form.  python tools/gram_random_sweep.py [count] [seed]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
import gpu_util as g

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
edges_n = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 257, 511, 640, 1024, 2047, 2048, 2049, 2200]
edges_r = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 257, 1023, 1024, 2048, 2049, 4097, 8191, 8193, 20000]
bad = 0
for it in range(count):
    n = int(rng.choice(edges_n)) if rng.random() < 0.5 else int(rng.integers(1, 700))
    rmax = max(1, min(20000, (1 << 24) // n))
    r = min(rmax, int(rng.choice(edges_r)) if rng.random() < 0.5 else int(rng.integers(1, rmax + 1)))
    if n <= 64 and rng.random() < 0.4:                 # the stream form of the narrow panels starts at 32768 rows (round 6)
        r = int(rng.choice([32767, 32768, 32769, 32784, 40001, 65536, 65537, 70000])) if rng.random() < 0.6 else int(rng.integers(32768, 90000))
    lda = r + int(rng.integers(0, 3)) * int(rng.integers(0, 5))
    shift = int(rng.integers(0, 2))
    sign = int(rng.choice([-1, 0, 1]))
    moi = int(rng.integers(0, 2))
    A = rng.random((r, n)) - 0.35
    b = rng.random(r) - 0.4
    buf = np.zeros(shift + lda * n)
    buf[shift:].reshape(n, lda)[:, :r] = A.T
    dbuf = g.to_dev(buf); dA = dbuf[shift:]
    xv = np.cumsum(rng.integers(1, 3, size=n)).astype(np.int64)
    vm_h = rng.permutation(int(xv[-1])).astype(np.int64) + 1
    xvar, vm, db = g.to_dev(xv), g.to_dev(vm_h), g.to_dev(b)
    nq = n * (n + 1) // 2
    oq, ol, oc = g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)
    ws = g.empty_f64(max(1, g.lib().pmt_quad_gram_workspace_bytes(r, n) // 8))
    g.call("pmt_quad_gram_f64", g.ptr(dA), lda, r, n, g.ptr(xvar), g.ptr(db) if sign else None, sign, moi, g.ptr(vm) if moi else None,
           g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), g.stream())
    q, l, c0 = g.terms_to_host(oq, nq, g.QT), g.terms_to_host(ol, n, g.LT), g.f64_to_host(oc, 1)[0]
    iu = np.triu_indices(n)
    G = A.T @ A
    want = 2 * G[iu]
    if not moi:
        want = np.where(iu[0] == iu[1], G[iu], want)
    idx = (lambda v: vm_h[v - 1]) if moi else (lambda v: v)
    c = (0.0 + sign * b) if sign else np.zeros(r)
    scale = 2 * (np.abs(A).T @ np.abs(A))[iu]
    ok = (np.array_equal(q["row"], idx(xv[iu[0]])) and np.array_equal(q["col"], idx(xv[iu[1]]))
          and np.all(np.abs(q["coeff"] - want) <= 1e-12 * scale + 1e-300)
          and np.array_equal(l["var"], idx(xv))
          and np.all(np.abs(l["coeff"] - 2 * A.T @ c) <= 1e-12 * (2 * np.abs(A).T @ np.abs(c)) + 1e-300)
          and abs(c0 - c @ c) <= 1e-13 * max(c @ c, 1e-300))
    # the CSC form writes the same coefficients (MOI form)
    if ok and moi:
        px = g.empty_f64(nq)
        g.call("pmt_quad_gram_csc_f64", g.ptr(dA), lda, r, n, g.ptr(xvar), g.ptr(db) if sign else None, sign, g.ptr(vm), 1.0, g.ptr(px), None, g.ptr(ol), g.ptr(oc),
               g.ptr(ws), g.stream())
        w2 = np.empty(nq); w2[iu[1] * (iu[1] + 1) // 2 + iu[0]] = q["coeff"]
        got = g.f64_to_host(px, nq)
        order = C.c_int()
        g.call("pmt_quad_gram_constant_order", r, n, C.byref(order), None, None)
        tiny = order.value == 0 and n <= 128                   # (the interpreter's node: row-order sums; the CSC form of a tiny shape is the stream-K node)
        sc2 = np.empty(nq); sc2[iu[1] * (iu[1] + 1) // 2 + iu[0]] = scale
        ok = np.all(np.abs(got - w2) <= 1e-12 * sc2 + 1e-300) if tiny else g.same_bits(got, w2)      # (signed data: relative to sum |a||b|)
    if not ok:
        bad += 1
        print("MISMATCH r=%d n=%d lda=%d shift=%d sign=%d moi=%d" % (r, n, lda, shift, sign, moi), flush=True)
print("gram_random_sweep: %d shapes, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
