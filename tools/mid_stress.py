#!/usr/bin/env python
"""Stress of the one-launch wide Gram node (gram_mid.hip): two streams launching it back to back, every result compared bit for bit with what
the stream's node gives alone.  usage: [PMT_LIB_PATH=...] python tools/mid_stress.py [rows cols [launches]]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as g  # noqa: E402

rows, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 512)
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(5)
nq = n * (n + 1) // 2


def node(dA, db, xvar, ws, stream, out=None):
    oq, ol, oc = out or (g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1))
    g.call("pmt_quad_gram_f64", g.ptr(dA), rows, rows, n, g.ptr(xvar), g.ptr(db), -1, 1, None, g.ptr(oq), g.ptr(ol), g.ptr(oc), g.ptr(ws), stream)
    return oq, ol, oc


data, alone = [], []
for k in range(2):
    dA, db = g.colmajor(rng.random((rows, n)) - 0.5), g.to_dev(rng.random(rows))
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    ws = g.empty_f64(g.lib().pmt_quad_gram_workspace_bytes(rows, n) // 8)
    data.append((dA, db, xvar, ws))
    oq, ol, oc = node(dA, db, xvar, ws, g.stream())
    alone.append((g.terms_to_host(oq, nq, g.QT), g.terms_to_host(ol, n, g.LT), g.f64_to_host(oc, 1)))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for rep in range(launches // 10):
    # (the poisoned output buffers are filled on torch's current stream: they exist, and the fills are done, before the other streams start)
    outs = [[(g.empty_terms(nq, g.QT), g.empty_terms(n, g.LT), g.empty_f64(1)) for _ in range(10)] for k in range(2)]
    torch.cuda.synchronize()
    for i in range(10):
        for k in range(2):
            node(*data[k], C.c_void_p(streams[k].cuda_stream), outs[k][i])
    torch.cuda.synchronize()
    for k in range(2):
        for i, (oq, ol, oc) in enumerate(outs[k]):
            q, l, c = g.terms_to_host(oq, nq, g.QT), g.terms_to_host(ol, n, g.LT), g.f64_to_host(oc, 1)
            wq, wl, wc = alone[k]
            dq = np.nonzero(q["coeff"].view(np.int64) != wq["coeff"].view(np.int64))[0]
            dl = np.nonzero(l["coeff"].view(np.int64) != wl["coeff"].view(np.int64))[0]
            if len(dq) or len(dl) or c[0] != wc[0] or not np.array_equal(q["row"], wq["row"]) or not np.array_equal(q["col"], wq["col"]):
                bad += 1
                if bad <= 5:
                    iu = np.triu_indices(n)
                    tiles = sorted(set(zip((iu[0][dq] // 64).tolist(), (iu[1][dq] // 64).tolist())))
                    rel = np.abs(q["coeff"][dq] - wq["coeff"][dq]) / np.maximum(np.abs(wq["coeff"][dq]), 1e-300)
                    print("rep %d stream %d launch %d: %d quadratic coefficients differ (tiles %s, max rel %.3g), %d linear, const %s" %
                          (rep, k, i, len(dq), tiles[:8], rel.max() if len(rel) else 0.0, len(dl), c[0] == wc[0]))
print("launches per stream: %d, differing results: %d" % (launches, bad))
