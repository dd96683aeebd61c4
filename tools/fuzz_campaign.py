"""A longer randomised campaign than the suite runs, on the GPU box: the suite's own parity checks called with random shapes.
    python tools/fuzz_campaign.py [seed] [cases]      -> one line per failure, a summary at the end (exit code 1 on any failure)
Covers what changed last: staged deliveries (band order, ragged sizes, stage hints) in both transports, the Gram node, the dense affine
nodes, the sparse block kernels against the oracle-shaped reference (scipy CSR) and random stacked models through the host hand-off."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import parametron_jl_amd as P  # noqa: E402
import test_gpu_host_csc as H  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import test_gpu_sparse as S  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
fails, ran = [], 0


def attempt(name, fn, *args):
    global ran
    ran += 1
    try:
        fn(*args)
    except Exception as e:                       # noqa: BLE001
        fails.append((name, args, "%s: %s" % (type(e).__name__, str(e)[:300])))
        print("FAIL", name, args, type(e).__name__, str(e)[:300], flush=True)
        traceback.print_exc(limit=3)


def random_stacked_model(k):
    """objective + a random stack of dense / bounds / sparse blocks through host_csc: host arrays == device hand-off == dense expectation"""
    import scipy.sparse as sp
    r = np.random.default_rng(1000 * seed + k)
    n, rr = int(r.integers(3, 260)), int(r.integers(1, 300))
    off = int(r.integers(0, 4))
    model = P.Model(P.MockOptimizer(variable_offset=off), quadratic_mode="canonical", handoff=str(r.choice(["host_csc", "device"])))
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((rr, n), 1, model)
    b = P.DeviceUniformParameter((rr,), 2, model)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    blocks = []
    for bi in range(int(r.integers(1, 4))):
        kind = str(r.choice(["dense", "bounds", "sparse"]))
        op = str(r.choice(["<=", ">=", "=="]))
        if kind == "dense":
            m = int(r.integers(1, 90))
            G = P.Parameter(model, val=r.random((m, n)))
            h = P.Parameter(model, val=r.random(m))
            P.constraint(model, G * x, op, h)
            blocks.append((kind, op, G, h))
        elif kind == "bounds":
            v = P.Parameter(model, val=r.random(n))
            P.constraint(model, x, op, v)
            blocks.append((kind, op, None, v))
        else:
            m = int(r.integers(1, 60))
            Ss = sp.random(m, n, density=float(r.uniform(0.05, 0.6)), format="csc", random_state=r, data_rvs=lambda kk: r.random(kk) + 0.1)
            Ss.sort_indices()
            if Ss.nnz == 0:
                continue
            Sp = P.Parameter(model, val=Ss.copy())
            e = P.Parameter(model, val=r.random(m))
            P.constraint(model, Sp * x, op, e)
            blocks.append((kind, op, Sp, e))
    if not blocks:
        model.close()
        return
    for it in range(2):
        for (kind, op, M, v) in blocks:
            if kind == "dense":
                M.val[...] = r.random(M.val.shape)
            elif kind == "sparse":
                M.val.data[...] = r.random(M.val.nnz) + 0.1
            v.val[...] = r.random(v.val.shape)
        P.solve(model)
        qp = model.device_qp
        got = qp.host.as_dict() if qp.host is not None else qp.fetch()
        if qp.host is not None:
            H.assert_host_equals_device(model)
        rows = qp.nrows
        Ad = sp.csc_matrix(got["A"], shape=(rows, n + off)).toarray()
        # expectation: rows stacked in the reference's update order (Constraints.__iter__)
        want = np.zeros((rows, n + off))
        r0 = 0
        for c in model.constraints:
            blk = next(bk for bk in blocks if (bk[2] is not None and any(bk[2] is p_ for p_ in _params(c))) or (bk[2] is None and any(bk[3] is p_ for p_ in _params(c))))
            kind, op, M, v = blk
            m = c.nrows
            dense = np.eye(n) if kind == "bounds" else (M.val if kind == "dense" else M.val.toarray())
            want[r0:r0 + m, off:] = dense
            r0 += m
        assert np.array_equal(Ad, want), "A differs"
    model.close()


def _params(c):
    from parametron_jl_amd.lazyexpression import schedule
    from parametron_jl_amd.parameter import Parameter
    return [x for x in schedule([c.expr]) if isinstance(x, Parameter)]


for k in range(cases):
    rows, cols = int(rng.integers(1, 3000)), int(rng.integers(1, 1900))
    ng = int(rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 16]))
    mode = int(rng.choice([0, 0, 2]))
    P.set_host_delivery(mode)
    attempt("deliver_csc[mode %d]" % mode, H.test_deliver_entry_point_matches_plain_csc, rows, cols, ng)
    if k % 2 == 0:
        attempt("deliver_quad[mode %d]" % mode, H.test_deliver_quadratic_terms_matches_plain_node, int(rng.integers(1, 2500)), int(rng.integers(1, 1300)), ng)
    P.set_host_delivery(0)
    if k % 3 == 0:
        attempt("gram", F.test_gram_node_random_shapes, int(rng.integers(1, 1500)), int(rng.integers(1, 700)))
        attempt("affine", F.test_affine_nodes_random_shapes, int(rng.integers(1, 900)), int(rng.integers(1, 900)))
    attempt("stacked_model", random_stacked_model, k)
print("fuzz campaign seed %d: %d checks, %d failures" % (seed, ran, len(fails)), flush=True)
sys.exit(1 if fails else 0)
