"""End-to-end solve!(model) timing through the host API at BASELINE config 2 (device-resident Parameters, mock optimizer):
update! = tape replay on the device + D2H of every MOI buffer into page-locked host memory + MOI.set hand-off."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parametron_jl_amd as P  # noqa: E402


def main():
    n, r, m = 4096, 4096, 512
    device = "--device-handoff" in sys.argv
    host_csc = "--host-csc" in sys.argv                    # [--serial]: fetch behind the re-evaluation instead of beside it
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical", use_graph="--graph" in sys.argv,
                    handoff="host_csc" if host_csc else ("device" if device else "moi"), overlap_fetch="--serial" not in sys.argv)
    x = [P.Variable(model) for _ in range(n)]
    if "--host-params" in sys.argv:
        # host-updated Parameters in the reference's `val=` form (buffers overwritten externally between solves, src/parameter.jl:88):
        # every solve! uploads A, b, C, d (151 MB).  --pinned: page-locked column-major buffers from model.parameter_array
        import numpy as np
        alloc = model.parameter_array if "--pinned" in sys.argv else (lambda *s: np.zeros(s, order="C" if "--row-major" in sys.argv else "F"))
        rng = np.random.default_rng(0)
        bufs = [alloc(r, n), alloc(r), alloc(m, n), alloc(m)]
        for a in bufs:
            a[...] = rng.random(a.shape)
        A, b, Cm, d = (P.Parameter(model, val=a) for a in bufs)
    else:
        A = P.DeviceUniformParameter((r, n), 1, model)
        b = P.DeviceUniformParameter((r,), 2, model)
        Cm = P.DeviceUniformParameter((m, n), 3, model)
        d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    residual = A * x - b
    P.objective(model, P.Minimize, P.dot(residual, residual))
    P.constraint(model, Cm * x == d)
    t0 = time.perf_counter(); P.solve(model); t_first = time.perf_counter() - t0
    for _ in range(3):
        P.solve(model)
    k = 10
    if "--staged" in sys.argv:
        # overlapped uploads: the values of solve k+1 travel on the copy stream while solve k runs (Model.stage_parameters)
        def step():
            model.stage_parameters()
            model.update(synchronize=False)
            model.optimizer.optimize()
    else:
        def step():
            P.solve(model)
    for _ in range(3):
        step()
    model.device().synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    model.device().synchronize()
    if "--staged" in sys.argv:
        model.wait_staged()
    dt = (time.perf_counter() - t0) / k
    if host_csc:
        import numpy as np
        h = model.device_qp.host
        dev = model.device_qp.fetch()
        hd = h.as_dict()
        same = {k: bool(np.array_equal(hd[k][0] if k in ("P", "A") else hd[k], dev[k][0] if k in ("P", "A") else dev[k])) for k in ("P", "A", "q", "l", "u")}
        print("host arrays equal the device hand-off after the timed loop:", same, "r", hd["r"] == dev["r"])
        print("host_csc hand-off (%s): steady solve! %.3f ms = %.1f /s; %.1f MB to the host per solve (PCIe floor at 54 GB/s: %.2f ms); P delivered by the contraction: %s"
              % ("serial" if "--serial" in sys.argv else "overlapped", dt * 1e3, 1 / dt, h.nbytes() / 1e6, h.nbytes() / 54e9 * 1e3, h.P_delivered_by_contraction))
        return
    if device:
        qp = model.device_qp
        t0 = time.perf_counter(); got = qp.fetch(); t_fetch = time.perf_counter() - t0
        nb = 8 * (qp.P.nnz + qp.A.nnz + qp.nvars + 2 * qp.nrows)
        print("device hand-off: first solve! (initialize + CSC structure) %.1f ms; steady solve! (tape + CSC values; %s) %.3f ms = %.1f /s;"
              " P nnz %d, A nnz %d; fetching the CSC values (pageable) %.1f MB in %.1f ms" % (t_first * 1e3, "151 MB of Parameter values uploaded per solve" if "--host-params" in sys.argv else "nothing crosses PCIe", dt * 1e3, 1 / dt, qp.P.nnz, qp.A.nnz, nb / 1e6, t_fetch * 1e3))
        return
    f = model.objective.f
    nbytes = f.quadratic_terms.nbytes + f.affine_terms.nbytes + list(model.constraints)[0].f.terms.nbytes
    print("first solve! (initialize + record) %.1f ms; steady solve! %.3f ms = %.1f /s; MOI bytes fetched %.1f MB (%.1f GB/s incl. compute)"
          % (t_first * 1e3, dt * 1e3, 1 / dt, nbytes / 1e6, nbytes / dt / 1e9))


if __name__ == "__main__":
    main()
