"""The synthetic BASELINE configurations (BASELINE.json `configs`) built through the host API — shared by bench.py,
tools/ and the tests so that "config 3" means the same model everywhere.

    C1  README Example 1: n = 8, m = 2                                     (README.md:23-57)
    C2  dense least-squares QP, n = r = 4096, m = 512 equality rows
    C3  C2's objective + G*x <= h (512 rows) + bounds x >= l, x <= u, every Parameter of those constraints in the reference's
        `val=` form (src/parameter.jl:88): host buffers the user overwrites between solves, uploaded at every update
    C5  sparse C (5 % non-zeros, fixed pattern), n = 16384, m = 4096: C*x == d, `val=` Parameters
(C4, the batch of 8192 independent n = 128 instances, is parametron_jl_amd.batch.BatchLSQ.)
"""
import numpy as np

from .model import Minimize, MockOptimizer, Model, constraint, objective
from .functions import Variable
from .lazyexpression import dot
from .parameter import DeviceUniformParameter, DeviceUniformSparseParameter, Parameter


def lsq_objective(model, n, r):
    """residual . residual with residual = A*x - b, A and b regenerated on the device at every update (README.md:36-43 rand!)."""
    x = [Variable(model) for _ in range(n)]
    A = DeviceUniformParameter((r, n), 1, model)
    b = DeviceUniformParameter((r,), 2, model)
    residual = A * x - b
    objective(model, Minimize, dot(residual, residual))
    return x, A, b


def config1(mode="literal", use_graph=False, **kw):
    n, m = 8, 2
    model = Model(MockOptimizer(), quadratic_mode=mode, use_graph=use_graph, **kw)
    x, A, b = lsq_objective(model, n, n)
    C = DeviceUniformParameter((m, n), 3, model)
    d = DeviceUniformParameter((m,), 4, model, scale=2.0)
    constraint(model, C * x == d)
    return model


def config2(**kw):
    n, r, m = 4096, 4096, 512
    model = Model(MockOptimizer(), quadratic_mode="canonical", **kw)
    x, A, b = lsq_objective(model, n, r)
    C = DeviceUniformParameter((m, n), 3, model)
    d = DeviceUniformParameter((m,), 4, model, scale=2.0)
    constraint(model, C * x == d)
    return model


def config3(pinned=True, seed=5, **kw):
    """Returns (model, host buffers): the caller overwrites the buffers between solves, as a user of `Parameter(model, val=...)` does."""
    n, r, mi = 4096, 4096, 512
    rng = np.random.default_rng(seed)
    model = Model(MockOptimizer(), quadratic_mode="canonical", **kw)
    x, A, b = lsq_objective(model, n, r)
    alloc = model.parameter_array if pinned else (lambda *s: np.zeros(s, order="F"))
    bufs = {"G": alloc(mi, n), "h": alloc(mi), "l": alloc(n), "u": alloc(n)}
    bufs["G"][...] = rng.random((mi, n)); bufs["h"][...] = rng.random(mi)
    bufs["l"][...] = -rng.random(n); bufs["u"][...] = rng.random(n)
    G, h, l, u = (Parameter(model, val=bufs[k]) for k in ("G", "h", "l", "u"))
    constraint(model, G * x, "<=", h)
    constraint(model, x, ">=", l)
    constraint(model, x, "<=", u)
    return model, bufs


def config5(seed=3, pinned=False, device_resident=False, **kw):
    """device_resident=True: nzval and d are regenerated on the device at every update (the boundary config 2 is benchmarked at);
    otherwise they are `val=` Parameters the host rewrites (27 MB cross PCIe per update)."""
    import scipy.sparse as sp
    m, n = 4096, 16384
    rng = np.random.default_rng(seed)
    k = int(0.05 * m)
    indptr = np.arange(0, (n + 1) * k, k, dtype=np.int64)
    indices = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for _ in range(n)]).astype(np.int64)
    model = Model(MockOptimizer(), **kw)
    if device_resident:
        Cs = sp.csc_matrix((np.ones(indices.size), indices, indptr), shape=(m, n), copy=False)
        x = [Variable(model) for _ in range(n)]
        Cp = DeviceUniformSparseParameter(Cs, 3, model)
        d = DeviceUniformParameter((m,), 4, model, scale=2.0)
        constraint(model, Cp * x == d)
        return model, Cs
    data = model.parameter_array(indices.size) if pinned else np.empty(indices.size)      # page-locked nzval: uploads at PCIe speed
    data[:] = rng.random(indices.size) + 0.1
    Cs = sp.csc_matrix((data, indices, indptr), shape=(m, n), copy=False)
    x = [Variable(model) for _ in range(n)]
    Cp = Parameter(model, val=Cs)
    d = Parameter(model, val=rng.random(m))
    constraint(model, Cp * x == d)
    return model, Cs
