"""Parameters: the leaves of the lazy-expression DAG (src/parameter.jl:36-104).

A Parameter is a placeholder for data with a dirty flag: calling it runs the user's update function at most once
per `setdirty!`.  The value lives on the host (user callbacks are host code, src/parameter.jl:101-102) and is
mirrored into a device buffer owned by the model's plan whenever it was recomputed; `DeviceUniformParameter`
keeps the value resident in HBM and regenerates it with a device kernel.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError


class Parameter:
    """Parameter(f, model)            out-of-place: val = f()          (src/parameter.jl:48, :73)
    Parameter(f, val, model)        in-place: f(val) mutates val     (src/parameter.jl:57)
    Parameter(model, val=val)       identity in-place: a work buffer the user updates manually (src/parameter.jl:88)
    """

    def __init__(self, *args, val=None):
        if len(args) == 1:
            if val is None:
                raise ArgumentError("Parameter(model; val=...) needs val")
            f, inplace, model = (lambda v: v), True, args[0]
        elif len(args) == 2:
            f, model = args
            inplace = False
            if val is not None:
                raise ArgumentError("Parameter(f, model) takes no val")
        elif len(args) == 3:
            f, val, model = args
            inplace = True
        else:
            raise ArgumentError("Parameter(f, model) | Parameter(f, val, model) | Parameter(model, val=...)")
        self.f = f
        self.inplace = inplace
        self.model = model
        self.dirty = True
        self.val = val
        self.version = 0            # bumped every time the update function ran
        self._dev = None            # device mirror (lazyexpression.DeviceValue), created on first device use
        self._dev_version = -1
        model.addparameter(self)

    def __repr__(self):
        return "Parameter{%s, …}(…)" % type(self.val).__name__

    def __call__(self):                                  # src/parameter.jl:93-99
        if self.dirty:
            self.update()
            self.dirty = False
        return self.val

    def update(self):                                    # src/parameter.jl:101-102
        if self.inplace:
            self.f(self.val)
        else:
            self.val = self.f()
        self.version += 1

    def setdirty(self):                                  # src/parameter.jl:104
        self.dirty = True

    # ---- lazy-expression syntax: a Parameter inside an operator builds a LazyExpression
    def _lazy(self, f, *args):
        from .lazyexpression import lazy
        return lazy(f, *args)

    def __mul__(self, o): return self._lazy("*", self, o)
    def __rmul__(self, o): return self._lazy("*", o, self)
    def __matmul__(self, o): return self._lazy("*", self, o)
    def __rmatmul__(self, o): return self._lazy("*", o, self)
    def __add__(self, o): return self._lazy("+", self, o)
    def __radd__(self, o): return self._lazy("+", o, self)
    def __sub__(self, o): return self._lazy("-", self, o)
    def __rsub__(self, o): return self._lazy("-", o, self)
    __array_priority__ = 2000

    @property
    def T(self):
        return self._lazy("adjoint", self)

    def __le__(self, o):
        from .lazyexpression import Relation
        return Relation(self, "<=", o)

    def __ge__(self, o):
        from .lazyexpression import Relation
        return Relation(self, ">=", o)


class DerivedParameter(Parameter):
    """Plain data computed from other Parameters by a Parameter-only expression (`p ⋅ p`, `p.x`, `p.x + 1`; test/model.jl:161,
    test/lazyexpression.jl:364-381).  In the reference such an expression is a LazyExpression that re-evaluates its arguments on every
    call; here the value is cached like any Parameter's but also recomputed whenever a source is dirty or was updated since."""

    def __init__(self, f, sources, model):
        super().__init__(f, model)
        self.sources = [s for s in sources if isinstance(s, Parameter)]
        self._seen = None

    def __call__(self):
        for s in self.sources:
            if isinstance(s, DerivedParameter):
                s()                                      # refresh the chain first: its version tells whether anything below changed
        stale = self.dirty or any(s.dirty for s in self.sources)
        if not stale:
            stale = self._seen != tuple(s.version for s in self.sources)
        if stale:
            self.update()
            self.dirty = False
            self._seen = tuple(s.version for s in self.sources)
        return self.val


class DeviceUniformParameter(Parameter):
    """A Parameter whose value is regenerated ON THE DEVICE at every update: val[i] = scale * U[0,1)(seed + 1000*epoch, i),
    the counter-based stream of SURVEY.md §8(d) — the device-resident analogue of `Parameter(rand!, zeros(n, n), model)`
    (README.md:36-43).  `shape` is (n,) or (rows, cols) (column-major on the device).  Calling it returns a host copy."""

    def __init__(self, shape, seed, model, scale=1.0, advance=True):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.seed, self.scale, self.advance = int(seed), float(scale), advance
        self.epoch = -1
        proto = np.zeros(self.shape if len(self.shape) == 1 else self.shape, dtype=np.float64, order="F")
        super().__init__(lambda v: v, proto, model)
        self.device_resident = True

    def update(self):
        self.epoch += 1
        self.version += 1
        self._host_stale = True

    def current_seed(self):
        return self.seed + (1000 * self.epoch if self.advance else 0)

    def __call__(self):
        super().__call__()
        from .lazyexpression import device_value_of
        dv = device_value_of(self)                       # uploads / regenerates if stale
        if getattr(self, "_host_stale", True):
            ctx = self.model.device()
            if len(self.shape) == 2:
                self.val = dv.fetch(ctx)                              # pitched copy out of the padded device layout
                ctx.synchronize()
            else:
                host = np.empty(int(self.shape[0]), dtype=np.float64)
                ctx.fetch(host, dv.buf, host.nbytes)
                ctx.synchronize()
                self.val = host
            self._host_stale = False
        return self.val


class DeviceUniformSparseParameter(DeviceUniformParameter):
    """A sparse Parameter (scipy.sparse.csc_matrix ↔ Julia SparseMatrixCSC) with a FIXED pattern whose non-zero values are regenerated
    ON THE DEVICE at every update: nzval[t] = offset + scale * U[0,1)(seed + 1000*epoch, t) — BASELINE config 5 with the values resident
    in HBM, like DeviceUniformParameter for dense values.  Calling it returns a host copy with the current values."""

    def __init__(self, pattern, seed, model, scale=1.0, advance=True):
        import scipy.sparse as sp
        pattern = sp.csc_matrix(pattern)
        self.pattern = sp.csc_matrix((np.zeros(pattern.nnz), pattern.indices.copy(), pattern.indptr.copy()), shape=pattern.shape)
        self.shape = tuple(int(s) for s in pattern.shape)
        self.seed, self.scale, self.advance = int(seed), float(scale), advance
        self.epoch = -1
        Parameter.__init__(self, lambda v: v, self.pattern, model)
        self.device_resident = True

    def __call__(self):
        Parameter.__call__(self)
        from .lazyexpression import device_value_of
        dv = device_value_of(self)
        if getattr(self, "_host_stale", True):
            ctx = self.model.device()
            host = np.empty(max(dv.nnz, 1), dtype=np.float64)
            ctx.fetch(host, dv.buf, 8 * dv.nnz)
            ctx.synchronize()
            self.pattern.data[:] = host[:dv.nnz]
            self.val = self.pattern
            self._host_stale = False
        return self.val
