"""parametron.jl_amd — MI355X-native implementation of Parametron.jl's parameter-update hot path.

Host-side mirror of the reference API (Model / Variable / Parameter / @expression / @objective / @constraint /
solve!, src/Parametron.jl:3-36) over the C ABI of libparametron_hip.so (include/parametron_hip.h).  Import as
`import parametron_jl_amd` (the directory name is not a Python identifier; parametron_jl_amd.py is the alias).
"""
from . import _lib  # noqa: F401
from ._lib import ArgumentError, DimensionMismatch, ErrorException  # noqa: F401
from .functions import AffineFunction, LinearTerm, QuadraticFunction, QuadraticTerm, Variable, canonicalize  # noqa: F401
from .parameter import DerivedParameter, DeviceUniformParameter, DeviceUniformSparseParameter, Parameter  # noqa: F401
from .lazyexpression import (LazyExpression, Relation, adjoint, bilinear, dot, expression, getindex, getproperty, lazy, prune_zero, transpose,  # noqa: F401
                             vcat, vect, wrap)
from .hostops import Transpose  # noqa: F401
from . import moi  # noqa: F401
from .model import (AbstractOptimizer, Maximize, Minimize, MockOptimizer, Model, constraint, dualstatus, initialize,  # noqa: F401
                    mock_model, objective, objectivevalue, primalstatus, setdirty, setobjective, solve,
                    terminationstatus, update, value)
from .handoff import DeviceQP  # noqa: F401



def profile_enable(on=True):
    """Per-kernel timing of every launch (the device analogue of findallocs, src/debug.jl:4-23)."""
    _lib.call("pmt_profile_enable", 1 if on else 0)


def profile_report():
    """{kernel: {launches, avg_ms, min_ms, max_ms}} since profile_enable(True)."""
    import ctypes as C
    L = _lib.load()
    n = L.pmt_profile_report(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    L.pmt_profile_report(buf, int(n) + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot, mn, mx = line.split("\t")
        out[name] = {"launches": int(cnt), "avg_ms": float(tot) / max(1, int(cnt)), "min_ms": float(mn), "max_ms": float(mx)}
    return out


def set_host_delivery(mode):
    """How results leave for the host (recorded fetches, band-wise deliveries): 0 / "auto" copy engine when there is one, 1 / "engine" copy
    engine or an error, 2 / "kernels" the kernel-copy fallback (pmt_set_host_delivery)."""
    mode = {"auto": 0, "engine": 1, "kernels": 2}.get(mode, mode)
    _lib.call("pmt_set_host_delivery", int(mode))


def host_delivery(device=0):
    """(mode, copy engine usable on `device`)"""
    import ctypes as C
    mode, engine = C.c_int(0), C.c_int(0)
    _lib.call("pmt_get_host_delivery", int(device), C.byref(mode), C.byref(engine))
    return mode.value, bool(engine.value)


def findallocs(io, expr):
    """findallocs(io, expr) (src/debug.jl:4-23): re-evaluate `expr` and report what it costs.  The reference prints the heap bytes each
    node of the expression tree allocates (they must be 0); on the device the analogue of an allocation is growth of the plan's
    memory (must be 0 after the first evaluation) and the cost of a node is the time of the kernels it launches."""
    from .lazyexpression import DeviceNode, evaluate, schedule
    if not isinstance(expr, DeviceNode):
        value = expr() if callable(expr) else expr
        io.write("%s: host value (%s), no device work\n" % (type(expr).__name__, type(value).__name__))
        return
    ctx = expr.model.device()
    evaluate(ctx, [expr]); ctx.synchronize()                              # first evaluation may size buffers
    before = ctx.bytes_allocated()
    profile_enable(True)
    try:
        expr.model.setdirty()
        evaluate(ctx, [expr]); ctx.synchronize()
        report = profile_report()
    finally:
        profile_enable(False)
    nodes = [x for x in schedule([expr]) if isinstance(x, DeviceNode)]
    io.write("%d device node(s): %s\n" % (len(nodes), ", ".join(x.builder for x in nodes)))
    for name, r in report.items():
        io.write("  %-40s %d launch(es)  %.4f ms\n" % (name, r["launches"], r["avg_ms"] * r["launches"]))
    io.write("plan memory growth during re-evaluation: %d bytes\n" % (ctx.bytes_allocated() - before))
