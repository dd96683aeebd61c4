"""Lazy-expression DAG and its rewrite rules (src/lazyexpression.jl:37-302), compiled to HIP kernel launches.

`lazy(f, *args)` is `optimize_toplevel(LazyExpression(f, args...))` (src/lazyexpression.jl:184-193):
  * no Parameter / LazyExpression among the arguments -> evaluated immediately on the host (hostops.apply);
  * otherwise the matching `optimize` rule (:198-302) creates a DeviceNode whose `dest` lives in
    HBM (allocated once, at node creation) and whose `emit` issues the C-ABI call(s) of include/parametron_hip.h that
    replace the reference's in-place builder (matvecmul!, vecsubtract!, vecdot!, bilinearmul!, mul!, scale!, vcat!, ...).
Evaluating an expression (`expr()`), or update!(model), walks the DAG once in the reference's argument order,
refreshes the dirty Parameters (host callback + H2D copy, or a device fill kernel) and runs each node's kernels.
There is no host fallback for array-valued nodes: an unsupported combination raises ArgumentError.
"""
import ctypes as C

import numpy as np

from . import hostops
from ._lib import LT, QT, ArgumentError, DimensionMismatch
from .device import (DAff, DAffVec, DDenseAff, DLinVec, DMat, DNum, DQuad, DSparseAff, DSpMat, DV, DVars, DVarsAff, DVec, P,
                     fetch_f64, fetch_terms)


def _is_sparse(v):
    return hasattr(v, "indptr") and hasattr(v, "indices") and hasattr(v, "data") and getattr(v, "format", None) == "csc"
from .functions import AffineFunction, LinearTerm, QuadraticFunction, QuadraticTerm, Variable, _isnum
from .hostops import Transpose, _RowTimesMatrix, elem_kind, is_vector
from .parameter import DerivedParameter, Parameter


class Relation:
    """lhs (<=|>=|==) rhs, the argument of @constraint (src/model.jl:224-249)."""

    def __init__(self, lhs, op, rhs):
        self.lhs, self.op, self.rhs = lhs, op, rhs

    def __bool__(self):
        raise ArgumentError("a constraint relation has no truth value; pass it to constraint(model, ...)")


class LazyExpression:
    """Base class: operator syntax + evaluation protocol (src/lazyexpression.jl:37-61)."""

    __array_priority__ = 3000

    def __mul__(self, o): return lazy("*", self, o)
    def __rmul__(self, o): return lazy("*", o, self)
    def __matmul__(self, o): return lazy("*", self, o)
    def __rmatmul__(self, o): return lazy("*", o, self)
    def __add__(self, o): return lazy("+", self, o)
    def __radd__(self, o): return lazy("+", o, self)
    def __sub__(self, o): return lazy("-", self, o)
    def __rsub__(self, o): return lazy("-", o, self)
    def __le__(self, o): return Relation(self, "<=", o)
    def __ge__(self, o): return Relation(self, ">=", o)

    def __eq__(self, o):
        return Relation(self, "==", o)

    __hash__ = object.__hash__

    @property
    def T(self):
        return lazy("adjoint", self)


def _is_lazy(x):
    return isinstance(x, (Parameter, LazyExpression))


# ---------------------------------------------------------------------------------------------------------
# kinds

def kind_of(v):
    """Kind of an evaluated argument (the `argtypes` of optimize_toplevel, src/lazyexpression.jl:187)."""
    if isinstance(v, DV):
        return v.kind
    if _isnum(v):
        return "num"
    if isinstance(v, np.ndarray) and v.dtype != object:
        if v.ndim == 0:
            return "num"
        return "vec" if v.ndim == 1 else ("mat" if v.ndim == 2 else "array")
    if isinstance(v, Variable):
        return "var"
    if isinstance(v, LinearTerm):
        return "lt"
    if isinstance(v, QuadraticTerm):
        return "qt"
    if isinstance(v, AffineFunction):
        return "aff"
    if isinstance(v, QuadraticFunction):
        return "quad"
    if isinstance(v, Transpose):
        return "t" + kind_of(v.parent)
    if isinstance(v, (_RowTimesMatrix, _LazyRowTimesMatrix)):
        return "rowmat"
    if _is_sparse(v):
        return "spmat"                                  # scipy.sparse.csc_matrix ↔ Julia SparseMatrixCSC (config 5)
    if is_vector(v):
        k = elem_kind(v)
        return {"num": "vec", "var": "varvec", "lt": "ltvec", "aff": "affvec", "empty": "vec"}.get(k, "vector<%s>" % k)
    return type(v).__name__


def _model_of(args):
    for a in args:
        if isinstance(a, Parameter):
            return a.model
        if isinstance(a, DeviceNode):
            return a.model
        if isinstance(a, (Transpose,)):
            m = _model_of([a.parent])
            if m is not None:
                return m
        if isinstance(a, _LazyRowTimesMatrix):
            return a.model
    return None


def device_value_of(x, ctx=None):
    """Device mirror of an expression argument; Parameters and constants are uploaded (once / when recomputed)."""
    if isinstance(x, DeviceNode):
        return x.out
    if isinstance(x, Parameter):
        ctx = ctx or x.model.device()
        if getattr(x, "_staged_pending", False) and x._dev is not None:
            # the value of this solve was evaluated and put on the copy stream by Model.stage_parameters(): consume it on the plan's stream
            # (a Parameter only side-lane records read is committed on the side stream: the plan's stream does not wait for its upload)
            side = getattr(x, "_commit_on_side_lane", False) and not (isinstance(x._dev, DMat) and getattr(x._dev, "_staged_kind", None) == "rowmajor")
            if side:
                ctx.commit_lane(1)
            try:
                _commit_staged_value(ctx, x._dev)
            finally:
                if side:
                    ctx.commit_lane(0)
            x._staged_pending = False
            x._dev_version = x.version
            ctx._staging_dirty = True
            _sync_mailbox(ctx, x, getattr(x, "val", None))
            return x._dev
        val = Parameter.__call__(x)                 # evalarg(::Parameter) (src/lazyexpression.jl:51)
        if x._dev is None:
            if getattr(x, "pattern", None) is not None:                   # DeviceUniformSparseParameter: fixed pattern, values made on the device
                x._dev = DSpMat(ctx, x.pattern)
            elif getattr(x, "device_resident", False):
                x._dev = DVec(ctx, x.shape[0]) if len(x.shape) == 1 else DMat(ctx, *x.shape)
            else:
                x._dev = _alloc_like(ctx, val)
        if x._dev_version != x.version and getattr(x, "_in_tape", False) and getattr(ctx, "_refreshing", False) and not ctx.recording:
            # the callback is an entry of the tape (a small model, Model._record_parameter_callbacks): the replay that follows draws the values
            if getattr(x, "_mailbox", None) is not None:
                _sync_mailbox(ctx, x, val)
            else:
                x._seed_word.value = x.current_seed() % (1 << 64)
            x._dev_version = x.version
            return x._dev
        if x._dev_version != x.version:
            if getattr(x, "device_resident", False):
                # a value only side-lane records read is regenerated on the side stream (inside update! only: there the replay joins the
                # side stream back): a transfer at the front of the side lane does not wait for the objective's callbacks on the plan's stream
                side = getattr(x, "_commit_on_side_lane", False) and getattr(ctx, "_refreshing", False) and not ctx.recording
                call = (lambda *a: ctx.call_on_lane(1, *a)) if side else ctx.call
                if isinstance(x._dev, DSpMat):
                    call("pmt_fill_uniform_f64", P(x._dev.buf), int(x._dev.nnz), C.c_uint64(x.current_seed()), x.scale)
                elif isinstance(x._dev, DMat):
                    call("pmt_fill_uniform_matrix_f64", P(x._dev.buf), x._dev.rows, x._dev.cols, x._dev.lda,
                         C.c_uint64(x.current_seed()), x.scale)
                else:
                    call("pmt_fill_uniform_f64", P(x._dev.buf), int(x.shape[0]), C.c_uint64(x.current_seed()), x.scale)
                if not side and getattr(x, "_read_unordered_by_lane3", False) and not ctx.recording:
                    # a recorded transfer at the very front of the side lane (lane 3) reads this buffer WITHOUT waiting for the plan's
                    # stream: a value written on the plan's stream must be complete before the next replay can start it
                    ctx.synchronize()
            else:
                _upload_value(ctx, x._dev, val)
                _sync_mailbox(ctx, x, val)
                if getattr(x, "_read_unordered_by_lane3", False) and not ctx.recording:
                    ctx.synchronize()                   # (a serial upload travels on the plan's stream as well)
            x._dev_version = x.version
        return x._dev
    return const_device_value(ctx, x)


def _sync_mailbox(ctx, x, val):
    """A host-updated Parameter of a small model has a page-locked MAILBOX the tape copies into its device buffer at every replay
    (Model._record_parameter_callbacks): whatever path gives the Parameter a new value keeps the mailbox equal to it."""
    if getattr(x, "_mailbox", None) is None or val is None:
        return
    if getattr(ctx, "_replay_pending", False):
        ctx.synchronize()                                 # the previous replay may still be reading the mailbox
    from .model import _write_mailbox
    _write_mailbox(x, val)


def _alloc_like(ctx, val):
    k = kind_of(val)
    if k == "num":
        return DNum(ctx)
    if k == "vec":
        return DVec(ctx, len(val))
    if k == "mat":
        return DMat(ctx, *np.shape(val))
    if k == "spmat":
        return DSpMat(ctx, val)
    raise ArgumentError("Parameters of type %s cannot be used in device expressions" % k)


def _upload_value(ctx, dv, val):
    if isinstance(dv, DNum):
        ctx.upload(dv.buf, np.array([val], dtype=np.float64))
    elif isinstance(dv, DVec):
        v = np.asarray(val, dtype=np.float64)
        if v.shape != (dv.n,):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.n,), v.shape))
        ctx.upload(dv.buf, v)
    elif isinstance(dv, DMat):
        m = np.asarray(val, dtype=np.float64)
        if m.shape != (dv.rows, dv.cols):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.rows, dv.cols), m.shape))
        dv.upload(ctx, m)                                                   # Julia column-major, padded leading dimension
    elif isinstance(dv, DSpMat):
        if not dv.same_pattern(val):
            raise DimensionMismatch("the sparsity pattern of a sparse Parameter must stay fixed across re-evaluations")
        ctx.upload(dv.buf, np.asarray(val.data, dtype=np.float64))
    else:
        raise ArgumentError("cannot upload into %s" % type(dv).__name__)


def _stage_value(ctx, dv, val):
    """Start the staged upload of a host value (copy stream) into the second buffer of its device mirror."""
    if isinstance(dv, DMat):
        m = np.asarray(val, dtype=np.float64)
        if m.shape != (dv.rows, dv.cols):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.rows, dv.cols), m.shape))
        dv.stage(ctx, m)
        return
    if isinstance(dv, DNum):
        host, nbytes = np.array([val], dtype=np.float64), 8
    elif isinstance(dv, DVec):
        host = np.asarray(val, dtype=np.float64)
        if host.shape != (dv.n,):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.n,), host.shape))
        nbytes = 8 * dv.n
    elif isinstance(dv, DSpMat):
        if not dv.same_pattern(val):
            raise DimensionMismatch("the sparsity pattern of a sparse Parameter must stay fixed across re-evaluations")
        host, nbytes = np.asarray(val.data, dtype=np.float64), 8 * dv.nnz
    else:
        raise ArgumentError("cannot stage an upload into %s" % type(dv).__name__)
    slot = ctx._stage_slot                              # one staging buffer per slot (pmt_plan_stage_slot)
    if not hasattr(dv, "_staging_slots"):
        dv._staging_slots = {}
    if slot not in dv._staging_slots:
        dv._staging_slots[slot] = ctx.alloc(max(nbytes, 8))
    dv._staged_bytes, dv._staged_slot = nbytes, slot
    ctx.stage_upload(dv._staging_slots[slot], host)


def _commit_staged_value(ctx, dv):
    if isinstance(dv, DMat):
        dv.commit(ctx)
    elif getattr(dv, "_staged_bytes", 0):
        ctx.commit_staged(dv.buf, dv._staging_slots[dv._staged_slot], dv._staged_bytes)


def const_device_value(ctx, x):
    """Upload a constant (Parameter-free) argument once."""
    k = kind_of(x)
    if k == "num":
        return DNum(ctx, float(x))
    if k == "vec":
        v = np.asarray(x, dtype=np.float64)
        d = DVec(ctx, len(v)); ctx.upload(d.buf, v); return d
    if k == "mat":
        m = np.asarray(x, dtype=np.float64)
        d = DMat(ctx, *m.shape); _upload_value(ctx, d, m); return d
    if k == "var":
        return DVars(ctx, [x])
    if k == "lt":                                       # LinearTerm constant (2 * x): an AffineFunction of one term, copyto! :420
        return const_device_value(ctx, AffineFunction.of(x))
    if k == "qt":                                       # QuadraticTerm constant (x^2, 3 * x * y): copyto! :432-434
        return const_device_value(ctx, QuadraticFunction.of(x))
    if k == "varvec":
        return DVars(ctx, list(x))
    if k == "aff":
        t, c = x.to_arrays()
        d = DAff(ctx, len(t)); ctx.upload(d.terms, t); ctx.upload(d.const, np.array([c])); return d
    if k == "quad":
        q, l, c = x.to_arrays()
        d = DQuad(ctx, len(q), len(l)); ctx.upload(d.quad, q); ctx.upload(d.lin, l); ctx.upload(d.const, np.array([c])); return d
    if k == "affvec":
        parts = [f.to_arrays() for f in x]
        row_ptr = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum([len(t) for t, _ in parts], out=row_ptr[1:])
        d = DAffVec(ctx, len(parts), row_ptr=row_ptr)
        if d.nterms:
            ctx.upload(d.terms, np.concatenate([t for t, _ in parts]))
        ctx.upload(d.consts, np.array([c for _, c in parts], dtype=np.float64))
        return d
    raise ArgumentError("constants of kind %s cannot enter a device expression" % k)


# ---------------------------------------------------------------------------------------------------------
# nodes

class DeviceNode(LazyExpression):
    """A rewritten (`optimize`d) and wrapped LazyExpression: in-place builder + pre-allocated dest, here in HBM."""

    def __init__(self, model, builder, inputs, out, emit, gram_candidate=None, prepare=None):
        self.model = model
        self.builder = builder            # name of the reference builder this node replaces
        self.inputs = inputs              # Parameters / DeviceNodes in the reference's argument order
        self.out = out
        self._emit = emit
        self._prepare = prepare           # run for every scheduled node before any emit (materialisation requests)
        self.gram_candidate = gram_candidate

    def prepare(self):
        if self._prepare is not None:
            self._prepare()

    def __repr__(self):
        return "LazyExpression{FunctionWrapper{…}(LazyExpression{%s, …}(…))}(…)" % self.builder

    def emit(self, ctx):
        self._emit(ctx)

    def canonicalize(self):
        """canonicalize(expr) as a lazy device node (sorted, duplicates combined; src/functions.jl:269-272, 381-386)."""
        return lazy("canonicalize", self)

    def __call__(self):
        """expr(): evaluate the DAG below this node and return the native value (fetched from the device)."""
        ctx = self.model.device()
        if isinstance(self.out, DQuad):
            self.out.materialize()
        if hasattr(self.out, "require_terms"):                   # implicit dense / bounds / sparse blocks
            self.out.require_terms()
        evaluate(ctx, [self])
        val = fetch_value(ctx, self.out)
        return val


def schedule(roots):
    """Post-order over the DAG in argument order = the order in which the reference first evaluates each Parameter and
    node (src/lazyexpression.jl:50-61); shared sub-expressions appear once (the reference recomputes them, with the
    same result)."""
    seen, order = set(), []

    def visit(x):
        if id(x) in seen:
            return
        seen.add(id(x))
        if isinstance(x, DeviceNode):
            for a in x.inputs:
                visit(a)
            order.append(x)
        elif isinstance(x, Parameter):
            order.append(x)
    for r in roots:
        visit(r)
    return order


def evaluate(ctx, roots):
    order = schedule(roots)
    for x in order:
        if isinstance(x, DeviceNode):
            x.prepare()
    for x in order:
        if isinstance(x, Parameter):
            device_value_of(x, ctx)
    for x in order:
        if isinstance(x, DeviceNode):
            x.emit(ctx)


def fetch_value(ctx, dv):
    """Native value of a device descriptor as the host types of functions.py."""
    if isinstance(dv, DNum):
        v = fetch_f64(ctx, dv.buf, 1); ctx.synchronize(); return float(v[0])
    if isinstance(dv, DVec):
        v = fetch_f64(ctx, dv.buf, dv.n); ctx.synchronize(); return v
    if isinstance(dv, DMat):
        v = dv.fetch(ctx); ctx.synchronize(); return v
    if isinstance(dv, DLinVec):
        t = fetch_terms(ctx, dv.terms, dv.n, LT); ctx.synchronize()
        return [LinearTerm(float(c), Variable(int(v))) for c, v in zip(t["coeff"], t["var"])]
    if isinstance(dv, DAffVec):
        if dv.terms is None:
            raise ArgumentError("this node is fused into its consumer and has no materialised value; it was created "
                                "before anyone asked for it — call the expression once before initialize!(model)")
        t = fetch_terms(ctx, dv.terms, dv.nterms, LT)
        c = fetch_f64(ctx, dv.consts, dv.rows)
        ctx.synchronize()
        rp = dv.host_row_ptr()
        return [AffineFunction.from_arrays(t[rp[i]:rp[i + 1]], c[i]) for i in range(dv.rows)]
    if isinstance(dv, DAff):
        t = fetch_terms(ctx, dv.terms, dv.nterms, LT)
        c = fetch_f64(ctx, dv.const, 1)
        ctx.synchronize()
        return AffineFunction.from_arrays(t, c[0])
    if isinstance(dv, DQuad):
        q = fetch_terms(ctx, dv.quad, dv.nq, QT)
        l = fetch_terms(ctx, dv.lin, dv.nl, LT)
        c = fetch_f64(ctx, dv.const, 1)
        ctx.synchronize()
        return QuadraticFunction.from_arrays(q, l, c[0])
    raise ArgumentError("cannot fetch %s" % type(dv).__name__)


# ---------------------------------------------------------------------------------------------------------
# rule helpers

def _dv(ctx, x):
    """device value of an argument at rule time (Parameters are evaluated once, like `evalarg` at :187)."""
    return device_value_of(x, ctx)


def _inputs(*args):
    out = []
    for a in args:
        if isinstance(a, (Parameter, DeviceNode)):
            out.append(a)
        elif isinstance(a, Transpose):
            out.extend(_inputs(a.parent))
        elif isinstance(a, _LazyRowTimesMatrix):
            out.extend(_inputs(a.x, a.Q))
    return out


def _affvec_parts(dv):
    """(terms, row_ptr_buf, row_len, consts) of a vector operand of vecadd!/vecsubtract!/vcat!."""
    if isinstance(dv, DVec):
        return None, None, 0, dv.buf                                 # numbers: constants only
    if isinstance(dv, DVars):
        return dv.lt(), None, 1, None                                # Variables: one (1.0, var) term, no constant
    if isinstance(dv, DAffVec):
        m = dv.materialized()
        return m.terms, m.row_ptr_buf, m.row_len, m.consts
    raise ArgumentError("not a vector operand: %s" % type(dv).__name__)


def _vec_len(dv):
    return dv.n if isinstance(dv, (DVec, DVars, DLinVec)) else dv.rows


def _row_lens(dv):
    if isinstance(dv, DVec):
        return np.zeros(dv.n, dtype=np.int64)
    if isinstance(dv, DVars):
        return np.ones(dv.n, dtype=np.int64)
    return np.diff(dv.host_row_ptr())


# ---- A * x ------------------------------------------------------------------------------------------------
def _emit_sparse_terms(c, out):
    """materialise a DSparseAff as native LinearTerms + constants (only if some consumer asked for them)"""
    if not out.need_terms:
        return
    sp = out.spmat
    if sp.block_cw:
        c.call("pmt_sparse_assemble_blocks_f64", P(sp.buf), P(sp.block_desc_buf), P(sp.block_idx_buf), P(sp.block_band_buf), P(out.xvars.buf),
               sp.rows, sp.cols, sp.nnz, sp.block_cw, P(out.vec.buf) if out.vec is not None else None, out.sign if out.vec is not None else 0,
               P(out.terms), P(out.consts) if out.vec is not None else None)
        return
    else:
        c.call("pmt_sparse_assemble_slabs_u32_f64" if sp.narrow else "pmt_sparse_assemble_slabs_f64", P(sp.buf), P(sp.perm_buf), P(out.term_var_buf),
               P(sp.slab_ptr_buf), sp.rows, sp.nslab, P(out.terms))
    if out.vec is not None:
        c.call("pmt_consts_f64", P(out.vec.buf), out.rows, out.sign, P(out.consts))


def _rule_spmatvec(model, ctx, A, x):
    """C * x for a sparse C with a fixed pattern (BASELINE config 5): terms for the structural non-zeros only, in the
    reference's row-major matvecmul! order (src/functions.jl:790-796 restricted to the pattern)."""
    dA, dx = _dv(ctx, A), _dv(ctx, x)
    if not isinstance(dx, DVars):
        raise ArgumentError("sparse matrix * %s is not supported" % kind_of(dx))
    if dA.cols != dx.n:
        raise DimensionMismatch("matvecmul!: size(A, 2) != length(x)")
    out = DSparseAff(ctx, dA, dx, None, 0)
    return DeviceNode(model, "matvecmul!", _inputs(A, x), out, lambda c: _emit_sparse_terms(c, out))


def _rule_matvec(model, ctx, A, x):
    dA, dx = _dv(ctx, A), _dv(ctx, x)
    if dA.cols != _vec_len(dx):
        raise DimensionMismatch("matvecmul!: size(A, 2) != length(x)")          # src/functions.jl:781
    if isinstance(dx, DVars):                                                    # rule :200-204, builder :775-798
        out = DDenseAff(ctx, dA, dx, None, 0)

        def emit(c):
            if out.need_terms:
                c.call("pmt_affine_assemble_f64", P(dA.buf), dA.lda, dA.rows, dA.cols, P(dx.buf), None, 0, P(out.terms), P(out.consts))
        return DeviceNode(model, "matvecmul!", _inputs(A, x), out, emit)
    if isinstance(dx, DAffVec):                                                  # builder :800-822
        X = dx.materialized()
        if not X.uniform():
            raise ArgumentError("matrix * Vector{AffineFunction} needs rows of equal length on the device")
        out = DAffVec(ctx, dA.rows, row_len=dA.cols * X.row_len)

        def emit(c):
            c.call("pmt_matvecmul_affs_f64", P(dA.buf), dA.lda, dA.rows, dA.cols, P(X.terms), X.row_len, P(X.consts), P(out.terms), P(out.consts))
        return DeviceNode(model, "matvecmul!", _inputs(A, x), out, emit)
    raise ArgumentError("matrix * %s is not supported" % kind_of(dx))


def _rule_adjoint_matrix(model, ctx, A):                                         # rule :206-217
    dA = _dv(ctx, A)
    out = DMat(ctx, dA.cols, dA.rows)

    def emit(c):
        c.call("pmt_transpose_f64", P(dA.buf), dA.lda, dA.rows, dA.cols, P(out.buf), out.lda)
    return DeviceNode(model, "adjoint", _inputs(A), out, emit)


# ---- x (+|-) y on vectors -------------------------------------------------------------------------------------
def _rule_vec_addsub(model, ctx, a, b, sign):                                    # rules :238-258, builders :751-764
    da, db = _dv(ctx, a), _dv(ctx, b)
    if _vec_len(da) != _vec_len(db):
        raise DimensionMismatch("vecadd!/vecsubtract!: lengths differ")          # src/functions.jl:755
    name = "vecadd!" if sign > 0 else "vecsubtract!"
    # fused dense forms: (A*x) (+|-) b  and  x (+|-) v
    if isinstance(da, DDenseAff) and da.vec is None and isinstance(db, DVec):
        out = DDenseAff(ctx, da.mat, da.xvars, db, sign)

        def emit(c):
            if out.need_terms:
                c.call("pmt_affine_assemble_f64", P(out.mat.buf), out.mat.lda, out.mat.rows, out.mat.cols, P(out.xvars.buf), P(db.buf), sign,
                       P(out.terms), P(out.consts))
        inner = a.inputs if isinstance(a, DeviceNode) else _inputs(a)
        return DeviceNode(model, name, list(inner) + _inputs(b), out, emit)
    if isinstance(da, DSparseAff) and da.vec is None and isinstance(db, DVec):
        out = DSparseAff(ctx, da.spmat, da.xvars, db, sign)
        inner = a.inputs if isinstance(a, DeviceNode) else _inputs(a)
        return DeviceNode(model, name, list(inner) + _inputs(b), out, lambda c: _emit_sparse_terms(c, out))
    if isinstance(da, DVars) and isinstance(db, DVec):
        out = DVarsAff(ctx, da, db, sign)

        def emit(c):
            if out.need_terms:
                c.call("pmt_vars_addsub_f64", P(da.buf), da.n, P(db.buf), sign, None, 0, P(out.terms), None, P(out.consts))
        return DeviceNode(model, name, _inputs(a, b), out, emit)
    if isinstance(da, DVec) and isinstance(db, DVec):
        raise ArgumentError("number vector (+|-) number vector is plain data: compute it inside a Parameter callback")
    ta, pa, la, ca = _affvec_parts(da)
    tb, pb, lb, cb = _affvec_parts(db)
    lens = _row_lens(da) + _row_lens(db)
    row_ptr = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    out = DAffVec(ctx, len(lens), row_ptr=row_ptr)

    def emit(c):
        c.call("pmt_affvec_combine_f64", out.rows, P(ta), P(pa), la, P(ca), P(tb), P(pb), lb, P(cb), 1 if sign > 0 else -1,
               P(out.terms), P(out.row_ptr_buf), out.row_len, P(out.consts))
    return DeviceNode(model, name, _inputs(a, b), out, emit)


# ---- dot(x, y) ------------------------------------------------------------------------------------------------
def _rule_dot(model, ctx, x, y):                                                 # rule :228-232, builders :665-731
    dx, dy = _dv(ctx, x), _dv(ctx, y)
    if _vec_len(dx) != _vec_len(dy):
        raise DimensionMismatch("vecdot!: lengths differ")                       # src/functions.jl:669,680,693,704
    n = _vec_len(dx)
    ins = _inputs(x, y)
    if isinstance(dx, DVec) and isinstance(dy, DVec):
        raise ArgumentError("dot of two number vectors is plain data: compute it inside a Parameter callback")
    if isinstance(dx, DVec) or isinstance(dy, DVec):
        v, o = (dx, dy) if isinstance(dx, DVec) else (dy, dx)
        if isinstance(o, DVars):                                                 # :676-687
            out = DAff(ctx, n)

            def emit(c):
                c.call("pmt_vecdot_numbers_vars_f64", P(v.buf), P(o.buf), n, P(out.terms), P(out.const))
            return DeviceNode(model, "vecdot!", ins, out, emit)
        X = o.materialized()                                                     # :665-674
        if not X.uniform():
            raise ArgumentError("dot(numbers, Vector{AffineFunction}) needs rows of equal length on the device")
        out = DAff(ctx, n * X.row_len)

        def emit(c):
            c.call("pmt_vecdot_numbers_affs_f64", P(v.buf), n, P(X.terms), X.row_len, P(X.consts), P(out.terms), P(out.const))
        return DeviceNode(model, "vecdot!", ins, out, emit)
    if isinstance(dx, DVars) and isinstance(dy, DVars):                          # :689-700
        out = DQuad(ctx, n, 0)

        def emit(c):
            c.call("pmt_vecdot_terms_f64", n, None, P(dx.buf), None, P(dy.buf), 0, None, P(out.quad))
        return DeviceNode(model, "vecdot!", ins, out, emit)
    if isinstance(dx, DVars) or isinstance(dy, DVars):                           # :702-709 over :537-546
        o, v = (dx, dy) if isinstance(dy, DVars) else (dy, dx)
        X = o.materialized()
        if not X.uniform():
            raise ArgumentError("dot(Vector{AffineFunction}, Vector{Variable}) needs rows of equal length on the device")
        out = DQuad(ctx, n * X.row_len, n)

        def emit(c):
            c.call("pmt_vecdot_affs_vars_f64", n, P(X.terms), X.row_len, P(X.consts), P(v.buf), 0, None, P(out.quad), P(out.lin))
        return DeviceNode(model, "vecdot!", ins, out, emit)
    if isinstance(dx, DAffVec) and isinstance(dy, DAffVec):                      # :702-709 over :548-576 (HOT LOOP 3)
        if not (dx.uniform() and dy.uniform()):
            raise ArgumentError("dot of two Vector{AffineFunction} needs rows of equal length on the device")
        nx, ny = dx.row_len, dy.row_len
        out = DQuad(ctx, n * nx * ny, n * (nx + ny), alloc=False)                # literal buffers only on demand

        def prepare():
            if out.quad is not None:                                             # literal form wanted: operands must be materialised
                dx.materialized(); dy.materialized()

        def emit(c):
            if out.quad is None:
                return                                                           # consumed by the canonical objective instead
            c.call("pmt_quad_expand_f64", n, P(dx.terms), nx, P(dx.consts), P(dy.terms), ny, P(dy.consts), 0, None,
                   P(out.quad), P(out.lin), P(out.const))
        gram = dx if (dx is dy and isinstance(dx, DDenseAff)) else None
        return DeviceNode(model, "vecdot!", ins, out, emit, gram_candidate=gram, prepare=prepare)
    raise ArgumentError("dot(%s, %s) is not supported on the device" % (kind_of(dx), kind_of(dy)))


class _LazyRowTimesMatrix:
    """transpose(x) * Q with Q a Parameter/expression, waiting for its right factor."""

    def __init__(self, model, x, Q):
        self.model, self.x, self.Q = model, x, Q

    def __mul__(self, o):
        return lazy("*", self, o)

    __matmul__ = __mul__


def _rule_bilinear(model, ctx, x, Q, y):                                         # rule :219-226, builder :840-858
    dx, dQ, dy = _dv(ctx, x), _dv(ctx, Q), _dv(ctx, y)
    if not (isinstance(dx, DVars) and isinstance(dy, DVars) and isinstance(dQ, DMat)):
        raise ArgumentError("transpose(x) * Q * y needs Variable vectors and a matrix")
    if (dQ.rows, dQ.cols) != (dx.n, dy.n):
        raise DimensionMismatch("bilinearmul!: size(Q) != (length(x), length(y))")   # src/functions.jl:845
    out = DQuad(ctx, dx.n * dy.n, 0)

    def emit(c):
        c.call("pmt_bilinear_f64", P(dQ.buf), dQ.lda, dQ.rows, dQ.cols, P(dx.buf), P(dy.buf), 0, None, P(out.quad))
    return DeviceNode(model, "bilinearmul!", _inputs(x, Q, y), out, emit)


# ---- scalar add!/subtract! ------------------------------------------------------------------------------------
def _scalar_affine_part(ctx, dv):
    """(terms, nterms, const) of the affine part of a scalar operand."""
    if isinstance(dv, DNum):
        return None, 0, dv.buf
    if isinstance(dv, DVars):
        return dv.lt(), dv.n, None
    if isinstance(dv, DAff):
        return dv.terms, dv.nterms, dv.const
    if isinstance(dv, DQuad):
        dv.materialize()
        return dv.lin, dv.nl, dv.const
    raise ArgumentError("not a scalar operand: %s" % type(dv).__name__)


def _rule_scalar_addsub(model, ctx, a, b, sign):                                 # rules :238-258, builders :452-502
    da, db = _dv(ctx, a), _dv(ctx, b)
    name = "add!" if sign > 0 else "subtract!"
    sb = 1 if sign > 0 else -1
    ta, na, ca = _scalar_affine_part(ctx, da)
    tb, nb, cb = _scalar_affine_part(ctx, db)
    if isinstance(da, DQuad) or isinstance(db, DQuad):
        qa, nqa = (da.quad, da.nq) if isinstance(da, DQuad) else (None, 0)
        qb, nqb = (db.quad, db.nq) if isinstance(db, DQuad) else (None, 0)
        out = DQuad(ctx, nqa + nqb, na + nb)

        def emit(c):
            c.call("pmt_quad_combine_f64", P(qa), nqa, P(qb), nqb, sb, P(out.quad))
            c.call("pmt_affvec_combine_f64", 1, P(ta), None, na, P(ca), P(tb), None, nb, P(cb), sb, P(out.lin), None, na + nb, P(out.const))
        return DeviceNode(model, name, _inputs(a, b), out, emit)
    out = DAff(ctx, na + nb)

    def emit(c):
        c.call("pmt_affvec_combine_f64", 1, P(ta), None, na, P(ca), P(tb), None, nb, P(cb), sb, P(out.terms), None, na + nb, P(out.const))
    return DeviceNode(model, name, _inputs(a, b), out, emit)


# ---- scalar mul! and vector scale! ----------------------------------------------------------------------------------
def _rule_mul_scalar(model, ctx, a, b):                                          # rules :260-274, builder :578 over :515-576
    da, db = _dv(ctx, a), _dv(ctx, b)
    ins = _inputs(a, b)
    if isinstance(da, DNum) or isinstance(db, DNum):
        s, f = (da, db) if isinstance(da, DNum) else (db, da)
        if isinstance(f, DVars) and f.n == 1:
            # Number * Variable -> LinearTerm (src/functions.jl:114-116): no `optimize` rule of its own in the reference (the generic
            # rule :198 calls `*` out of place, isbits result); on the device an AffineFunction of one term, constant 0
            out = DAff(ctx, 1)

            def emit(c):
                c.call("pmt_scale_vars_f64", P(f.buf), 1, P(s.buf), 0.0, P(out.terms))
            return DeviceNode(model, "*", ins, out, emit)
        if isinstance(f, DNum):
            raise ArgumentError("number * number is plain data: compute it inside a Parameter callback")
        if isinstance(f, DAff):
            out = DAff(ctx, f.nterms)

            def emit(c):
                c.call("pmt_affvec_scale_f64", 1, f.nterms, P(f.terms), P(f.const), P(s.buf), 0.0, P(out.terms), P(out.const))
            return DeviceNode(model, "mul!", ins, out, emit)
        if isinstance(f, DQuad):
            f.materialize()
            out = DQuad(ctx, f.nq, f.nl)

            def emit(c):
                c.call("pmt_quad_scale_f64", P(f.quad), f.nq, P(s.buf), 0.0, P(out.quad))
                c.call("pmt_affvec_scale_f64", 1, f.nl, P(f.lin), P(f.const), P(s.buf), 0.0, P(out.lin), P(out.const))
            return DeviceNode(model, "mul!", ins, out, emit)
    if isinstance(da, DAff) and isinstance(db, DAff):                            # aff * aff (:548-576)
        out = DQuad(ctx, da.nterms * db.nterms, da.nterms + db.nterms)

        def emit(c):
            c.call("pmt_quad_expand_f64", 1, P(da.terms), da.nterms, P(da.const), P(db.terms), db.nterms, P(db.const), 0, None,
                   P(out.quad), P(out.lin), P(out.const))
        return DeviceNode(model, "mul!", ins, out, emit)
    if (isinstance(da, DAff) and isinstance(db, DVars) and db.n == 1) or (isinstance(db, DAff) and isinstance(da, DVars) and da.n == 1):
        f, v = (da, db) if isinstance(da, DAff) else (db, da)                    # aff * Variable (:537-546)
        out = DQuad(ctx, f.nterms, 1)

        def emit(c):
            c.call("pmt_vecdot_affs_vars_f64", 1, P(f.terms), f.nterms, P(f.const), P(v.buf), 0, None, P(out.quad), P(out.lin))
        return DeviceNode(model, "mul!", ins, out, emit)
    raise ArgumentError("%s * %s is not supported on the device" % (kind_of(da), kind_of(db)))


def _rule_scale(model, ctx, s, v):                                               # rules :284-290, builder :873-925
    ds, dv = _dv(ctx, s), _dv(ctx, v)
    ins = _inputs(s, v)
    if isinstance(dv, DVars):
        out = DLinVec(ctx, dv.n)

        def emit(c):
            c.call("pmt_scale_vars_f64", P(dv.buf), dv.n, P(ds.buf), 0.0, P(out.terms))
        return DeviceNode(model, "scale!", ins, out, emit)
    if isinstance(dv, DVec):
        out = DVec(ctx, dv.n)

        def emit(c):
            c.call("pmt_scale_numbers_f64", P(dv.buf), dv.n, P(ds.buf), 0.0, P(out.buf))
        return DeviceNode(model, "scale!", ins, out, emit)
    if isinstance(dv, DAffVec):
        Y = dv.materialized()
        out = DAffVec(ctx, Y.rows, row_ptr=Y.host_row_ptr())

        def emit(c):
            c.call("pmt_affvec_scale_f64", Y.rows, Y.nterms, P(Y.terms), P(Y.consts), P(ds.buf), 0.0, P(out.terms), P(out.consts))
        return DeviceNode(model, "scale!", ins, out, emit)
    raise ArgumentError("number * %s is not supported on the device" % kind_of(dv))


# ---- vcat / vect / convert --------------------------------------------------------------------------------------
def _rule_vcat(model, ctx, *vs):                                                 # rule :276-278, builder :969-994
    dvs = [_dv(ctx, v) for v in vs]
    for d in dvs:
        if not isinstance(d, (DAffVec, DVars)):
            raise ArgumentError("vcat of %s is not supported on the device" % kind_of(d))
    lens = np.concatenate([_row_lens(d) for d in dvs]) if dvs else np.zeros(0, dtype=np.int64)
    row_ptr = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    out = DAffVec(ctx, len(lens), row_ptr=row_ptr)
    if out.row_ptr_buf is None:            # uniform result: still address pieces through an explicit row_ptr
        out.row_ptr_all = ctx.upload_new(row_ptr)
    else:
        out.row_ptr_all = out.row_ptr_buf
    parts = [_affvec_parts(d) for d in dvs]
    offsets = np.concatenate([[0], np.cumsum([_vec_len(d) for d in dvs])]).astype(np.int64)

    def emit(c):
        for (t, p, l, cst), d, r0 in zip(parts, dvs, offsets[:-1]):
            rows = _vec_len(d)
            if rows == 0:
                continue
            zero = None
            c.call("pmt_affvec_combine_f64", rows, P(t), P(p), l, P(cst), None, None, 0, zero, 1,
                   P(out.terms), P(out.row_ptr_all + 8 * int(r0)), 0, P(out.consts + 8 * int(r0)))
    return DeviceNode(model, "vcat!", _inputs(*vs), out, emit)


def _rule_vect(model, ctx, x):                                                   # rule :292-298
    dx = _dv(ctx, x)
    if isinstance(dx, DAff):
        out = DAffVec(ctx, 1, row_len=dx.nterms, alloc=False)
        out.terms, out.consts = dx.terms, dx.const                               # one-element vector aliasing the scalar's buffers
        return DeviceNode(model, "vect", _inputs(x), out, lambda c: None)
    raise ArgumentError("[%s] is not supported on the device" % kind_of(dx))



# ---- canonicalize! on the device ------------------------------------------------------------------------------------
def _canonical_order(kind, *index_arrays):
    """pmt_canonical_order_* (host, once): (perm, seg_ptr, out index arrays)."""
    from . import _lib
    n = len(index_arrays[0])
    vp = C.c_void_p
    perm = np.empty(max(n, 1), dtype=np.int64)
    seg_ptr = np.empty(n + 1, dtype=np.int64)
    outs = [np.empty(max(n, 1), dtype=np.int64) for _ in index_arrays]
    nseg = C.c_int64()
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in index_arrays]
    name = "pmt_canonical_order_affine" if kind == "aff" else "pmt_canonical_order_quadratic"
    _lib.call(name, n, *[a.ctypes.data_as(vp) for a in arrs], perm.ctypes.data_as(vp), seg_ptr.ctypes.data_as(vp),
              *[o.ctypes.data_as(vp) for o in outs], C.byref(nseg))
    s = nseg.value
    return perm[:n], seg_ptr[:s + 1], [o[:s] for o in outs]


def _rule_canonicalize(model, ctx, x):
    """canonicalize(f) for an AffineFunction / QuadraticFunction node (src/functions.jl:269-272, 381-386).  The indices of
    the operand are read back ONCE here (they never change afterwards); each re-evaluation is a segmented coefficient sum."""
    dx = _dv(ctx, x)
    if not isinstance(x, DeviceNode) or not isinstance(dx, (DAff, DQuad)):
        raise ArgumentError("canonicalize on the device needs an AffineFunction or QuadraticFunction expression")
    if isinstance(dx, DQuad):
        dx.materialize()
    evaluate(ctx, [x])                                                     # once (prepare + emit): the (static) indices are in HBM now

    def order(terms_ptr, nterms, dtype):
        """(perm, seg_ptr) device buffers, nseg, and an initialiser for the output terms.  The ordering is computed on the device from
        the term buffer where it lies (pmt_canonical_order_device: radix sort + run boundaries; only the run count comes back); indices
        that do not fit the packed key (>= 2^32) take the host ordering instead."""
        nbytes = dtype.itemsize
        dperm, dseg = ctx.alloc(8 * max(nterms, 1)), ctx.alloc(8 * (nterms + 1))
        nseg = C.c_int64()
        try:
            ctx.synchronize()
            ctx.call_now("pmt_canonical_order_device", P(terms_ptr), nterms, nbytes, P(dperm), P(dseg), C.byref(nseg))
        except ArgumentError as e:
            if "2^32" not in str(e):                                    # only the too-large-index case has a host fallback; anything else
                raise                                                   # (null pointer, bad term size) is an error, not a slow path
            # the host-computed perm / seg reuse the two buffers allocated above (same sizes), nothing is left behind
            t = fetch_terms(ctx, terms_ptr, nterms, dtype); ctx.synchronize()
            if dtype is LT:
                perm, seg, (ov,) = _canonical_order("aff", t["var"])
                init = np.zeros(len(ov), dtype=LT); init["var"] = ov
            else:
                perm, seg, (orow, ocol) = _canonical_order("quad", t["row"], t["col"])
                init = np.zeros(len(orow), dtype=QT); init["row"] = orow; init["col"] = ocol
            ctx.upload(dperm, np.ascontiguousarray(perm, dtype=np.int64)); ctx.upload(dseg, np.ascontiguousarray(seg, dtype=np.int64))
            return dperm, dseg, len(init), lambda out_ptr: ctx.upload(out_ptr, init)
        n = int(nseg.value)
        return dperm, dseg, n, lambda out_ptr: ctx.call_now("pmt_canonical_init_terms", P(terms_ptr), nbytes, P(dperm), P(dseg), n, P(out_ptr))

    if isinstance(dx, DAff):
        dperm, dseg, nseg, init = order(dx.terms, dx.nterms, LT)
        out = DAff(ctx, nseg)
        init(out.terms)

        def emit(c):
            c.call("pmt_segment_sum_f64", P(dx.terms), 16, P(dperm), P(dseg), nseg, P(out.terms), 16)
            c.call("pmt_copy_bytes", P(out.const), P(dx.const), 8)
        return DeviceNode(model, "canonicalize!", [x], out, emit)
    dqperm, dqseg, nq, qinit = order(dx.quad, dx.nq, QT)
    dlperm, dlseg, nl, linit = order(dx.lin, dx.nl, LT)
    out = DQuad(ctx, nq, nl)
    qinit(out.quad); linit(out.lin)

    def emit(c):
        c.call("pmt_segment_sum_f64", P(dx.quad), 24, P(dqperm), P(dqseg), nq, P(out.quad), 24)
        c.call("pmt_segment_sum_f64", P(dx.lin), 16, P(dlperm), P(dlseg), nl, P(out.lin), 16)
        c.call("pmt_copy_bytes", P(out.const), P(dx.const), 8)
    return DeviceNode(model, "canonicalize!", [x], out, emit)


# ---------------------------------------------------------------------------------------------------------
# optimize_toplevel

VECTOR_KINDS = {"vec", "varvec", "affvec", "ltvec"}
SCALAR_FUNC_KINDS = {"aff", "quad", "var", "lt", "qt"}


def _arg_kind(a):
    if isinstance(a, Parameter):
        if getattr(a, "pattern", None) is not None:
            return "spmat"
        if getattr(a, "device_resident", False):
            return "vec" if len(a.shape) == 1 else "mat"
        return kind_of(a())                           # evaluates the Parameter once, like evalarg at :187
    if isinstance(a, Transpose):
        return "t" + _arg_kind(a.parent)
    if isinstance(a, DeviceNode):
        return a.out.kind
    return kind_of(a)


def _host_value(a):
    """Host value of an argument of a Parameter-only (plain data) expression.  A DeviceNode argument (only adjoint nodes of matrix
    Parameters produce plain data) is EVALUATED, not just fetched: its buffer is written by its own emit, which nobody else runs when
    the node feeds nothing but derived data (fetching alone returned zeros on the first call and the previous solve's value later)."""
    if isinstance(a, (Parameter, DeviceNode)):
        return a()
    if isinstance(a, Transpose) and _is_lazy(a.parent):
        return Transpose(_host_value(a.parent))
    return a


def _source_parameters(args):
    """The Parameters a derived value depends on, through DeviceNode arguments too (DerivedParameter follows their versions)."""
    out, seen = [], set()

    def visit(a):
        if id(a) in seen:
            return
        seen.add(id(a))
        if isinstance(a, Parameter):
            out.append(a)
        elif isinstance(a, DeviceNode):
            for i in a.inputs:
                visit(i)
        elif isinstance(a, Transpose):
            visit(a.parent)
    for a in args:
        visit(a)
    return out


def lazy(f, *args):
    """optimize_toplevel(LazyExpression(f, args...)) — src/lazyexpression.jl:184-193."""
    if not any(_is_lazy(a) or isinstance(a, _LazyRowTimesMatrix) or (isinstance(a, Transpose) and _is_lazy(a.parent)) for a in args):
        return f(*args) if callable(f) else hostops.apply(f, *args)             # :189-192
    model = _model_of(args)
    if model is None:
        raise ArgumentError("expression contains no Parameter with a model")
    if callable(f):
        # any other function of Parameters (user functions, hcat, getindex, reshape, ...: test/lazyexpression.jl:63-82, 384-404):
        # the generic rule :198 keeps the call as it is — host data preparation, re-evaluated when a source changed
        fn, fargs = f, args

        def call():
            return fn(*[_host_value(a) for a in fargs])
        return DerivedParameter(call, _source_parameters(args), model)
    if f == "getproperty":                                                       # rule :300-302 (GetField): p.x as plain derived data
        obj, name = args
        return DerivedParameter(lambda: getattr(_host_value(obj), name), _source_parameters([obj]), model)
    kinds = [_arg_kind(a) for a in args]
    if f in ("+", "-") and len(args) > 2 and f == "+":                           # rule :234-236
        return lazy("+", lazy("+", args[0], args[1]), *args[2:])
    if all(k in ("num", "vec", "mat") for k in kinds) and f != "adjoint":
        # No decision variable involved (e.g. `p ⋅ p`, test/model.jl:161): plain data derived from Parameters.  The
        # reference leaves such calls unoptimised (generic rule :198, an allocating host call); here they become a derived
        # out-of-place Parameter — host data preparation, like any user callback — whose value is uploaded when it changes.
        host_args = args

        def recompute():
            return hostops.apply(f, *[_host_value(a) for a in host_args])
        return DerivedParameter(recompute, _source_parameters(args), model)
    ctx = model.device()
    if f == "*":
        if len(args) == 3 and kinds[0] == "tvarvec":
            return _rule_bilinear(model, ctx, args[0].parent, args[1], args[2])
        a, b = args
        ka, kb = kinds
        if ka == "mat" and kb in ("varvec", "affvec"):
            return _rule_matvec(model, ctx, a, b)
        if ka == "spmat" and kb == "varvec":
            return _rule_spmatvec(model, ctx, a, b)
        if ka == "tvarvec" and kb == "mat":
            return _LazyRowTimesMatrix(model, a.parent, b)
        if ka == "rowmat":
            x, Q = (a.x, a.Q)
            return _rule_bilinear(model, ctx, x, Q, b)
        if ka in ("tvarvec", "taffvec", "tvec") and kb in VECTOR_KINDS:          # x' * y -> dot (src/functions.jl:824-829)
            return _rule_dot(model, ctx, a.parent, b)
        if ka == "num" and kb in VECTOR_KINDS:
            return _rule_scale(model, ctx, a, b)
        if kb == "num" and ka in VECTOR_KINDS:
            return _rule_scale(model, ctx, b, a)
        if (ka in SCALAR_FUNC_KINDS | {"num"}) and (kb in SCALAR_FUNC_KINDS | {"num"}):
            return _rule_mul_scalar(model, ctx, a, b)
        raise ArgumentError("%s * %s is not supported on the device" % (ka, kb))
    if f in ("+", "-"):
        a, b = args
        ka, kb = kinds
        sign = +1 if f == "+" else -1
        if ka in VECTOR_KINDS and kb in VECTOR_KINDS:
            return _rule_vec_addsub(model, ctx, a, b, sign)
        if (ka in SCALAR_FUNC_KINDS | {"num"}) and (kb in SCALAR_FUNC_KINDS | {"num"}):
            return _rule_scalar_addsub(model, ctx, a, b, sign)
        raise ArgumentError("%s %s %s is not supported on the device" % (ka, f, kb))
    if f == "dot":
        a, b = args
        if kinds[0] in VECTOR_KINDS and kinds[1] in VECTOR_KINDS:
            return _rule_dot(model, ctx, a, b)
        return _rule_mul_scalar(model, ctx, a, b)                                # dot of scalars = x * y (src/functions.jl:636-638)
    if f == "adjoint":
        (a,) = args
        if kinds[0] == "mat":
            return _rule_adjoint_matrix(model, ctx, a)
        if kinds[0] in VECTOR_KINDS:
            return Transpose(a)
        return a                                                                  # scalars are their own adjoints (:645)
    if f == "canonicalize":
        return _rule_canonicalize(model, ctx, args[0])
    if f == "vcat":
        return _rule_vcat(model, ctx, *args)
    if f == "vect":
        return _rule_vect(model, ctx, *args)
    if f in ("convert", "identity"):
        return args[-1]                                                           # rule :280-282: copyto! of references = alias
    raise ArgumentError("Unhandled expression head: %s" % (f,))                   # src/lazyexpression.jl:167-177


# ---- functional syntax (the reference's @expression on explicit calls)
def dot(x, y):
    return lazy("dot", x, y)


def transpose(x):
    return lazy("adjoint", x)


adjoint = transpose


def vcat(*vs):
    return lazy("vcat", *vs)


def vect(x):
    return lazy("vect", x)


def bilinear(x, Q, y):
    """transpose(x) * Q * y"""
    return lazy("*", Transpose(x) if not isinstance(x, Transpose) else x, Q, y)


def expression(thunk):
    """@expression <code>: in Python the operators of Parameter / LazyExpression already build the lazy DAG, so the
    'macro' simply evaluates the thunk (or returns its argument)."""
    return thunk() if callable(thunk) and not _is_lazy(thunk) else thunk


def prune_zero(expr, atol=0.0):
    """prune_zero(expr(); atol) for a device expression (src/functions.jl:294-297, 409-413): the expression is re-evaluated, its term
    lists are compacted ON THE DEVICE (pmt_prune_zero_f64) and only the surviving terms are fetched.  The result is a native host
    function — the number of terms is data dependent, so this is not a node of the solve path (nor is it in the reference)."""
    if not isinstance(expr, DeviceNode):
        from .functions import prune_zero as host_prune
        return host_prune(expr() if _is_lazy(expr) else expr, atol)
    ctx = expr.model.device()
    out = expr.out
    if not isinstance(out, (DAff, DQuad)):
        raise ArgumentError("prune_zero needs an AffineFunction or QuadraticFunction expression")
    if isinstance(out, DQuad):
        out.materialize()
    evaluate(ctx, [expr])

    bufs = expr.__dict__.setdefault("_prune_bufs", {})          # plan memory is never freed: allocate once per node

    def compact(ptr, n, dtype, tol):
        nbytes = dtype.itemsize
        if ptr not in bufs:
            ws_bytes = int(ctx.lib.pmt_prune_zero_workspace_bytes(n, nbytes))
            bufs[ptr] = (ctx.alloc(nbytes * max(n, 1)), ctx.alloc(8), ctx.alloc(ws_bytes), ws_bytes)
        dst, cnt, ws, ws_bytes = bufs[ptr]
        ctx.call("pmt_prune_zero_f64", P(ptr), n, nbytes, float(tol), P(dst), P(cnt), P(ws), ws_bytes)
        k = np.zeros(1, dtype=np.int64)
        ctx.fetch(k, cnt, 8); ctx.synchronize()
        t = fetch_terms(ctx, dst, int(k[0]), dtype); ctx.synchronize()
        return t
    c = fetch_f64(ctx, out.const, 1)
    if isinstance(out, DAff):
        t = compact(out.terms, out.nterms, LT, atol)
        ctx.synchronize()
        return AffineFunction.from_arrays(t, c[0])
    q = compact(out.quad, out.nq, QT, atol)
    l = compact(out.lin, out.nl, LT, 0.0)                 # the reference prunes the affine part with the DEFAULT atol (:410)
    ctx.synchronize()
    return QuadraticFunction.from_arrays(q, l, c[0])


def getindex(x, *idx):
    """@expression p[i, j] / p[:, j]"""
    return lazy(lambda v: v[idx if len(idx) > 1 else idx[0]], x)


def getproperty(x, name):
    """@expression p.name (src/lazyexpression.jl:144, 300-302)"""
    return lazy("getproperty", x, name)


def wrap(expr):
    """wrap(wrap(e)) === wrap(e) (test/lazyexpression.jl:106-107): nodes are already type-erased."""
    return expr
