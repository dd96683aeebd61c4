"""MathOptInterface hand-off (src/moi_interop.jl): function/set records, Objective / Constraint / Constraints and the
native -> MOI copies `update!(moi_f, f, varmap)` (:35-81).

MOI function buffers are numpy structured arrays with exactly the Julia isbits layouts, so a
Julia host can pass `pointer(moi_f.terms)` as the destination of pmt_plan_fetch.  For non-constant records the copy
itself runs on the device (the pack kernels write MOI terms, through `varmap`, into device twins of these buffers);
constant records are converted once on the host at setup, as in the reference (`isconstant`, :123,132,153,169).
"""
import ctypes as C

import numpy as np

from ._lib import LT, QT, VAT, ArgumentError, DimensionMismatch, ErrorException
from .device import DAff, DAffVec, DDenseAff, DQuad, DSparseAff, DVarsAff, P
from .functions import AffineFunction, LinearTerm, QuadraticFunction, QuadraticTerm, Variable, _isnum
from .lazyexpression import DeviceNode, kind_of

MIN_SENSE, MAX_SENSE = "MIN_SENSE", "MAX_SENSE"


# ---- sets (MOI.AbstractSet)
class _Set:
    def __init__(self, value=None):
        self.value = value

    def __repr__(self):
        return "%s(%r)" % (type(self).__name__, self.value)

    def __eq__(self, o):
        return type(o) is type(self) and o.value == self.value

    def __hash__(self):
        return hash((type(self).__name__, self.value))


class GreaterThan(_Set): pass
class LessThan(_Set): pass
class EqualTo(_Set): pass
class Nonnegatives(_Set): pass
class Nonpositives(_Set): pass
class Zeros(_Set): pass
class Integer(_Set): pass
class ZeroOne(_Set): pass


# ---- functions (MOI.AbstractFunction)
def _zeros(n, dtype):
    return np.zeros(n, dtype=dtype)


# `alloc(n, dtype)` lets the device path place the term buffers in page-locked host memory (DeviceContext.pinned_array)
class ScalarAffineFunction:
    def __init__(self, nterms=0, alloc=_zeros):
        self.terms = alloc(nterms, LT)
        self.constant = 0.0


class ScalarQuadraticFunction:
    def __init__(self, naff=0, nquad=0, alloc=_zeros):
        self.affine_terms = alloc(naff, LT)
        self.quadratic_terms = alloc(nquad, QT)
        self.constant = 0.0


class VectorAffineFunction:
    def __init__(self, nterms=0, nrows=0, alloc=_zeros):
        self._terms = alloc(nterms, VAT)
        self.constants = alloc(nrows, np.float64)
        self.coefficients_unavailable = None      # a reason: the coefficients are NOT on the host (handoff="host_csc", dense block)

    @property
    def terms(self):
        """MOI.VectorAffineTerm[] (src/moi_interop.jl:64-81).  With handoff="host_csc" the terms of a dense block A*x (+|-) b are never packed —
        the solver's A is delivered as CSC values straight out of the Parameter's buffer — and only their STRUCTURE (`structure`) is host
        data: reading the coefficients is an error, not a NaN."""
        if self.coefficients_unavailable:
            raise ErrorException(self.coefficients_unavailable)
        return self._terms

    @terms.setter
    def terms(self, value):
        self._terms = value

    @property
    def structure(self):
        """(output rows, optimizer variables) of the terms — always host data"""
        return self._terms["out"], self._terms["var"]


class SingleVariable:
    def __init__(self, variable):
        self.variable = variable


# ---- host restatement of update!(moi_f, f, varmap) for CONSTANT records (setup time)
def _vm(varmap, idx):
    return idx if varmap is None else int(varmap[idx - 1])


def update_scalar_affine(moi_f, f, varmap=None):                      # src/moi_interop.jl:35-43
    moi_f.constant = float(f.constant)
    moi_f.terms = np.zeros(len(f.linear), dtype=LT)
    for i, t in enumerate(f.linear):
        moi_f.terms[i] = (t.coeff, _vm(varmap, t.var.index))
    return moi_f


def update_scalar_quadratic(moi_f, f, varmap=None):                   # src/moi_interop.jl:45-62
    update_scalar_affine_part = ScalarAffineFunction()
    update_scalar_affine(update_scalar_affine_part, f.affine, varmap)
    moi_f.constant = update_scalar_affine_part.constant
    moi_f.affine_terms = update_scalar_affine_part.terms
    moi_f.quadratic_terms = np.zeros(len(f.quadratic), dtype=QT)
    for i, t in enumerate(f.quadratic):
        coeff = 2 * t.coeff if t.rowvar == t.colvar else t.coeff        # :58
        moi_f.quadratic_terms[i] = (coeff, _vm(varmap, t.rowvar.index), _vm(varmap, t.colvar.index))
    return moi_f


def update_vector_affine(moi_f, fs, varmap=None):                     # src/moi_interop.jl:64-81
    n = sum(len(f.linear) for f in fs)
    moi_f.constants = np.zeros(len(fs), dtype=np.float64)
    moi_f.terms = np.zeros(n, dtype=VAT)
    i = 0
    for row, f in enumerate(fs):
        for t in f.linear:
            moi_f.terms[i] = (row + 1, t.coeff, _vm(varmap, t.var.index))
            i += 1
        moi_f.constants[row] = f.constant
    return moi_f


def canonical_function_kind(kind):                                     # src/moi_interop.jl:96-101
    if kind in ("var", "lt", "aff", "num"):
        return "aff"
    if kind in ("varvec", "affvec"):
        return "affvec"
    if kind in ("qt", "quad"):
        return "quad"
    raise ArgumentError("no canonical function type for %s" % kind)


def _to_native(kind, val):
    if kind == "aff":
        return AffineFunction.of(float(val) if _isnum(val) else val)
    if kind == "quad":
        return QuadraticFunction.of(val)
    return [AffineFunction.of(float(v) if _isnum(v) else v) for v in val]


def _gram_rows(gram):
    """Row count handed to the Gram kernel: the zero-padded count (device.row_padded) when both the matrix and the vector carry the
    zero padding rows — whole 16-row stages keep the contraction on its branch-free path; zero rows change nothing."""
    from .device import DMat, DVec, row_padded
    rows = gram.mat.rows
    padded = row_padded(rows)
    if padded == rows or not isinstance(gram.mat, DMat) or gram.mat.lda < padded:
        return rows
    if gram.vec is not None and not (isinstance(gram.vec, DVec) and getattr(gram.vec, "padded", 0) >= padded):
        return rows
    return padded


class _Record:
    """Common part of Objective and Constraint (src/moi_interop.jl:113-129, 141-166)."""

    def _setup(self, model, expr):
        self.model = model
        self.expr = expr
        self.isconstant = not isinstance(expr, DeviceNode)                # "it's just a value; not a LazyExpression" (:123)
        self.dev = None                                                   # device twin of the MOI buffers (set by compile)
        if self.isconstant:
            self.kind = canonical_function_kind(kind_of(expr))
            native = _to_native(self.kind, expr)
            if self.kind == "aff":
                self.f = update_scalar_affine(ScalarAffineFunction(), native)
            elif self.kind == "quad":
                self.f = update_scalar_quadratic(ScalarQuadraticFunction(), native)
            else:
                self.f = update_vector_affine(VectorAffineFunction(), native)
            self.nrows = len(native) if self.kind == "affvec" else 1
        else:
            self.kind = canonical_function_kind(expr.out.kind)
            self.f = None                                                 # sized by compile()
            self.nrows = expr.out.rows if self.kind == "affvec" else 1

    # ---- device side of update!(moi_f, expr(), varmap)
    def compile(self, ctx, varmap_buf, quadratic_mode, handoff_varmap=None):
        """Allocate the MOI buffers (host + device twin) and return the emitter of the MOI copy.  `handoff_varmap` (host array,
        device hand-off only): the final model_var_to_optimizer; a Gram objective whose variables stay in increasing order under it
        writes the solver's CSC values of P directly from the contraction's epilogue and no quadratic term structs at all."""
        out = self.expr.out
        self.varmap_hooks = []                                            # called with the new host varmap whenever it changes
        self.side_lane_ok = False
        self.on_side_lane = False
        # A SMALL model (Model.initialize: launch-bound on the device) has no device twin of its MOI buffers: the kernels store straight into
        # the page-locked host arrays of the function object (a few KB over PCIe from inside the one launch), and update! ends with ONE stream
        # synchronisation instead of a D2H copy per buffer (~10 us each: five of them were half of solve! at n = 100)
        zero_copy = bool(getattr(self.model, "_small", False))

        def twin(host, nbytes):
            if zero_copy and nbytes > 0 and host.nbytes >= nbytes:
                return host.ctypes.data
            return ctx.alloc(max(nbytes, 16))
        self._cbuf = ctx.pinned_array(1, np.float64)                      # the scalar functions' constant lands here
        if self.kind == "aff":
            n = out.nterms
            self.f = ScalarAffineFunction(n, alloc=ctx.pinned_array)
            dev_terms = twin(self.f.terms, 16 * n)
            # a small model: the constant (a word the expression's node left in HBM) is copied into its page-locked word by one more entry of
            # the tape — a node of the one launch — instead of a D2H copy behind every replay (a hipMemcpyAsync of 8 bytes is ~5 us of host time)
            dconst = self._cbuf.ctypes.data if zero_copy else out.const
            self.dev = {"terms": dev_terms, "const": dconst}

            def emit(c):
                c.call("pmt_pack_scalar_affine_f64", P(out.terms), n, P(varmap_buf), P(dev_terms))
                if zero_copy:
                    c.call("pmt_copy_bytes", P(dconst), P(out.const), 8)
            return emit
        if self.kind == "quad":
            gram = getattr(self.expr, "gram_candidate", None)
            literal_terms = out.nq
            use_gram = gram is not None and gram.xvars.strictly_increasing() and (
                quadratic_mode == "canonical" or (quadratic_mode == "auto" and literal_terms > (1 << 24)))
            if use_gram and handoff_varmap is not None and np.all(np.diff(handoff_varmap[gram.xvars.vars - 1]) > 0):
                n = gram.mat.cols
                self.f = ScalarQuadraticFunction(n, 0, alloc=ctx.pinned_array)
                dp, dl, dc = ctx.alloc(8 * max(n * (n + 1) // 2, 1)), ctx.alloc(16 * max(n, 1)), ctx.alloc(8)
                ws = ctx.alloc(max(16, int(ctx.lib.pmt_quad_gram_workspace_bytes(_gram_rows(gram), n))))
                self.dev = {"P_values": dp, "P_vars": handoff_varmap[gram.xvars.vars - 1], "lin": dl, "const": dc}
                self.mode = "canonical-csc"
                vec = gram.vec.buf if gram.vec is not None else None
                alpha = -1.0 if self.model.sense == "Maximize" else 1.0
                host_P = None
                if getattr(self.model, "handoff", "") == "host_csc" and getattr(self.model, "_overlap_fetch", False):
                    # host solver hand-off: the contraction finishes P column band by column band and every finished group of bands leaves
                    # for this page-locked array while the rest is still being computed (pmt_quad_gram_csc_deliver_f64)
                    host_P = ctx.pinned_array(max(n * (n + 1) // 2, 1), np.float64)
                    self.dev["P_host"] = host_P

                def emit(c):
                    if host_P is not None:
                        c.call("pmt_quad_gram_csc_deliver_f64", P(gram.mat.buf), gram.mat.lda, _gram_rows(gram), n, P(gram.xvars.buf), P(vec),
                               gram.sign if vec else 0, P(varmap_buf), alpha, P(dp), host_P.ctypes.data_as(C.c_void_p), 0, P(dl), P(dc), P(ws))
                    else:
                        c.call("pmt_quad_gram_csc_f64", P(gram.mat.buf), gram.mat.lda, _gram_rows(gram), n, P(gram.xvars.buf), P(vec),
                               gram.sign if vec else 0, P(varmap_buf), alpha, P(dp), None, P(dl), P(dc), P(ws))
                return emit
            if use_gram:
                n = gram.mat.cols
                nq = n * (n + 1) // 2
                self.f = ScalarQuadraticFunction(n, nq, alloc=ctx.pinned_array)
                dq, dl, dc = twin(self.f.quadratic_terms, 24 * nq), twin(self.f.affine_terms, 16 * n), twin(self._cbuf, 8)
                ws = ctx.alloc(max(16, int(ctx.lib.pmt_quad_gram_workspace_bytes(_gram_rows(gram), n))))
                self.dev = {"quad": dq, "lin": dl, "const": dc}
                self.mode = "canonical"
                vec = gram.vec.buf if gram.vec is not None else None

                deliver = bool(getattr(self.model, "_overlap_moi", False)) and nq > 0
                self._quad_delivered = deliver

                def emit(c):
                    if deliver:
                        # the quadratic terms leave for f.quadratic_terms (page-locked) row band by row band while the contraction runs
                        c.call("pmt_quad_gram_deliver_f64", P(gram.mat.buf), gram.mat.lda, _gram_rows(gram), n, P(gram.xvars.buf), P(vec),
                               gram.sign if vec else 0, 1, P(varmap_buf), P(dq), self.f.quadratic_terms.ctypes.data_as(C.c_void_p), 0, P(dl), P(dc), P(ws))
                    else:
                        c.call("pmt_quad_gram_f64", P(gram.mat.buf), gram.mat.lda, _gram_rows(gram), n, P(gram.xvars.buf), P(vec), gram.sign if vec else 0,
                               1, P(varmap_buf), P(dq), P(dl), P(dc), P(ws))
                return emit
            self.mode = "literal"
            out.materialize()
            self.f = ScalarQuadraticFunction(out.nl, out.nq, alloc=ctx.pinned_array)
            dq, dl = twin(self.f.quadratic_terms, 24 * out.nq), twin(self.f.affine_terms, 16 * out.nl)
            dconst = self._cbuf.ctypes.data if zero_copy else out.const          # (as for the affine function above)
            self.dev = {"quad": dq, "lin": dl, "const": dconst}

            def emit(c):
                c.call("pmt_pack_scalar_quadratic_f64", P(out.quad), out.nq, P(varmap_buf), P(dq))
                c.call("pmt_pack_scalar_affine_f64", P(out.lin), out.nl, P(varmap_buf), P(dl))
                if zero_copy:
                    c.call("pmt_copy_bytes", P(dconst), P(out.const), 8)
            return emit
        # Vector{AffineFunction}
        # handoff="host_csc": the deliverable is the solver's CSC arrays on the host and the index map is fixed (Model.initialize), so the MOI
        # terms of a dense block A*x (+|-) b or of x (+|-) v would be an intermediate nobody reads: A's CSC values are the Parameter matrix
        # column by column (they leave straight out of its buffer, handoff.py) and the coefficients of x (+|-) v are the constant 1.0.
        # Only the constants 0 (+|-) b are rebuilt per re-evaluation; the term STRUCTURE (rows, optimizer variables) is static host data.
        if handoff_varmap is not None and getattr(self.model, "handoff", None) == "host_csc" and isinstance(out, (DDenseAff, DVarsAff)) and not out.need_terms:
            xv = np.asarray(handoff_varmap, dtype=np.int64)[out.xvars.vars - 1]
            dense = isinstance(out, DDenseAff)
            if not dense or (len(xv) and np.all(np.diff(xv) > 0)):            # (a dense block whose columns are permuted or repeated keeps its terms)
                self.f = VectorAffineFunction(out.nterms, out.rows)
                t = self.f._terms
                if dense:
                    t["out"], t["var"], t["coeff"] = np.repeat(np.arange(1, out.rows + 1), out.mat.cols), np.tile(xv, out.rows), np.nan
                    self.f.coefficients_unavailable = ("handoff=\"host_csc\": the MOI terms of a dense constraint block are not packed (its coefficients "
                                                       "are the Parameter matrix, delivered as the CSC values of model.device_qp.host); use .structure for "
                                                       "rows / variables, or handoff=\"moi\" for the reference's MOI functions")
                else:
                    t["out"], t["var"], t["coeff"] = np.arange(1, out.rows + 1), xv, 1.0
                dc = ctx.alloc(8 * max(out.rows, 1))
                ctx.zero(dc, 8 * max(out.rows, 1))
                self.dev = {"consts": dc}
                self.side_lane_ok = True
                self.terms_static = True

                def emit(c):
                    if out.vec is not None and out.rows:
                        c.call("pmt_consts_f64", P(out.vec.buf), out.rows, out.sign, P(dc))
                return emit
        self.f = VectorAffineFunction(out.nterms, out.rows, alloc=ctx.pinned_array)
        dt = twin(self.f._terms, 24 * out.nterms)
        if isinstance(out, DDenseAff) and not out.need_terms:
            dc = twin(self.f.constants, 8 * out.rows)
            self.dev = {"terms": dt, "consts": dc}
            self.side_lane_ok = True          # reads Parameter values only, writes its own MOI buffers (Model.initialize: side lane)
            vec = out.vec.buf if out.vec is not None else None

            def emit(c):
                # on the side lane (Model.initialize sets on_side_lane before recording): the low-footprint kernel that is co-resident with
                # the contraction it runs beside
                c.call("pmt_affine_pack_vector_background_f64" if self.on_side_lane else "pmt_affine_pack_vector_f64", P(out.mat.buf), out.mat.lda,
                       out.mat.rows, out.mat.cols, P(out.xvars.buf), P(vec), out.sign if vec else 0, P(varmap_buf), 0, P(dt), P(dc))
            return emit
        if isinstance(out, DVarsAff) and not out.need_terms:
            dc = twin(self.f.constants, 8 * out.rows)
            self.dev = {"terms": dt, "consts": dc}
            self.side_lane_ok = True          # reads Parameter values only, writes its own MOI buffers (Model.initialize: side lane)

            def emit(c):
                c.call("pmt_vars_addsub_f64", P(out.xvars.buf), out.rows, P(out.vec.buf), out.sign, P(varmap_buf), 0, None, P(dt), P(dc))
            return emit
        if isinstance(out, DSparseAff) and not out.need_terms:
            dc = twin(self.f.constants, 8 * out.rows)
            self.dev = {"terms": dt, "consts": dc}
            self.side_lane_ok = True          # reads Parameter values only, writes its own MOI buffers (Model.initialize: side lane)
            sp = out.spmat
            # varmap folded into the static variable words: it changes with the optimizer's index map (mapindices!, src/model.jl:100-107),
            # not per re-evaluation, so the kernel reads varmap[x[col]] instead of gathering it for every term.  Block form: one word per
            # COLUMN (staged in LDS per column band); slab form: one per term
            if sp.block_cw:
                colvar = ctx.alloc(8 * max(sp.cols, 1))

                def refresh(varmap_host):
                    v = out.xvars.vars if varmap_host is None else np.asarray(varmap_host, dtype=np.int64)[out.xvars.vars - 1]
                    ctx.upload(colvar, np.ascontiguousarray(v, dtype=np.int64))
                self.varmap_hooks.append(refresh)
                refresh(handoff_varmap)

                def emit(c):
                    # terms and constants (0 (+|-) d) in one launch
                    c.call("pmt_sparse_pack_vector_blocks_f64", P(sp.buf), P(sp.block_desc_buf), P(sp.block_idx_buf), P(sp.block_band_buf), P(colvar),
                           sp.rows, sp.cols, sp.nnz, sp.block_cw, None, 0, P(out.vec.buf) if out.vec is not None else None,
                           out.sign if out.vec is not None else 0, P(dt), P(dc))
                return emit
            mapped = ctx.alloc((4 if sp.narrow else 8) * max(sp.nnz, 1))

            def refresh(varmap_host):
                if sp.nnz:
                    v = out.term_var if varmap_host is None else np.asarray(varmap_host, dtype=np.int64)[out.term_var - 1]
                    if sp.narrow and (v.max() >= 2 ** 32 or v.min() < 0):
                        raise DimensionMismatch("sparse constraint: optimizer variable indices of 2^32 or more with a 32-bit pattern")
                    ctx.upload(mapped, v.astype(np.uint32) if sp.narrow else np.ascontiguousarray(v))
            self.varmap_hooks.append(refresh)
            refresh(handoff_varmap)

            def emit(c):
                c.call("pmt_sparse_pack_vector_slabs_u32_f64" if sp.narrow else "pmt_sparse_pack_vector_slabs_f64", P(sp.buf), P(sp.perm_buf), P(mapped),
                       P(sp.slab_ptr_buf), sp.rows, sp.nslab, None, 0, P(dt))
                if out.vec is not None:
                    c.call("pmt_consts_f64", P(out.vec.buf), out.rows, out.sign, P(dc))
            return emit
        m = out.materialized()
        self.dev = {"terms": dt, "consts": m.consts}

        def emit(c):
            c.call("pmt_pack_vector_affine_f64", P(m.terms), P(m.row_ptr_buf), m.rows, m.row_len, P(varmap_buf), 0, P(dt))
        return emit

    def record_fetch(self, ctx):
        """while recording (Model._overlap_moi): the same copies as fetch(), as tape entries behind this record's launches on their lane —
        they leave while the rest of the tape is still running (pmt_plan_record_fetch); the objective's quadratic terms are delivered by
        the contraction itself when it is the canonical node"""
        f, d = self.f, self.dev
        if self.isconstant or d is None:
            return
        if self.kind in ("aff", "quad"):
            self._c = ctx.pinned_array(1, np.float64)
        if self.kind == "aff":
            ctx.record_fetch(f.terms, d["terms"], f.terms.nbytes)
            ctx.record_fetch(self._c, d["const"], 8)
        elif self.kind == "quad":
            if "quad" in d and not getattr(self, "_quad_delivered", False):
                ctx.record_fetch(f.quadratic_terms, d["quad"], f.quadratic_terms.nbytes)
            ctx.record_fetch(f.affine_terms, d["lin"], f.affine_terms.nbytes)
            ctx.record_fetch(self._c, d["const"], 8)
        else:
            ctx.record_fetch(f.terms, d["terms"], f.terms.nbytes)
            ctx.record_fetch(f.constants, d["consts"], f.constants.nbytes)
        self._fetch_recorded = True

    def fetch(self, ctx):
        """D2H of the MOI buffers into the host function object (asynchronous; caller synchronises).  A buffer whose device twin IS the
        host array (a small model, compile) needs no copy."""
        if getattr(self, "_fetch_recorded", False):
            return
        f, d = self.f, self.dev

        def get(host, key):
            if key in d and d[key] != host.ctypes.data:
                ctx.fetch(host, d[key], host.nbytes)
        if self.kind == "aff":
            get(f.terms, "terms")
            self._c = self._cbuf; get(self._c, "const")
        elif self.kind == "quad":
            get(f.quadratic_terms, "quad")                                # (absent when P's CSC values are written directly)
            get(f.affine_terms, "lin")
            self._c = self._cbuf; get(self._c, "const")
        else:
            get(f._terms, "terms")                                        # (absent for a host_csc record whose terms are static, compile)
            get(f.constants, "consts")

    def fetch_list(self):
        """(host array, key of self.dev) pairs fetch() copies — for a host that registers them once (Model._create_model_run)"""
        f = self.f
        if self.kind == "aff":
            self._c = self._cbuf
            return [(f.terms, "terms"), (self._c, "const")]
        if self.kind == "quad":
            self._c = self._cbuf
            return [(f.quadratic_terms, "quad"), (f.affine_terms, "lin"), (self._c, "const")]
        return [(f._terms, "terms"), (f.constants, "consts")]

    def finish_fetch(self):
        if self.kind in ("aff", "quad"):
            self.f.constant = float(self._c[0])


class Objective(_Record):                                                 # src/moi_interop.jl:113-137
    def __init__(self, model, expr):
        self._setup(model, expr)
        if self.kind not in ("aff", "quad"):
            raise ArgumentError("the objective must be a scalar affine or quadratic function")


class Constraint(_Record):                                                # src/moi_interop.jl:141-175
    def __init__(self, model, expr, set_, function=None):
        self.set = set_
        self.modelindex = None
        self.optimizerindex = None
        if function is not None:                                          # SingleVariable-in-Integer/ZeroOne (:161-165)
            self.model, self.expr, self.isconstant, self.kind, self.f, self.nrows, self.dev = model, None, True, "single", function, 1, None
            return
        self._setup(model, expr)

    @property
    def spec(self):
        fname = {"aff": "scalaraffinefunction", "quad": "scalarquadraticfunction", "affvec": "vectoraffinefunction", "single": "singlevariable"}[self.kind]
        return fname + "_in_" + type(self.set).__name__.lower()


# the 14 typed constraint vectors in the reference's fixed order (src/moi_interop.jl:180-193)
CONSTRAINT_ORDER = [
    "scalaraffinefunction_in_greaterthan", "scalaraffinefunction_in_lessthan", "scalaraffinefunction_in_equalto",
    "vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives", "vectoraffinefunction_in_zeros",
    "scalarquadraticfunction_in_greaterthan", "scalarquadraticfunction_in_lessthan", "scalarquadraticfunction_in_equalto",
    "vectorquadraticfunction_in_nonnegatives", "vectorquadraticfunction_in_nonpositives", "vectorquadraticfunction_in_zeros",
    "singlevariable_in_integer", "singlevariable_in_zeroone",
]


class Constraints:                                                        # src/moi_interop.jl:195-262
    def __init__(self):
        self.by_spec = {name: [] for name in CONSTRAINT_ORDER}

    def push(self, c):
        if c.spec not in self.by_spec:
            raise ArgumentError("unsupported constraint type %s" % c.spec)
        self.by_spec[c.spec].append(c)

    def __iter__(self):                                                    # update! order (:236-247)
        for name in CONSTRAINT_ORDER:
            yield from self.by_spec[name]

    def __len__(self):
        return sum(len(v) for v in self.by_spec.values())
