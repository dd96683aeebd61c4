"""Device-resident solver hand-off.

In the reference the data leaves Parametron at `MOI.set(optimizer, ObjectiveFunction / ConstraintFunction, f)`
(src/moi_interop.jl:134,171); MathOptInterface 0.8 and the solver wrapper (OSQP.jl, ...) then rebuild the solver's
matrices from the term lists on the host.  `DeviceQP` does that step in HBM, so a GPU QP solver — or a host solver that
wants 67 MB of CSC values instead of 252 MB of term structs — reads

    minimize 1/2 x'Px + q'x + r   subject to   l <= Ax <= u

with P upper triangular and P, A in CSC (0-based Int64 indices, OSQP's C layout).  The CSC structure depends only on the
static indices and is computed once (pmt_csc_order, host); `refresh()` rebuilds the values from the model's device MOI
buffers (pmt_csc_values_f64, pmt_qp_bounds_f64) after `update!(model)`.  Constraint rows are stacked in the reference's
update order (src/moi_interop.jl:236-247).  Maximize is handed over as minimize of the negated objective.
"""
import ctypes as C

import numpy as np

from . import _lib, moi
from ._lib import ArgumentError, ErrorException
from .device import DDenseAff, DSparseAff, DVarsAff, P

_SET_KIND = {moi.EqualTo: 0, moi.Zeros: 0, moi.GreaterThan: 1, moi.Nonnegatives: 1, moi.LessThan: 2, moi.Nonpositives: 2}
DEFAULT_INFTY = 1e20


def _csc_order(rows, cols, nrows, ncols, upper):
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    n = len(rows)
    perm, seg = np.zeros(max(n, 1), dtype=np.int64), np.zeros(n + 1, dtype=np.int64)
    col_ptr, row_idx = np.zeros(ncols + 1, dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
    nnz = C.c_int64(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.call("pmt_csc_order", n, vp(rows), vp(cols), int(nrows), int(ncols), int(bool(upper)), vp(perm), vp(seg), vp(col_ptr), vp(row_idx),
              C.byref(nnz))
    k = nnz.value
    return perm[:n], seg[:k + 1], col_ptr, row_idx[:k]


class _Block:
    """One source of matrix entries: a device term buffer (or host values for constant functions) and where its runs go."""

    def __init__(self, rows, cols, coeff_ptr=None, stride=0, host_coeff=None, source_perm=None, dense=None, param=None):
        self.rows, self.cols, self.coeff_ptr, self.stride, self.host_coeff = rows, cols, coeff_ptr, stride, host_coeff
        self.source_perm = source_perm            # entry i's coefficient sits at coeff_ptr + source_perm[i] * stride (None: i * stride)
        self.dense = dense                        # DMat of a dense block A*x (+|-) b whose terms are row-major (row*cols + col): its values ARE the matrix
        self.param = param                        # ... and the Parameter it mirrors


class _Rect:
    """A dense block inside a solver matrix's CSC values: column j of the Parameter matrix (rows doubles at mat + j*lda) sits at
    values[first + j*pitch ...] — a pitched copy, no term is read."""

    def __init__(self, first, pitch, mat, offsets=None, param=None):
        self.first, self.pitch, self.mat = int(first), int(pitch), mat
        self.offsets = offsets              # per-column positions when they are not a constant pitch apart (then pitch is 0)
        self.param = param

    def host_resident(self):
        """the block's Parameter is updated by the HOST (val= buffer or callback, src/parameter.jl:88,101-102): its values are already
        there and need not come back from the device"""
        return self.param is not None and not getattr(self.param, "device_resident", False) and self.offsets is None


class CSC:
    def __init__(self, nrows, ncols, col_ptr, row_idx, values_ptr, rects=(), only_rects=False):
        self.shape = (nrows, ncols)
        self.col_ptr, self.row_idx, self.values_ptr = col_ptr, row_idx, values_ptr
        self.nnz = len(row_idx)
        self.rects = list(rects)            # dense blocks that are pitched copies of Parameter matrices
        self.only_rects = only_rects        # every value that changes per re-evaluation belongs to one of them (the rest is static)


def _parameter_of(expr, dmat):
    """the Parameter whose device mirror `dmat` is, among the arguments of the record's expression"""
    from .lazyexpression import schedule
    from .parameter import Parameter
    for x in schedule([expr]):
        if isinstance(x, Parameter) and getattr(x, "_dev", None) is dmat:
            return x
    return None


class _Transfer:
    """one device -> host copy of a re-evaluation: `nbytes` linear, or `height` rows of `nbytes` with the given pitches"""

    def __init__(self, host_ptr, dev_ptr, nbytes, dst_pitch=0, src_pitch=0, height=0, param_id=None):
        self.host_ptr, self.dev_ptr, self.nbytes, self.dst_pitch, self.src_pitch, self.height = int(host_ptr), int(dev_ptr), int(nbytes), dst_pitch, src_pitch, height
        self.param_id = param_id                  # id() of the Parameter a pitched transfer reads, if any

    def record(self, ctx):
        if self.height:
            ctx.record_fetch_2d(self.host_ptr, self.dst_pitch, self.dev_ptr, self.src_pitch, self.nbytes, self.height)
        else:
            ctx.record_fetch_ptr(self.host_ptr, self.dev_ptr, self.nbytes)

    def fetch(self, ctx):
        if self.height:
            ctx.fetch_2d_ptr(self.host_ptr, self.dst_pitch, self.dev_ptr, self.src_pitch, self.nbytes, self.height)
        else:
            ctx.fetch_ptr(self.host_ptr, self.dev_ptr, self.nbytes)


class HostQP:
    """What a HOST solver takes per solve (OSQP: update(Px=, Ax=, q=, l=, u=)) — numpy arrays over page-locked memory that the
    re-evaluation refills in place; the structure arrays (Pi, Pp, Ai, Ap) are computed once.  `wait()` blocks until the arrays of the
    current re-evaluation have landed (Model.update(synchronize=True) has already done so)."""

    def __init__(self, qp):
        self._qp = qp
        ctx = qp.ctx
        obj_dev = qp.model.objective.dev or {}
        self.Px = obj_dev["P_host"] if "P_host" in obj_dev else ctx.pinned_array(max(qp.P.nnz, 1), np.float64)
        self.P_delivered_by_contraction = "P_host" in obj_dev
        self.Px = self.Px[:qp.P.nnz]
        self.Pi, self.Pp = qp.P.row_idx, qp.P.col_ptr
        self._small = ctx.pinned_array(max(qp._small_len, 1), np.float64)        # q | l | u, one transfer
        n, m = qp.nvars, qp.nrows
        self.q, self.l, self.u = self._small[:n], self._small[n:n + m], self._small[n + m:n + 2 * m]
        self.Ax = ctx.pinned_array(max(qp.A.nnz, 1), np.float64)[:qp.A.nnz]
        self.Ai, self.Ap = qp.A.row_idx, qp.A.col_ptr
        self._r = ctx.pinned_array(1, np.float64)
        self._r[0] = qp._obj_const[1]

    @property
    def r(self):
        return self._qp.sign * float(self._r[0])

    def transfers(self, late=None):
        """the device -> host copies of one re-evaluation; P is absent when the contraction delivers it itself.
        late=False: what is ready early — A's values (dense blocks: pitched copies straight out of their Parameter buffers, ready when the
        re-evaluation starts; otherwise the hand-off's gather output) and q | l | u; late=True: what only the objective's own kernels
        finish — its constant (the node's serial c'c chain) and P when it is not delivered band by band; None: all."""
        qp = self._qp
        if qp.A.only_rects:
            # A's static entries (bounds rows: 1.0) were fetched once at set-up; per solve only the dense blocks travel
            # (blocks of host-updated Parameters do not travel at all: host_copies())
            early = [_Transfer(self.Ax.ctypes.data + 8 * r.first, r.mat.buf, 8 * r.mat.rows, 8 * r.pitch, 8 * r.mat.lda, r.mat.cols,
                               param_id=id(r.param) if r.param is not None else None)
                     for r in qp.A.rects if not r.host_resident()]
        else:
            early = [_Transfer(self.Ax.ctypes.data, qp.A.values_ptr, self.Ax.nbytes)]
        early.append(_Transfer(self._small.ctypes.data, qp._small_ptr, 8 * qp._small_len))
        tail = []
        if qp._obj_const[0]:
            tail.append(_Transfer(self._r.ctypes.data, qp._obj_const[0], 8))
        if not self.P_delivered_by_contraction:
            tail.append(_Transfer(self.Px.ctypes.data, qp.P.values_ptr, self.Px.nbytes))
        t = early + tail if late is None else (tail if late else early)
        return [x for x in t if x.nbytes]

    def host_copies(self):
        """The dense blocks of A whose Parameter the HOST updates: copied on the host, from the Parameter's buffer into the block's row range
        of every column of Ax (pmt_host_copy_2d, a few worker threads), while the device re-evaluates the rest — they do not cross PCIe a
        second time.  Called by Model.update() once the re-evaluation has been enqueued."""
        qp = self._qp
        if not qp.A.only_rects:
            return
        for r in qp.A.rects:
            if not r.host_resident():
                continue
            val = r.param()                               # this solve's value (evaluated when the Parameters were refreshed)
            m, n = r.mat.rows, r.mat.cols
            dst = self.Ax.ctypes.data + 8 * r.first
            if isinstance(val, np.ndarray) and val.dtype == np.float64 and val.shape == (m, n) and (m == 1 or val.strides[0] == 8) and \
                    (n == 1 or val.strides[1] >= 8 * m):
                _lib.call("pmt_host_copy_2d", C.c_void_p(dst), 8 * r.pitch, C.c_void_p(val.ctypes.data), int(val.strides[1]) if n > 1 else 8 * m, 8 * m, n, 0)
            else:                                         # row-major / other layouts: numpy's strided copy (slow; column-major buffers avoid it)
                view = np.lib.stride_tricks.as_strided(self.Ax[r.first:], shape=(n, m), strides=(8 * r.pitch, 8))
                view[...] = np.asarray(val, dtype=np.float64).T

    def bytes_over_pcie(self):
        """what crosses PCIe per solve: everything except the blocks the host copies itself"""
        qp = self._qp
        kept = sum(8 * r.mat.rows * r.mat.cols for r in qp.A.rects if r.host_resident()) if qp.A.only_rects else 0
        return self.nbytes() - kept

    def nbytes(self):
        return self.Px.nbytes + self.q.nbytes + self.Ax.nbytes + self.l.nbytes + self.u.nbytes + 8

    def wait(self):
        self._qp.ctx.fetch_synchronize()
        self._qp.ctx.synchronize()

    def as_dict(self):
        return {"P": (self.Px, self.Pi, self.Pp), "q": self.q, "r": self.r, "A": (self.Ax, self.Ai, self.Ap), "l": self.l, "u": self.u}


class DeviceQP:
    def __init__(self, model, infty=DEFAULT_INFTY, in_tape=False, host=None):
        """in_tape: False (the hand-off kernels are launched behind the tape by refresh()), "side" (recorded as side-lane entries) or
        "main" (recorded at the end of the tape).  host: None, "overlap" (the solver's arrays leave for page-locked host memory as
        RECORDED fetches, each as soon as its producer is done) or "serial" (fetched behind the whole re-evaluation)."""
        if in_tape is True:
            in_tape = "side"
        if host not in (None, "overlap", "serial"):
            raise ArgumentError("DeviceQP: host must be None, 'overlap' or 'serial'")
        if not model.initialized:
            raise ErrorException("DeviceQP needs an initialized model (initialize!(model) / solve!(model) first)")
        self.model, self.infty = model, float(infty)
        ctx = self.ctx = model.device()
        if model._records:
            model._run_tape()                     # host MOI buffers now carry the optimizer's indices (after mapindices!)
        vm = model.model_var_to_optimizer
        self.nvars = n = int(vm.max()) if len(vm) else 0
        self.sign = -1.0 if model.sense == "Maximize" else 1.0
        self._launches = []
        self._lazy_launches = []                  # device copies only an inspection (fetch()) reads
        self._host_mode = host

        # ---- objective: P (upper triangular), q, r
        obj = model.objective
        f = obj.f
        qblocks, lblocks = [], []
        direct_P = None
        if obj.kind == "quad" and not obj.isconstant and "P_values" in obj.dev:
            # the plan was specialised (moi._Record.compile): the Gram epilogue already writes sign * P in CSC order;
            # column c of P belongs to the variable with optimizer index c+1, its rows are the variables before it
            pv = obj.dev["P_vars"] - 1                                   # 0-based optimizer indices, strictly increasing
            k = len(pv)
            col_ptr = np.zeros(n + 1, dtype=np.int64)
            col_ptr[pv + 1] = np.arange(1, k + 1)
            col_ptr = np.cumsum(col_ptr)
            direct_P = CSC(n, n, col_ptr, pv[np.tril_indices(k)[1]].astype(np.int64), obj.dev["P_values"])
            at = f.affine_terms
            lblocks.append(_Block(at["var"].copy(), None, obj.dev["lin"], 16))
        elif obj.kind == "quad":
            qt, at = f.quadratic_terms, f.affine_terms
            if obj.isconstant:
                qblocks.append(_Block(vm[qt["row"] - 1], vm[qt["col"] - 1], host_coeff=qt["coeff"].copy()))
                lblocks.append(_Block(vm[at["var"] - 1], None, host_coeff=at["coeff"].copy()))
            else:
                qblocks.append(_Block(qt["row"].copy(), qt["col"].copy(), obj.dev["quad"], 24))
                lblocks.append(_Block(at["var"].copy(), None, obj.dev["lin"], 16))
        else:
            at = f.terms
            if obj.isconstant:
                lblocks.append(_Block(vm[at["var"] - 1], None, host_coeff=at["coeff"].copy()))
            else:
                lblocks.append(_Block(at["var"].copy(), None, obj.dev["terms"], 16))
        self.P = direct_P if direct_P is not None else self._build_matrix(qblocks, n, n, upper=True, alpha=self.sign)
        # q, l and u share ONE device block (q | l | u): a host delivery ships them as a single transfer
        mrows = sum(c.nrows for c in model.constraints)
        self._small_ptr = ctx.alloc(8 * max(n + 2 * mrows, 1))
        self._small_len = n + 2 * mrows
        self.q_ptr = self._small_ptr
        ctx.upload(self.q_ptr, np.zeros(max(n, 1)))
        for b in lblocks:
            self._add_vector_block(b, n, self.q_ptr, self.sign)
        self._obj_const = (None if obj.isconstant else obj.dev["const"], float(f.constant) if obj.isconstant else 0.0)

        # ---- constraints: A, l, u (rows stacked in update! order)
        ablocks, bounds, row0 = [], [], 0
        for c in model.constraints:
            if c.kind == "single":
                raise ArgumentError("DeviceQP: integer / binary constraints have no place in a QP hand-off")
            if c.kind == "quad":
                raise ArgumentError("DeviceQP: quadratic constraints are not part of the OSQP form")
            kind = _SET_KIND[type(c.set)]
            value = float(c.set.value) if isinstance(c.set, (moi.EqualTo, moi.GreaterThan, moi.LessThan)) and c.set.value is not None else 0.0
            if c.kind == "aff":
                t = c.f.terms
                rows = np.full(len(t), row0 + 1, dtype=np.int64)
                if c.isconstant:
                    ablocks.append(_Block(rows, vm[t["var"] - 1], host_coeff=t["coeff"].copy()))
                    bounds.append((row0, 1, kind, value, None, np.array([c.f.constant], dtype=np.float64)))
                else:
                    ablocks.append(_Block(rows, t["var"].copy(), c.dev["terms"], 16))
                    bounds.append((row0, 1, kind, value, c.dev["const"], None))
            else:
                t = c.f._terms          # (structure; the coefficients of a static dense block are not host data: moi.VectorAffineFunction)
                if c.isconstant:
                    ablocks.append(_Block(t["out"] + row0, vm[t["var"] - 1], host_coeff=t["coeff"].copy()))
                    bounds.append((row0, c.nrows, kind, value, None, np.asarray(c.f.constants, dtype=np.float64).copy()))
                else:
                    out = c.expr.out
                    if isinstance(out, DDenseAff) and not out.need_terms:
                        # a dense block's coefficients are the Parameter matrix itself (terms row-major: row*cols + col)
                        # (host_csc: the record packs no terms at all — moi._Record.compile — and the structure in c.f.terms is static)
                        ablocks.append(_Block(t["out"] + row0, t["var"].copy(), c.dev["terms"] + 8 if "terms" in c.dev else None, 24, dense=out.mat,
                                              param=_parameter_of(c.expr, out.mat)))
                    elif isinstance(out, DVarsAff) and not out.need_terms:
                        # x (+|-) v: the coefficient of every row is the 1.0 of copyto!(f, ::Variable) (src/functions.jl:421) — static
                        ablocks.append(_Block(t["out"] + row0, t["var"].copy(), host_coeff=t["coeff"].copy()))
                    elif isinstance(out, DSparseAff) and not out.need_terms:
                        # a sparse block's coefficients are read where they already are — the Parameter's nzval (CSC order) — not out of
                        # the 24-byte terms the scatter kernel wrote for the MOI side
                        ablocks.append(_Block(t["out"] + row0, t["var"].copy(), out.spmat.buf, 8, source_perm=out.spmat.perm))
                    else:
                        ablocks.append(_Block(t["out"] + row0, t["var"].copy(), c.dev["terms"] + 8, 24))
                    bounds.append((row0, c.nrows, kind, value, c.dev["consts"], None))
            row0 += c.nrows
        self.nrows = m = row0
        self.A = self._build_matrix(ablocks, m, n, upper=False, alpha=1.0)
        self.l_ptr, self.u_ptr = self._small_ptr + 8 * n, self._small_ptr + 8 * (n + m)
        if m and any(dev_consts is not None for (_, _, _, _, dev_consts, _) in bounds):
            # all rows in one launch: row i reads its constant through an address (constant functions get a device copy of theirs)
            cptr, kinds, values_ = np.zeros(m, dtype=np.uint64), np.zeros(m, dtype=np.int32), np.zeros(m)
            for (r0, nr, kind, value, dev_consts, host_consts) in bounds:
                base = dev_consts if dev_consts is not None else ctx.upload_new(np.ascontiguousarray(host_consts, dtype=np.float64))
                cptr[r0:r0 + nr] = np.uint64(base) + np.uint64(8) * np.arange(nr, dtype=np.uint64)
                kinds[r0:r0 + nr], values_[r0:r0 + nr] = kind, value
            self._launches.append(("pmt_qp_bounds_rows_f64", (P(ctx.upload_new(cptr)), P(ctx.upload_new(kinds)), P(ctx.upload_new(values_)), m, self.infty,
                                                              P(self.l_ptr), P(self.u_ptr))))
        else:
            for (r0, nr, kind, value, dev_consts, host_consts) in bounds:          # constant functions only: bounds never change
                b = value - host_consts
                lo = np.full(nr, -self.infty) if kind == 2 else b
                hi = np.full(nr, self.infty) if kind == 1 else b
                ctx.upload(self.l_ptr + 8 * r0, lo); ctx.upload(self.u_ptr + 8 * r0, hi)
        ctx.synchronize()
        self._in_tape = False
        self.host = HostQP(self) if host else None
        model._fetches_read_parameters = self.host is not None and self.A.only_rects
        if self.host is not None and self.A.only_rects and self.A.nnz:
            # A's static entries, once: the per-solve transfers rewrite the dense blocks only
            for name, args in self._lazy_launches:
                ctx.call(name, *args)
            ctx.fetch_ptr(self.host.Ax.ctypes.data, self.A.values_ptr, self.host.Ax.nbytes)
            ctx.synchronize()
        if host == "overlap" and not in_tape:
            in_tape = "main"                        # the fetches are tape entries, so their producers have to be too
        if in_tape and (self._launches or host == "overlap") and model._records:
            # NOTE (side lane): the q gather reads the objective's affine part, which the Gram node writes on the calling stream's side
            # stream; lane-1 entries are replayed on that same side stream (plan.hip `replay`, gram.hip `side_stream`), i.e. behind it.
            ctx.begin_record()
            try:
                ctx.set_lane(1 if in_tape == "side" else 0)
                for name, args in self._launches:
                    ctx.call(name, *args)
                if host == "overlap":
                    side_made = model._side_refreshed_parameter_ids() if in_tape == "side" else set()
                    # a lane-3 transfer is ordered against its Parameter's producer only when that producer ran on the SIDE stream; the
                    # Parameters concerned are marked, and whatever writes one of them on the plan's stream instead (the first replay below,
                    # before Model._mark_side_lane_parameters; a value read outside update!) synchronises behind it (lazyexpression.device_value_of)
                    for x_, _ in model._parameter_readers().values():
                        if id(x_) in side_made:
                            x_._read_unordered_by_lane3 = True
                    for t in self.host.transfers(late=False):
                        # a dense block straight out of its Parameter buffer depends on nothing of this re-evaluation: front of the side lane
                        # (lane 2) — and when the Parameter's value is itself produced on the side stream (only side-lane records read it),
                        # without waiting for the plan's stream at all (lane 3: not for the objective's callbacks either)
                        front = 3 if (t.param_id in side_made) else 2
                        ctx.set_lane(front if (t.height and in_tape == "side") else (1 if in_tape == "side" else 0))
                        t.record(ctx)
                ctx.set_lane(0)
                if host == "overlap":                       # behind the objective's own kernels on the plan's stream
                    for t in self.host.transfers(late=True):
                        t.record(ctx)
            finally:
                ctx.end_record()
            self._in_tape = True
            self._in_tape_lane = in_tape
            model._run_tape(fetch=False)
        else:
            self.refresh()
        if self.host is not None:
            self.host.wait()

    # ---- structure
    def _build_matrix(self, blocks, nrows, ncols, upper, alpha):
        ctx = self.ctx
        local = []
        for b in blocks:
            perm, seg, col_ptr, row_idx = _csc_order(b.rows, b.cols, nrows, ncols, upper)
            cols = np.repeat(np.arange(ncols, dtype=np.int64), np.diff(col_ptr))
            local.append((b, perm, seg, cols, row_idx))
        # union of the blocks' entries in CSC order; entries of different blocks never coincide for A (disjoint rows), and P has one block
        allc = np.concatenate([l[3] for l in local]) if local else np.zeros(0, dtype=np.int64)
        allr = np.concatenate([l[4] for l in local]) if local else np.zeros(0, dtype=np.int64)
        key = allc * max(nrows, 1) + allr
        ukey, inverse = np.unique(key, return_inverse=True)
        if len(ukey) != len(key):
            raise ArgumentError("DeviceQP: two blocks contribute to the same matrix entry")
        ucols, urows = ukey // max(nrows, 1), ukey % max(nrows, 1)
        col_ptr = np.zeros(ncols + 1, dtype=np.int64)
        np.add.at(col_ptr, ucols + 1, 1)
        col_ptr = np.cumsum(col_ptr)
        if len(local) == 1 and alpha == 1.0 and local[0][0].host_coeff is None and len(ukey):
            # one block whose entries already lie in the matrix's CSC order, one entry per run (a sparse constraint matrix handed over
            # as it came): the values ARE the source buffer; nothing to launch per re-evaluation
            b, perm, seg = local[0][0], local[0][1], local[0][2]
            src = perm if b.source_perm is None else b.source_perm[perm]
            if b.stride == 8 and len(seg) == len(perm) + 1 and np.array_equal(src, np.arange(len(src))):
                return CSC(nrows, ncols, col_ptr, urows.astype(np.int64), b.coeff_ptr)
        values = None
        static = np.zeros(max(len(ukey), 1))
        pos = 0
        addr, segs, dsts, nterms = [], [], [], 0            # the device blocks, folded into ONE gather launch (term addresses are static)
        rects = []
        for (b, perm, seg, cols, row_idx) in local:
            k = len(row_idx)
            dst = np.ascontiguousarray(inverse[pos:pos + k], dtype=np.int64)
            pos += k
            rect = self._dense_rect(b, perm, seg, dst, alpha) if k else None
            if rect is not None:
                rect.param = b.param
            if b.host_coeff is not None:
                sums = np.add.reduceat(b.host_coeff[perm], seg[:-1]) if k else np.zeros(0)
                static[dst] = alpha * sums
            elif rect is not None:
                rects.append(rect)
            elif k and b.coeff_ptr is None:
                raise ErrorException("DeviceQP: a dense block without MOI terms must leave as a copy of its Parameter matrix")
            elif k:
                src = perm if b.source_perm is None else b.source_perm[perm]
                addr.append(np.uint64(b.coeff_ptr) + src.astype(np.uint64) * np.uint64(b.stride))
                segs.append(seg[:-1] + nterms)
                dsts.append(dst)
                nterms += len(perm)
        pitched = all(r.offsets is None for r in rects)
        if len(rects) == 1 and pitched and not addr and rects[0].pitch == rects[0].mat.lda and len(ukey) == rects[0].mat.rows * rects[0].mat.cols:
            # the matrix IS one dense Parameter whose device copy is not padded: nothing to launch, nothing to copy
            return CSC(nrows, ncols, col_ptr, urows.astype(np.int64), rects[0].mat.buf, rects, only_rects=True)
        values = ctx.alloc(8 * max(len(ukey), 1))
        ctx.upload(values, static)
        for r in rects:
            off = None if r.offsets is None else P(ctx.upload_new(r.offsets))
            launch = ("pmt_copy_2d_f64", (P(r.mat.buf), r.mat.lda, P(values + 8 * r.first), r.pitch, off, r.mat.rows, r.mat.cols))
            # a host hand-off whose dense blocks leave straight from their Parameters needs the device copy only for inspection (fetch())
            (self._lazy_launches if (self._host_mode and not addr and pitched) else self._launches).append(launch)
        if addr:
            seg_all = np.concatenate(segs + [np.array([nterms], dtype=np.int64)]).astype(np.int64)
            dst_all = np.concatenate(dsts)
            identity = len(dst_all) == len(ukey) and np.array_equal(dst_all, np.arange(len(dst_all)))
            self._launches.append(("pmt_csc_values_gather_f64", (P(ctx.upload_new(np.concatenate(addr))), nterms, P(ctx.upload_new(seg_all)), len(dst_all),
                                                                 alpha, None if identity else P(ctx.upload_new(dst_all)), P(values))))
        return CSC(nrows, ncols, col_ptr, urows.astype(np.int64), values, rects, only_rects=bool(rects) and not addr and pitched)

    @staticmethod
    def _dense_rect(b, perm, seg, dst, alpha):
        """the block as a pitched copy of its Parameter matrix, or None: its columns must meet the matrix's columns in ascending order (the
        optimizer's indices of x ascending, the usual case) so that column j's values are a contiguous run, the runs a constant pitch apart"""
        mat = b.dense
        if mat is None or alpha != 1.0 or len(seg) != len(perm) + 1 or mat.rows == 0 or mat.cols == 0 or len(perm) != mat.rows * mat.cols:
            return None
        want = (np.arange(mat.rows, dtype=np.int64)[None, :] * mat.cols + np.arange(mat.cols, dtype=np.int64)[:, None]).reshape(-1)
        if not np.array_equal(perm, want):
            return None
        d = dst.reshape(mat.cols, mat.rows)
        if not np.array_equal(d, d[:, :1] + np.arange(mat.rows, dtype=np.int64)[None, :]):
            return None
        pitch = int(d[1, 0] - d[0, 0]) if mat.cols > 1 else mat.rows
        if pitch < mat.rows or not np.array_equal(d[:, 0], d[0, 0] + pitch * np.arange(mat.cols, dtype=np.int64)):
            return _Rect(0, 0, mat, offsets=np.ascontiguousarray(d[:, 0], dtype=np.int64))      # the other blocks' heights vary from column to column
        return _Rect(d[0, 0], pitch, mat)

    def _add_vector_block(self, b, n, q_ptr, alpha):
        ctx = self.ctx
        k = len(b.rows)
        if k == 0:
            return
        perm, seg, _, row_idx = _csc_order(b.rows, np.ones(k, dtype=np.int64), n, 1, False)
        if b.host_coeff is not None:
            q = np.zeros(n); q[row_idx] = alpha * np.add.reduceat(b.host_coeff[perm], seg[:-1])
            ctx.upload(q_ptr, q)
        else:
            self._launches.append(("pmt_csc_values_f64", (P(b.coeff_ptr), b.stride, k, P(ctx.upload_new(perm)), P(ctx.upload_new(seg)), len(row_idx),
                                                          alpha, P(ctx.upload_new(row_idx)), P(q_ptr))))

    # ---- per re-evaluation
    def refresh(self):
        """Rebuild P.x, q, A.x, l, u from the model's current device MOI buffers (call after update!(model) / solve!(model))."""
        if not self._in_tape:                       # otherwise part of the model's tape: rebuilt by update!(model) itself
            for name, args in self._launches:
                self.ctx.call(name, *args)
        if self.host is not None and not (self._in_tape and self._host_mode == "overlap"):
            for t in self.host.transfers():         # behind everything on the plan's stream (serial mode)
                t.fetch(self.ctx)

    # ---- host views (tests, host solvers)
    def _f64(self, ptr, n):
        out = np.empty(n)
        self.ctx.fetch(out, ptr, 8 * n)
        self.ctx.synchronize()
        return out

    def fetch(self):
        """dict(P=(x, i, p), q, r, A=(x, i, p), l, u) on the host; r is the objective constant."""
        for name, args in self._lazy_launches:
            self.ctx.call(name, *args)
        cptr, cval = self._obj_const
        r = float(self._f64(cptr, 1)[0]) if cptr else cval
        return {
            "P": (self._f64(self.P.values_ptr, self.P.nnz), self.P.row_idx, self.P.col_ptr),
            "q": self._f64(self.q_ptr, self.nvars), "r": self.sign * r,
            "A": (self._f64(self.A.values_ptr, self.A.nnz), self.A.row_idx, self.A.col_ptr),
            "l": self._f64(self.l_ptr, self.nrows), "u": self._f64(self.u_ptr, self.nrows),
        }
