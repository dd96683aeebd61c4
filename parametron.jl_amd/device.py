"""Device context and device value descriptors for one Model.

One Model = one pmt_plan (include/parametron_hip.h): it owns the HIP stream and every device buffer — the `dest`
of each lazy-expression node is allocated when the node is created (↔ `dest = deepcopy(expr())`,
src/lazyexpression.jl:202,230,243) and never reallocated, so steady-state update!() performs no allocation.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LT, QT, VAT, ArgumentError, DimensionMismatch


class DeviceContext:
    def __init__(self, device=0):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.plan = C.c_void_p()
        _lib.call("pmt_plan_create", int(device), None, C.byref(self.plan))
        self.stream = C.c_void_p(self.lib.pmt_plan_stream(self.plan))
        self.rec = C.c_void_p(self.lib.pmt_plan_recording_stream(self.plan))
        self.recording = False
        self._keep = []          # host arrays that must outlive asynchronous uploads
        self._keep_staged = []   # ... and staged uploads (copy stream)
        self._staging_dirty = False
        self._stage_slot = 0            # current staging slot; _next_stage_slot alternates per stage_parameters(), _pending_slot is what update! consumes
        self._next_stage_slot = 0
        self._pending_slot = 0

    def close(self):
        if self.plan:
            self.lib.pmt_plan_destroy(self.plan)
            self.plan = C.c_void_p()
            for p in getattr(self, "_pinned", []):
                self.lib.pmt_host_free(C.c_void_p(p))
            self._pinned = []

    def pinned_array(self, n, dtype):
        """numpy array of `n` elements in page-locked host memory (the MOI function buffers the device results are copied into).
        The memory lives until close(); the array must not be used after that."""
        dtype = np.dtype(dtype)
        nbytes = max(int(n) * dtype.itemsize, 16)
        p = C.c_void_p()
        _lib.call("pmt_host_alloc", nbytes, C.byref(p))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p.value)
        buf = (C.c_char * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(n))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        _lib.call("pmt_plan_alloc", self.plan, int(nbytes), C.byref(p))
        return p.value

    def bytes_allocated(self):
        return int(self.lib.pmt_plan_bytes_allocated(self.plan))

    def upload(self, dptr, host):
        host = np.ascontiguousarray(host)
        if host.nbytes == 0:
            return
        self._keep.append(host)
        _lib.call("pmt_plan_upload", self.plan, C.c_void_p(dptr), host.ctypes.data_as(C.c_void_p), host.nbytes)

    def upload_new(self, host):
        host = np.ascontiguousarray(host)
        p = self.alloc(max(host.nbytes, 8))
        self.upload(p, host)
        return p

    def fetch(self, host, dptr, nbytes):
        if nbytes == 0:
            return
        _lib.call("pmt_plan_fetch", self.plan, host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), int(nbytes))

    def record_fetch(self, host, dptr, nbytes):
        """while recording: a D2H copy as a tape entry of the current lane; at replay it leaves on the plan's FETCH stream as soon as the
        entries recorded before it on that lane are done, while the rest of the tape is still busy (pmt_plan_record_fetch)"""
        if nbytes == 0:
            return
        _lib.call("pmt_plan_record_fetch", self.plan, host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), int(nbytes))

    def fetch_ptr(self, host_ptr, dptr, nbytes):
        if nbytes:
            _lib.call("pmt_plan_fetch", self.plan, C.c_void_p(host_ptr), C.c_void_p(dptr), int(nbytes))

    def fetch_2d_ptr(self, host_ptr, dst_pitch, dptr, src_pitch, width_bytes, height):
        if width_bytes and height:
            _lib.call("pmt_plan_fetch_2d", self.plan, C.c_void_p(host_ptr), int(dst_pitch), C.c_void_p(dptr), int(src_pitch), int(width_bytes), int(height))

    def record_fetch_ptr(self, host_ptr, dptr, nbytes):
        if nbytes:
            _lib.call("pmt_plan_record_fetch", self.plan, C.c_void_p(host_ptr), C.c_void_p(dptr), int(nbytes))

    def record_fetch_2d(self, host_ptr, dst_pitch, dptr, src_pitch, width_bytes, height):
        """while recording: the pitched form of record_fetch (pmt_plan_record_fetch_2d) — `height` rows of `width_bytes`"""
        if width_bytes and height:
            _lib.call("pmt_plan_record_fetch_2d", self.plan, C.c_void_p(host_ptr), int(dst_pitch), C.c_void_p(dptr), int(src_pitch), int(width_bytes),
                      int(height))

    def fetch_synchronize(self):
        """host: every copy on the plan's fetch stream (recorded fetches, delivered CSC values) has landed"""
        _lib.call("pmt_plan_fetch_synchronize", self.plan)

    def zero(self, dptr, nbytes):
        _lib.call("pmt_plan_zero", self.plan, C.c_void_p(dptr), int(nbytes))

    def synchronize(self):
        _lib.call("pmt_plan_synchronize", self.plan)
        self._keep.clear()
        self._replay_pending = False

    # ---- staged uploads (copy stream; include/parametron_hip.h "Staged (overlapped) uploads")
    def stage_upload(self, staging_ptr, host):
        host = np.ascontiguousarray(host)
        if host.nbytes == 0:
            return
        self._keep_staged.append(host)
        _lib.call("pmt_plan_stage_upload", self.plan, C.c_void_p(staging_ptr), host.ctypes.data_as(C.c_void_p), host.nbytes)

    def commit_staged(self, dptr, staging_ptr, nbytes):
        _lib.call("pmt_plan_commit_staged", self.plan, C.c_void_p(dptr), C.c_void_p(staging_ptr), int(nbytes))

    def wait_staged(self):
        _lib.call("pmt_plan_wait_staged", self.plan)

    def commit_lane(self, lane):
        """0: the commits that follow go to the plan's stream; 1: to its side stream (Parameters only side-lane entries read)"""
        _lib.call("pmt_plan_commit_lane", self.plan, int(lane))

    def staging_consumed(self):
        _lib.call("pmt_plan_staging_consumed", self.plan)

    def set_stage_slot(self, slot):
        """the staging slot (0 / 1) the stage / wait / commit / consumed calls act on; Model.stage_parameters() alternates it so that the
        copy of update k+1 does not wait for the commits of update k"""
        self._stage_slot = int(slot)
        _lib.call("pmt_plan_stage_slot", self.plan, int(slot))

    def staged_synchronize(self):
        """host: the staged uploads issued so far have left the host buffers"""
        _lib.call("pmt_plan_staged_synchronize", self.plan)
        self._keep_staged.clear()

    # ---- launches: immediate on the plan's stream, or appended to the tape while recording
    def launch_stream(self):
        return self.rec if self.recording else self.stream

    def call(self, name, *args):
        _lib.call(name, *args, self.launch_stream())

    def call_on_lane(self, lane, name, *args):
        """an immediate call on the stream a lane's entries are replayed on (pmt_plan_lane_stream): device-side Parameter callbacks of values
        only side-lane entries read"""
        if not hasattr(self, "_lane_streams"):
            self._lane_streams = {}
        if lane not in self._lane_streams:
            st = C.c_void_p()
            _lib.call("pmt_plan_lane_stream", self.plan, int(lane), C.byref(st))
            self._lane_streams[lane] = st
        _lib.call(name, *args, self._lane_streams[lane])

    def call_now(self, name, *args):
        """setup-time call on the plan's stream even while recording (e.g. the one-off device ordering of canonicalize!)"""
        _lib.call(name, *args, self.stream)

    def begin_record(self):
        _lib.call("pmt_plan_begin_record", self.plan)
        self.recording = True

    def end_record(self):
        _lib.call("pmt_plan_end_record", self.plan)
        self.recording = False

    def set_lane(self, lane):
        """while recording: 1 = the following calls are side-lane entries (independent of the rest of the tape), 0 = back to the plan's stream"""
        _lib.call("pmt_plan_set_lane", self.plan, int(lane))

    def replay(self):
        _lib.call("pmt_plan_update", self.plan)
        self._replay_pending = True           # (a replay may still be reading host mailboxes: lazyexpression.device_value_of)

    def instantiate_graph(self):
        _lib.call("pmt_plan_instantiate_graph", self.plan)

    def tape_length(self):
        return int(self.lib.pmt_plan_tape_length(self.plan))

    def fused(self):
        """small plans (include/parametron_hip.h): {groups, nodes, exec_length} — runs of small tape entries replayed as ONE launch each, the
        tape entries they replace, and what one replay executes"""
        g, n, ln = C.c_int(), C.c_int(), C.c_int64()
        _lib.call("pmt_plan_fused", self.plan, C.byref(g), C.byref(n), C.byref(ln))
        return {"groups": g.value, "nodes": n.value, "exec_length": ln.value, "phases": int(self.lib.pmt_plan_fused_phases(self.plan)),
                "workgroups": int(self.lib.pmt_plan_fused_workgroups(self.plan))}

    def set_fusion(self, on):
        _lib.call("pmt_plan_set_fusion", self.plan, 1 if on else 0)


def P(ptr):
    return C.c_void_p(ptr) if ptr else None


# ---------------------------------------------------------------------------------------------------------
# device value descriptors (what a node's `dest` looks like in HBM)

class DV:
    kind = "?"


class DNum(DV):
    """Number: f64[1]."""
    kind = "num"

    def __init__(self, ctx, value=None):
        self.buf = ctx.alloc(8)
        if value is not None:
            ctx.upload(self.buf, np.array([value], dtype=np.float64))


class DVec(DV):
    """Vector{Float64}: f64[n]."""
    kind = "vec"

    def __init__(self, ctx, n):
        self.n = int(n)
        self.padded = row_padded(self.n)                 # allocated length; entries [n, padded) stay zero
        self.buf = ctx.alloc(8 * max(self.padded, 1))
        ctx.zero(self.buf, 8 * max(self.padded, 1))


def row_padded(rows):
    """Row counts of at least 64 are padded to a multiple of 16 with ZERO rows in the device copies of Parameter matrices and
    vectors: the Gram kernel's branch-free path needs whole 16-row stages, and zero rows add nothing to A'A, A'c or c'c, so the
    host simply passes the padded count (4090 rows: 1.31 ms through the bounds-checked loop, 1.18 ms padded to 4096)."""
    rows = int(rows)
    return (rows + 15) // 16 * 16 if rows >= 64 else rows


def padded_lda(rows):
    """Leading dimension of the DEVICE copy of a column-major matrix with `rows` rows.  A column stride that is a multiple of
    4 KiB (e.g. 4096 doubles = 32 KiB, the BASELINE shape) maps every column segment of a tile onto the same memory channel;
    measured on MI355X (warm GPU): +64 doubles takes the affine-assembly kernel from 6.05 to 6.69 TB/s; the MFMA-bound Gram
    kernel is indifferent (profiles/r01c_lda_padding.txt).  The values and their (row, column) meaning are unchanged — only the
    placement in HBM."""
    rows = row_padded(rows)
    return rows + 64 if rows >= 512 and rows % 512 == 0 else rows


class DMat(DV):
    """Matrix{Float64}, column-major with leading dimension lda >= rows (padded_lda)."""
    kind = "mat"

    def __init__(self, ctx, rows, cols):
        self.rows, self.cols = int(rows), int(cols)
        self.lda = padded_lda(self.rows)
        self.buf = ctx.alloc(8 * max(self.lda * self.cols, 1))
        if self.lda != self.rows:
            ctx.zero(self.buf, 8 * self.lda * self.cols)   # the padding rows are never written again (pitched copies, fill kernels)

    def upload(self, ctx, m):
        """host ndarray (rows, cols) -> device copy (column j at buf + j*lda*8).  A column-major (Julia order) array is copied as
        it is; a row-major one (numpy's default) is uploaded as it is too and transposed ON THE DEVICE — converting 4096 x 4096 on
        the host with numpy takes 0.6 s, the device transpose 0.1 ms."""
        m = np.asarray(m, dtype=np.float64)
        if not (self.rows and self.cols):
            return
        if m.flags.c_contiguous and not m.flags.f_contiguous:
            if getattr(self, "_stage", None) is None:
                self._stage = ctx.alloc(8 * self.rows * self.cols)
            ctx.upload(self._stage, m)
            # the row-major bytes are a column-major (cols x rows) matrix with leading dimension cols
            _lib.call("pmt_transpose_f64", C.c_void_p(self._stage), self.cols, self.cols, self.rows, C.c_void_p(self.buf), self.lda, ctx.stream)
            return
        m = np.asfortranarray(m)
        ctx._keep.append(m)
        _lib.call("pmt_plan_upload_2d", ctx.plan, C.c_void_p(self.buf), 8 * self.lda, m.ctypes.data_as(C.c_void_p), 8 * self.rows,
                  8 * self.rows, self.cols)

    def stage(self, ctx, m):
        """start a STAGED upload of the value (copy stream): a column-major array goes into a second buffer with the padded layout, a
        row-major one is staged as it is and transposed by the commit"""
        m = np.asarray(m, dtype=np.float64)
        if not (self.rows and self.cols):
            self._staged_kind = None
            return
        slot = ctx._stage_slot                                     # one pair of staging buffers per slot (pmt_plan_stage_slot)
        if not hasattr(self, "_stage_rm"):
            self._stage_rm, self._staging_cm = {}, {}
        self._staged_slot = slot
        if m.flags.c_contiguous and not m.flags.f_contiguous:
            if slot not in self._stage_rm:
                self._stage_rm[slot] = ctx.alloc(8 * self.rows * self.cols)
            ctx.stage_upload(self._stage_rm[slot], m)
            self._staged_kind = "rowmajor"
            return
        if slot not in self._staging_cm:
            self._staging_cm[slot] = ctx.alloc(8 * self.lda * self.cols)
            if self.lda != self.rows:
                ctx.zero(self._staging_cm[slot], 8 * self.lda * self.cols)
                ctx.synchronize()                                  # (setup: the padding rows are zero before the copy stream writes beside them)
        m = np.asfortranarray(m)
        ctx._keep_staged.append(m)
        _lib.call("pmt_plan_stage_upload_2d", ctx.plan, C.c_void_p(self._staging_cm[slot]), 8 * self.lda, m.ctypes.data_as(C.c_void_p), 8 * self.rows,
                  8 * self.rows, self.cols)
        self._staged_kind = "colmajor"

    def commit(self, ctx):
        """plan stream: the staged value becomes the Parameter's value"""
        kind = getattr(self, "_staged_kind", None)
        slot = getattr(self, "_staged_slot", 0)
        if kind == "rowmajor":
            ctx.wait_staged()
            _lib.call("pmt_transpose_f64", C.c_void_p(self._stage_rm[slot]), self.cols, self.cols, self.rows, C.c_void_p(self.buf), self.lda, ctx.stream)
        elif kind == "colmajor":
            ctx.commit_staged(self.buf, self._staging_cm[slot], 8 * self.lda * self.cols)
        self._staged_kind = None

    def fetch(self, ctx):
        out = np.empty((self.rows, self.cols), dtype=np.float64, order="F")
        if self.rows and self.cols:
            _lib.call("pmt_plan_fetch_2d", ctx.plan, out.ctypes.data_as(C.c_void_p), 8 * self.rows, C.c_void_p(self.buf), 8 * self.lda,
                      8 * self.rows, self.cols)
        return out


class DVars(DV):
    """Vector{Variable}: static Int64 indices; `lt` = the same variables as LinearTerm(1.0, var) (copyto!(f, ::Variable), src/functions.jl:421)."""
    kind = "varvec"

    def __init__(self, ctx, variables):
        self.vars = np.array([v.index for v in variables], dtype=np.int64)
        self.n = len(self.vars)
        self.buf = ctx.upload_new(self.vars) if self.n else ctx.alloc(8)
        self._lt = None
        self.ctx = ctx

    def lt(self):
        if self._lt is None:
            t = np.empty(self.n, dtype=LT)
            t["coeff"] = 1.0
            t["var"] = self.vars
            self._lt = self.ctx.upload_new(t)
        return self._lt

    def strictly_increasing(self):
        return bool(np.all(np.diff(self.vars) > 0))


class DLinVec(DV):
    """Vector{LinearTerm{Float64}}: LT[n]."""
    kind = "ltvec"

    def __init__(self, ctx, n):
        self.n = int(n)
        self.terms = ctx.alloc(16 * max(self.n, 1))


class DAffVec(DV):
    """Vector{AffineFunction{Float64}}: flat LT buffer + per-row constants; uniform rows (row_len) or ragged (row_ptr)."""
    kind = "affvec"

    def __init__(self, ctx, rows, row_len=None, row_ptr=None, alloc=True):
        self.ctx = ctx
        self.rows = int(rows)
        if row_ptr is not None:
            self.row_ptr = np.asarray(row_ptr, dtype=np.int64)
            self.row_len = 0
            self.nterms = int(self.row_ptr[-1]) if self.rows else 0
            lens = np.diff(self.row_ptr)
            if self.rows and np.all(lens == lens[0]):          # uniform after all
                self.row_len, self.row_ptr = int(lens[0]), None
        else:
            self.row_ptr = None
            self.row_len = int(row_len)
            self.nterms = self.rows * self.row_len
        self.row_ptr_buf = ctx.upload_new(self.row_ptr) if self.row_ptr is not None else None
        self.terms = ctx.alloc(16 * max(self.nterms, 1)) if alloc else None
        self.consts = ctx.alloc(8 * max(self.rows, 1)) if alloc else None

    def uniform(self):
        return self.row_ptr is None

    def host_row_ptr(self):
        if self.row_ptr is not None:
            return self.row_ptr
        return np.arange(self.rows + 1, dtype=np.int64) * self.row_len

    def materialized(self):
        return self


class DDenseAff(DAffVec):
    """A*x (+|-) b kept implicit (A, xvar, b, sign): variable indices are the column's, so nothing but A and b is read.
    The LinearTerm block is written only if a consumer needs it (`require_terms`)."""

    def __init__(self, ctx, mat, xvars, vec, sign):
        super().__init__(ctx, mat.rows, row_len=mat.cols, alloc=False)
        self.mat, self.xvars, self.vec, self.sign = mat, xvars, vec, sign
        self.need_terms = False

    def require_terms(self):
        if not self.need_terms:
            self.need_terms = True
            self.terms = self.ctx.alloc(16 * max(self.nterms, 1))
            self.consts = self.ctx.alloc(8 * max(self.rows, 1))
        return self

    def materialized(self):
        return self.require_terms()


class DVarsAff(DAffVec):
    """x (+|-) v for x::Vector{Variable} kept implicit: one term (1.0, x[i]) per row."""

    def __init__(self, ctx, xvars, vec, sign):
        super().__init__(ctx, xvars.n, row_len=1, alloc=False)
        self.xvars, self.vec, self.sign = xvars, vec, sign
        self.need_terms = False

    def require_terms(self):
        if not self.need_terms:
            self.need_terms = True
            self.terms = self.ctx.alloc(16 * max(self.nterms, 1))
            self.consts = self.ctx.alloc(8 * max(self.rows, 1))
        return self

    def materialized(self):
        return self.require_terms()


class DSpMat(DV):
    """SparseMatrixCSC{Float64,Int64} with a FIXED pattern: only nzval lives on the device per re-evaluation; the row-major
    order of the structural non-zeros (perm, term_row, term_col, row_ptr) is computed once (pmt_sparse_rowmajor_order)."""
    kind = "spmat"

    def __init__(self, ctx, csc):
        import ctypes as C
        self.rows, self.cols = csc.shape
        self.nnz = int(csc.nnz)
        colptr = np.ascontiguousarray(csc.indptr, dtype=np.int64) + 1            # Julia 1-based
        rowval = np.ascontiguousarray(csc.indices, dtype=np.int64) + 1
        self.indptr, self.indices = csc.indptr.copy(), csc.indices.copy()
        self.perm = np.empty(self.nnz, dtype=np.int64)
        self.term_row = np.empty(self.nnz, dtype=np.int64)
        self.term_col = np.empty(self.nnz, dtype=np.int64)
        self.row_ptr = np.empty(self.rows + 1, dtype=np.int64)
        vp = C.c_void_p
        _lib.call("pmt_sparse_rowmajor_order", self.rows, self.cols, colptr.ctypes.data_as(vp), rowval.ctypes.data_as(vp),
                  self.perm.ctypes.data_as(vp), self.term_row.ctypes.data_as(vp), self.term_col.ctypes.data_as(vp), self.row_ptr.ctypes.data_as(vp))
        self.buf = ctx.alloc(8 * max(self.nnz, 1))                                # nzval
        self.narrow = self.nnz < 2 ** 32                                        # 32-bit index streams (sparse.hip, IDX = uint32_t)
        self.perm_buf = (ctx.upload_new(self.perm.astype(np.uint32) if self.narrow else self.perm)) if self.nnz else ctx.alloc(8)
        # XCD-aware scatter: per-row boundaries of 8 column slabs (one per XCD), see sparse.hip
        self.nslab = 8
        self.slab_ptr = np.zeros(max(self.rows, 1) * (self.nslab + 1), dtype=np.int64)
        if self.rows:
            _lib.call("pmt_sparse_slab_ptr", self.rows, self.cols, self.nslab, self.row_ptr.ctypes.data_as(vp), self.term_col.ctypes.data_as(vp),
                      self.slab_ptr.ctypes.data_as(vp))
        self.slab_ptr_buf = ctx.upload_new(self.slab_ptr)
        # block form (CSC -> row-major through LDS, sparse.hip): used when the pattern allows it and a row's part of a column band is long
        # enough to be written as a run (>= 16 terms on average); block_cw == 0: slab form
        self.block_cw = 0
        if self.nnz and self.rows and self.cols:
            cw = C.c_int(0)
            _lib.call("pmt_sparse_blocks_width", self.rows, self.cols, colptr.ctypes.data_as(vp), rowval.ctypes.data_as(vp), C.byref(cw))
            cw = cw.value
            if cw and self.nnz >= 16 * self.rows * (-(-self.cols // cw)):
                nrb, ncb = -(-self.rows // 128), -(-self.cols // cw)
                desc = np.zeros(nrb * self.cols, dtype=np.uint64)
                idx = np.zeros(self.nnz, dtype=np.uint32)
                band = np.zeros(self.rows * (ncb + 1), dtype=np.int64)
                _lib.call("pmt_sparse_blocks_build", self.rows, self.cols, colptr.ctypes.data_as(vp), rowval.ctypes.data_as(vp),
                          self.perm.ctypes.data_as(vp), self.term_col.ctypes.data_as(vp), self.row_ptr.ctypes.data_as(vp), cw,
                          desc.ctypes.data_as(vp), idx.ctypes.data_as(vp), band.ctypes.data_as(vp))
                self.block_cw = cw
                self.block_desc_buf, self.block_idx_buf, self.block_band_buf = ctx.upload_new(desc), ctx.upload_new(idx), ctx.upload_new(band)

    def same_pattern(self, csc):
        return csc.shape == (self.rows, self.cols) and np.array_equal(csc.indptr, self.indptr) and np.array_equal(csc.indices, self.indices)


class DSparseAff(DAffVec):
    """C*x (+|-) d for a sparse C kept implicit; terms exist for the structural non-zeros only, in row-major order."""

    def __init__(self, ctx, spmat, xvars, vec, sign):
        super().__init__(ctx, spmat.rows, row_ptr=spmat.row_ptr.copy(), alloc=False)
        if self.row_ptr is None:                      # DAffVec collapsed a uniform pattern: keep the explicit row_ptr semantics
            self.row_ptr = spmat.row_ptr.copy()
            self.row_len = 0
            self.row_ptr_buf = ctx.upload_new(self.row_ptr)
        self.spmat, self.xvars, self.vec, self.sign = spmat, xvars, vec, sign
        self.term_var = xvars.vars[spmat.term_col - 1] if spmat.nnz else np.zeros(0, dtype=np.int64)
        self._term_var_buf = None
        self.need_terms = False

    def index_stream(self, values):
        """device copy of a per-term index array in the width the matrix's perm stream has (both streams of a launch share one type)"""
        if not self.spmat.nnz:
            return self.ctx.alloc(8)
        if self.spmat.narrow and (values.max() >= 2 ** 32 or values.min() < 0):
            raise DimensionMismatch("sparse node: variable indices of 2^32 or more with a 32-bit pattern")
        return self.ctx.upload_new(values.astype(np.uint32) if self.spmat.narrow else np.ascontiguousarray(values, dtype=np.int64))

    @property
    def term_var_buf(self):
        if self._term_var_buf is None:
            self._term_var_buf = self.index_stream(self.term_var)
        return self._term_var_buf

    def uniform(self):
        return False

    def require_terms(self):
        if not self.need_terms:
            self.need_terms = True
            self.terms = self.ctx.alloc(16 * max(self.nterms, 1))
            self.consts = self.ctx.alloc(8 * max(self.rows, 1))
        return self

    def materialized(self):
        return self.require_terms()


class DAff(DV):
    """AffineFunction{Float64}: LT[nterms] + f64[1]."""
    kind = "aff"

    def __init__(self, ctx, nterms, alloc=True):
        self.nterms = int(nterms)
        self.terms = ctx.alloc(16 * max(self.nterms, 1)) if alloc else None
        self.const = ctx.alloc(8) if alloc else None


class DQuad(DV):
    """QuadraticFunction{Float64}: QT[nq] + LT[nl] + f64[1].  Buffers may be deferred (`materialize`) because the literal
    expansion of a large residual . residual cannot exist in memory."""
    kind = "quad"

    def __init__(self, ctx, nq, nl, alloc=True):
        self.ctx = ctx
        self.nq, self.nl = int(nq), int(nl)
        self.quad = self.lin = self.const = None
        if alloc:
            self.materialize()

    def materialize(self):
        if self.quad is None:
            need = 24 * self.nq + 16 * self.nl
            if need > (200 << 30):
                raise MemoryError("the literal (uncombined) expansion needs %.1f GB of QuadraticTerms; use the canonical "
                                  "objective mode (Model(..., quadratic_mode='canonical'))" % (need / 1e9))
            self.quad = self.ctx.alloc(24 * max(self.nq, 1))
            self.lin = self.ctx.alloc(16 * max(self.nl, 1))
            self.const = self.ctx.alloc(8)
        return self


def fetch_terms(ctx, ptr, n, dtype):
    out = np.empty(int(n), dtype=dtype)
    ctx.fetch(out, ptr, out.nbytes)
    return out


def fetch_f64(ctx, ptr, n):
    out = np.empty(int(n), dtype=np.float64)
    ctx.fetch(out, ptr, out.nbytes)
    return out


__all__ = ["DeviceContext", "DV", "DNum", "DVec", "DMat", "DVars", "DLinVec", "DAffVec", "DDenseAff", "DVarsAff", "DSpMat", "DSparseAff", "DAff", "DQuad",
           "fetch_terms", "fetch_f64", "P", "LT", "QT", "VAT", "ArgumentError"]
