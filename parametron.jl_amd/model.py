"""Model: the user API and the driver of the hot path (src/model.jl:1-274).

    model = Model(optimizer)
    x = [Variable(model) for _ in range(n)]
    A = Parameter(f, np.zeros((n, n)), model) ...
    residual = A * x - b                       # @expression A * x - b
    objective(model, Minimize, dot(residual, residual))
    constraint(model, C * x == d)
    solve(model)                               # solve!(model): initialize! once, update!, optimize!

update!(model) = setdirty!; refresh dirty Parameters in the reference's evaluation order (host callback + H2D copy, or a
device fill); replay the recorded tape of node kernels and MOI-pack kernels on the model's HIP stream
(pmt_plan_update — no allocation, no host term bookkeeping); copy the MOI buffers to the host function objects and hand
them to the optimizer with MOI.set (third party from there on).
"""
import ctypes as C

import numpy as np

from . import moi
from ._lib import ArgumentError, DimensionMismatch, ErrorException
from .device import DeviceContext
from .functions import AffineFunction, Variable, _isnum
from .lazyexpression import DeviceNode, Relation, evaluate, lazy, schedule
from .parameter import Parameter

Minimize, Maximize = "Minimize", "Maximize"


class AbstractOptimizer:
    """The slice of MathOptInterface the reference drives (src/model.jl:118,157,166-194; src/moi_interop.jl:134,171).
    A concrete optimizer wraps a solver; the package ships only MockOptimizer (↔ src/mockmodel.jl)."""

    def copy_to(self, backend):
        """MOI.copy_to(optimizer, backend) -> index map {'variables': int64[nvars] optimizer index of Variable k (1-based),
        'constraints': {constraint: optimizer constraint index}}"""
        raise NotImplementedError

    def set_objective_function(self, f):
        raise NotImplementedError

    def set_constraint_function(self, index, f):
        raise NotImplementedError

    def optimize(self):
        raise NotImplementedError

    def variable_primal(self, index):
        raise NotImplementedError

    def objective_value(self):
        raise NotImplementedError

    def termination_status(self):
        raise NotImplementedError

    def primal_status(self):
        raise NotImplementedError

    def dual_status(self):
        raise NotImplementedError


class MockOptimizer(AbstractOptimizer):
    """mock_model()'s optimizer (src/mockmodel.jl:3-6): records what it is given, optimises nothing."""

    def __init__(self, variable_offset=0):
        self.variable_offset = variable_offset
        self.objective = None
        self.constraints = {}
        self.sets = {}
        self.nvars = 0
        self.sense = None
        self.optimize_calls = 0
        self.set_calls = 0

    def copy_to(self, backend):
        self.nvars = backend.nvars
        self.sense = backend.sense
        self.objective = backend.objective.f
        cmap = {}
        for i, c in enumerate(backend.constraints):
            cmap[c] = i
            self.constraints[i] = c.f
            self.sets[i] = c.set
        return {"variables": np.arange(1, backend.nvars + 1, dtype=np.int64) + self.variable_offset, "constraints": cmap}

    def set_device_qp(self, qp):
        self.device_qp = qp

    def set_host_qp(self, qp):
        self.host_qp = qp

    def set_objective_function(self, f):
        self.objective = f
        self.set_calls += 1

    def set_constraint_function(self, index, f):
        self.constraints[index] = f
        self.set_calls += 1

    def optimize(self):
        self.optimize_calls += 1

    def variable_primal(self, index):
        return 0.0

    def objective_value(self):
        return 0.0

    def termination_status(self):
        return "OPTIMIZE_NOT_CALLED" if not self.optimize_calls else "OPTIMAL"

    def primal_status(self):
        return "NO_SOLUTION"

    def dual_status(self):
        return "NO_SOLUTION"


def _mailbox_writer(x):
    """_write_mailbox with the layout resolved once: val -> the mailbox, in the device layout"""
    from .device import DMat, DVec
    dv, mb = x._dev, x._mailbox
    if isinstance(dv, DMat):
        view, shape = mb.reshape(dv.cols, dv.lda)[:, :dv.rows], (dv.rows, dv.cols)

        def write(val):
            m = np.asarray(val, dtype=np.float64)
            if m.shape != shape:
                raise DimensionMismatch("Parameter changed shape: %r -> %r" % (shape, m.shape))
            view[...] = m.T
    elif isinstance(dv, DVec):
        view, shape = mb[:dv.n], (dv.n,)

        def write(val):
            v = np.asarray(val, dtype=np.float64)
            if v.shape != shape:
                raise DimensionMismatch("Parameter changed shape: %r -> %r" % (shape, v.shape))
            view[...] = v
    else:
        def write(val):
            mb[0] = float(val)
    return write


def _write_mailbox(x, val):
    """the value of a host-updated Parameter into its page-locked mailbox, in the layout of the device buffer (padded leading dimension)"""
    from .device import DMat, DVec
    if val is None:
        return
    dv, mb = x._dev, x._mailbox
    if isinstance(dv, DMat):
        m = np.asarray(val, dtype=np.float64)
        if m.shape != (dv.rows, dv.cols):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.rows, dv.cols), m.shape))
        mb.reshape(dv.cols, dv.lda)[:, :dv.rows] = m.T
    elif isinstance(dv, DVec):
        v = np.asarray(val, dtype=np.float64)
        if v.shape != (dv.n,):
            raise DimensionMismatch("Parameter changed shape: %r -> %r" % ((dv.n,), v.shape))
        mb[:dv.n] = v
    else:
        mb[0] = float(val)


class _Backend:
    """What MOI.copy_to sees of the ParametronMOIModel backend (src/moi_interop.jl:2-11)."""

    def __init__(self, nvars, sense, objective, constraints):
        self.nvars, self.sense, self.objective, self.constraints = nvars, sense, objective, constraints


class Model:
    def __init__(self, optimizer, quadratic_mode="auto", device=0, use_graph=False, handoff="moi", side_lane=True, overlap_fetch=True):       # src/model.jl:10-22
        if quadratic_mode not in ("auto", "literal", "canonical"):
            raise ArgumentError("quadratic_mode must be 'auto', 'literal' or 'canonical'")
        if handoff not in ("moi", "device", "host_csc"):
            raise ArgumentError("handoff must be 'moi' (host MOI functions, the reference's boundary), 'device' (CSC data in HBM) or "
                                "'host_csc' (CSC values / q / bounds in page-locked host arrays: what a host OSQP's update takes)")
        self.handoff = handoff
        # host_csc: the arrays leave for the host while the re-evaluation is still running (recorded fetches, band-wise delivery of P);
        # False fetches them behind it (the A/B in bench.py)
        self._overlap_fetch = bool(overlap_fetch) and handoff == "host_csc"
        # the reference's own boundary (MOI function objects on the host): the MOI buffers leave as recorded fetches while the tape is still
        # running and the objective's quadratic terms row band by row band out of the contraction (pmt_quad_gram_deliver_f64) — not under
        # a hipGraph replay (copies leave the capture)
        self._overlap_moi = bool(overlap_fetch) and handoff == "moi" and not use_graph
        if handoff == "host_csc" and use_graph:
            raise ArgumentError("handoff='host_csc' replays the tape as launches (its copies leave a graph capture); use_graph must be False")
        self.device_qp = None
        self.params = []
        self.optimizer = optimizer
        self.initialized = False
        self.nvars = 0
        self.sense = Minimize
        self.objective = moi.Objective(self, AffineFunction.zero())     # default objective (issue #62, src/model.jl:15)
        self.constraints = moi.Constraints()
        self.model_var_to_optimizer = np.zeros(0, dtype=np.int64)
        self.quadratic_mode = quadratic_mode
        self._device_index = device
        self._ctx = None
        self._use_graph = use_graph
        self._side_lane = side_lane
        self._records = []

    def __repr__(self):
        return "Model{Float64, %s}(…)" % type(self.optimizer).__name__

    # ---- device
    def device(self):
        if self._ctx is None:
            self._ctx = DeviceContext(self._device_index)            # raises loudly without a GPU / built library
        return self._ctx

    def parameter_array(self, *shape):
        """Page-locked, column-major float64 array for the value of a host-updated Parameter (`Parameter(f, val, model)`,
        `Parameter(model, val=val)`, src/parameter.jl:57,88).  Uploads from it are asynchronous and run at PCIe speed; an ordinary
        numpy array works too, but a pageable source is staged by the driver and a row-major matrix is first converted to Julia's
        column-major order on the host.  The memory lives until close()."""
        shape = tuple(int(s) for s in (shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape))
        n = int(np.prod(shape)) if shape else 1
        flat = self.device().pinned_array(n, np.float64)
        flat[:] = 0.0
        return flat.reshape(shape, order="F")

    def close(self):
        if self._ctx is not None:
            if getattr(self, "_model_run", None) is not None:
                self._ctx.lib.pmt_model_destroy(self._model_run)
                self._model_run = None
            self._ctx.close()
            self._ctx = None

    # ---- setup API
    def setdirty(self):                                                # src/model.jl:40
        for p in self.params:
            p.setdirty()

    # ---- overlapped uploads of host-updated Parameters
    def stage_parameters(self):
        """Evaluate the host-updated Parameters NOW — exactly what the next update!() would do with them (setdirty! + call,
        src/parameter.jl:93-104) — and start copying their values to the device on the plan's COPY stream.  The call returns at once; the
        PCIe copy runs while the device is still busy with the previous re-evaluation, and the next update!() / solve!() consumes the
        staged values on the device instead of uploading (they are not evaluated a second time).  Call it as soon as the values of the
        next solve are in their host buffers; the buffers must not be overwritten again before wait_staged() returns (page-locked
        buffers from parameter_array() make the copy truly asynchronous).  Without this call nothing changes: update!() uploads serially."""
        if not self.initialized:
            raise ErrorException("stage_parameters needs an initialized model (call initialize!(model) or solve!(model) once)")
        if not self._records:
            return
        from .lazyexpression import _stage_value
        from .parameter import DerivedParameter
        ctx = self.device()
        if len(ctx._keep_staged) > 64:           # the caller never waits: do not let the references to old host values pile up
            ctx.staged_synchronize()
        # alternate the staging slot: these values go into the buffers the update BEFORE the previous one used, so the copy does not wait
        # for the previous update's commits, only for those of the one before it
        ctx._pending_slot = ctx._next_stage_slot
        ctx._next_stage_slot ^= 1
        ctx.set_stage_slot(ctx._pending_slot)
        for x in self._order:
            if not isinstance(x, Parameter) or getattr(x, "device_resident", False) or isinstance(x, DerivedParameter) or x._dev is None:
                continue
            x.setdirty()
            val = Parameter.__call__(x)
            _stage_value(ctx, x._dev, val)
            x._staged_pending = True

    def wait_staged(self):
        """Block until the staged uploads have left the host buffers (which may then be overwritten for the solve after next)."""
        if self._ctx is not None:
            self._ctx.staged_synchronize()

    def addparameter(self, p):                                         # src/model.jl:42
        self.params.append(p)
        return p

    def _add_variable(self):                                           # src/model.jl:49-53
        if self.initialized:
            raise ErrorException("Model has already been initialized.")
        self.nvars += 1
        return self.nvars

    def setobjective(self, sense, expr):                               # src/model.jl:60-66
        if self.initialized:
            raise ErrorException("Model was already initialized. setobjective! can only be called before initialization.")
        if sense not in (Minimize, Maximize):
            raise ArgumentError("sense must be Minimize or Maximize")
        self.objective = moi.Objective(self, expr)
        self.sense = sense

    def add_constraint(self, c):                                       # src/model.jl:68-73
        if self.initialized:
            raise ErrorException("Model was already initialized. add_constraint can only be called before initialization.")
        c.modelindex = len(self.constraints)
        self.constraints.push(c)

    def _add(self, expr, scalar_set, vector_set):                      # src/model.jl:75-95
        kind = moi.canonical_function_kind(expr.out.kind if isinstance(expr, DeviceNode) else moi.kind_of(expr))
        if kind == "affvec":
            n = expr.out.rows if isinstance(expr, DeviceNode) else len(expr)
            self.add_constraint(moi.Constraint(self, expr, vector_set(n)))
        else:
            self.add_constraint(moi.Constraint(self, expr, scalar_set(0.0)))

    def add_nonnegative_constraint(self, expr): self._add(expr, moi.GreaterThan, moi.Nonnegatives)
    def add_nonpositive_constraint(self, expr): self._add(expr, moi.LessThan, moi.Nonpositives)
    def add_zero_constraint(self, expr): self._add(expr, moi.EqualTo, moi.Zeros)

    def add_integer_constraint(self, x):                               # src/model.jl:97
        self.add_constraint(moi.Constraint(self, None, moi.Integer(), function=moi.SingleVariable(x)))

    def add_binary_constraint(self, x):                                # src/model.jl:98
        self.add_constraint(moi.Constraint(self, None, moi.ZeroOne(), function=moi.SingleVariable(x)))

    # ---- initialize! / update! / solve!
    def initialize(self):                                              # src/model.jl:117-122
        backend = _Backend(self.nvars, self.sense, self.objective, list(self.constraints))
        records = [r for r in [self.objective] + list(self.constraints) if not r.isconstant]
        self._records = records
        early = self.handoff in ("device", "host_csc")
        if early:
            # device hand-off: the optimizer never sees host MOI functions, so the index map is fixed BEFORE the plan is recorded and
            # the plan can be specialised on it (P's CSC values straight from the contraction when the variable order is preserved)
            indexmap = self.optimizer.copy_to(backend)
            for c in self.constraints:
                c.optimizerindex = indexmap["constraints"][c]
            self.model_var_to_optimizer = np.asarray(indexmap["variables"], dtype=np.int64).copy()
        if records:
            ctx = self.device()
            # SMALL models (README Example 1): launch-bound on the device.  Their Parameter callbacks are recorded into the tape and the whole
            # update! replays as one small plan (csrc/small.hip); overlapped recorded fetches — a signal kernel and a copy-engine transfer per
            # MOI buffer, worth it for megabytes — would cut the run of small entries and cost more launches than the copies they hide
            def _elements(ps):
                return sum(int(getattr(getattr(p_, "val", None), "nnz", np.size(getattr(p_, "val", 0)))) for p_ in ps)
            self._small = (not self._use_graph and self.handoff == "moi" and
                           _elements(p_ for p_ in self.params if not getattr(p_, "device_resident", False)) <= self.SMALL_MODEL_ELEMENTS and
                           _elements(p_ for p_ in self.params if getattr(p_, "device_resident", False)) <= self.SMALL_MODEL_DEVICE_ELEMENTS)
            if self._small:
                self._overlap_moi = False
            self._varmap_buf = ctx.alloc(8 * max(self.nvars, 1))
            ident = np.arange(1, self.nvars + 1, dtype=np.int64)     # IdentityVarMap until mapindices! (src/moi_interop.jl:32-33)
            ctx.upload(self._varmap_buf, self.model_var_to_optimizer if early else ident)
            if self.quadratic_mode == "canonical":
                # any other quadratic objective: generic device canonicalize! (sorted, duplicates combined) before the MOI copy
                for r in records:
                    gram = getattr(r.expr, "gram_candidate", None)
                    if r.kind == "quad" and not (gram is not None and gram.xvars.strictly_increasing()):
                        r.expr = r.expr.canonicalize()
            emitters = [r.compile(ctx, self._varmap_buf, self.quadratic_mode, self.model_var_to_optimizer if early else None) for r in records]
            self._order = schedule([r.expr for r in records])
            for x in self._order:
                if isinstance(x, DeviceNode):
                    x.prepare()
            ctx.begin_record()
            try:
                self._record_parameter_callbacks(ctx)
                for x in self._order:
                    if isinstance(x, DeviceNode):
                        x.emit(ctx)
                # MOI copies of constraints built straight from their Parameters are independent of every other record (update! of one
                # Constraint, src/moi_interop.jl:168-175).  Beside a canonical least-squares objective they go to the plan's side lane:
                # queued behind the contraction's small reductions, they run while its workgroups drain and its fix-up pass runs,
                # instead of adding their kernels and in-stream gaps behind it.
                gram = any(getattr(r, "mode", "").startswith("canonical") and r.kind == "quad" and getattr(r.expr, "gram_candidate", None) is not None
                           for r in records)
                self._lane_records = []
                # one small kernel on the lane does not pay (config 2: the co-resident pack slows the contraction by what it saves); several
                # do (config 3: -0.15 ms), and so does the device hand-off, whose launches join them on the lane
                # (never in a SMALL model: its Parameter values arrive through entries at the FRONT of the tape — mailbox copies, seeded fills
                # — while side-lane entries fork at the top of the replay and would read the buffers before those entries have run; and a
                # lane would only cut the one-launch plan in pieces)
                eligible = [r for r in records if self._side_lane_ok(r)] if (gram and self._side_lane and not self._small) else []
                # (and with the overlapped MOI boundary a constraint on the lane is packed early: its terms cross PCIe during the contraction)
                use_lane = len(eligible) >= 2 or (len(eligible) >= 1 and (self.handoff != "moi" or self._overlap_moi))
                emit_order = list(zip(records, emitters))
                if self._small and gram:
                    # a SMALL model: the records are independent of each other, so the one whose MOI copy is not an interpreter node — the
                    # canonical least-squares objective beyond tiny shapes (gram_tall.hip: two launches) — goes last; the constraints' packs
                    # then join the run of small entries at the front of the tape (callbacks, residual) in its ONE launch
                    def is_gram(r):
                        return getattr(r, "mode", "").startswith("canonical") and r.kind == "quad" and getattr(r.expr, "gram_candidate", None) is not None
                    emit_order = [re for re in emit_order if not is_gram(re[0])] + [re for re in emit_order if is_gram(re[0])]
                elif use_lane:
                    # round 6c: the side lane's records go in FRONT of the objective — its one-launch node (gram_mid.hip) holds every CU
                    # with persistent workgroups and lane entries recorded behind it wait for it; recorded first they take CUs first and
                    # the node's workgroups start as CUs come free (config 3, staged uploads: 1.244 -> 1.222 ms per step)
                    emit_order = [re for re in emit_order if any(re[0] is x for x in eligible)] + [re for re in emit_order if not any(re[0] is x for x in eligible)]
                for r, e in emit_order:
                    side = use_lane and any(r is x for x in eligible)
                    if side:
                        ctx.set_lane(1)
                        self._lane_records.append(r)
                        r.on_side_lane = True
                    e(ctx)
                    if self._overlap_moi:
                        r.record_fetch(ctx)                    # behind its producers, on their lane
                    if side:
                        ctx.set_lane(0)
            finally:
                ctx.end_record()
            self._model_run = None
            self._create_model_run(ctx)
            # first evaluation with the identity map so that copy_to sees sized, filled functions (src/moi_interop.jl:127,157)
            self._run_tape()
        if not early:
            indexmap = self.optimizer.copy_to(backend)
            self._mapindices(indexmap)
        self.initialized = True
        if self.handoff != "moi":
            from .handoff import DeviceQP
            # When every constraint's MOI copy sits on the side lane and the objective is the Gram node (whose affine part is written on
            # the same side stream), the hand-off launches read side-stream outputs only: they are appended to the tape as side-lane
            # entries too and leave the plan's stream to the contraction.  Otherwise they are launched behind the tape on every update.
            obj = self.objective
            in_tape = bool(records) and (obj.isconstant or "P_values" in (obj.dev or {})) and \
                all(c.isconstant or any(c is r for r in getattr(self, "_lane_records", [])) for c in self.constraints) and \
                any(not c.isconstant for c in self.constraints)
            host = None if self.handoff == "device" else ("overlap" if self._overlap_fetch else "serial")
            self.device_qp = DeviceQP(self, in_tape="side" if in_tape else False, host=host)
        self._mark_side_lane_parameters()
        if records and self._use_graph:
            self.device().instantiate_graph()

    # measured crossover (tools/mid_table.py, solve! with host Parameters and a do-nothing optimizer): the small-model path (mailboxes, one or two
    # launches, MOI buffers stored straight into the function objects) beats uploads + separate kernels + overlapped fetches up to ~300 000
    # Parameter elements (n = 128, r = 240: 69 vs 156 us; n = 300, r = 800: 212 vs 229; n = 400, r = 800: equal)
    SMALL_MODEL_ELEMENTS = 262144
    # Parameters regenerated ON the device (DeviceUniformParameter): nothing to upload, so the separate-kernel path catches up earlier
    # (update! on the device, n = 128 / r = 240: 33 vs 40 us; n = 200 / r = 600: 75 vs 68)
    SMALL_MODEL_DEVICE_ELEMENTS = 65536

    def _record_parameter_callbacks(self, ctx):
        """SMALL models: the device-side callbacks of the DeviceUniformParameters go INTO the tape, at its front, with their seeds in host
        words (pmt_fill_uniform_dyn_f64) — update! then stores the next seeds and replays, and the library runs callbacks and tape as one
        small plan: one launch (csrc/small.hip).  README Example 1's update! is launch-bound on the device: four callbacks + five kernels
        took ~48 us where the reference's CPU walk of the same DAG takes ~15 (README.md:132-136)."""
        from .parameter import DeviceUniformParameter
        from .device import DMat, DVec
        if not getattr(self, "_small", False):
            return
        ps = [x for x in self._order if isinstance(x, Parameter)]
        from .device import DNum
        for x in ps:
            dv = getattr(x, "_dev", None)
            if not isinstance(x, DeviceUniformParameter) and isinstance(dv, (DMat, DVec, DNum)) and not getattr(x, "device_resident", False):
                # a HOST-updated Parameter (callback f(val) or Parameter(model, val=buf), src/parameter.jl:57,88) of a small model: its value
                # travels through a page-locked MAILBOX in the device layout, which the first entries of the tape copy into the Parameter's
                # buffer — inside the one small-plan launch, read straight from host memory.  update! writes the mailbox (a numpy copy of a
                # few hundred bytes) instead of issuing one hipMemcpyAsync per Parameter.
                n_doubles = (dv.lda * dv.cols) if isinstance(dv, DMat) else (dv.padded if isinstance(dv, DVec) else 1)
                if n_doubles <= 0:
                    continue
                x._mailbox = ctx.pinned_array(n_doubles, np.float64)
                x._mailbox[:] = 0.0
                x._mailbox_write = _mailbox_writer(x)
                _write_mailbox(x, x.val if getattr(x, "val", None) is not None else None)
                ctx.call("pmt_copy_bytes", C.c_void_p(dv.buf), C.c_void_p(x._mailbox.ctypes.data), 8 * n_doubles)
                x._in_tape = True
                continue
            if not isinstance(x, DeviceUniformParameter) or getattr(x, "pattern", None) is not None or not isinstance(dv, (DMat, DVec)):
                continue
            x._seed_word = C.c_uint64(x.current_seed() % (1 << 64))
            rows, cols, lda = (dv.rows, dv.cols, dv.lda) if isinstance(dv, DMat) else (int(x.shape[0]), 1, int(x.shape[0]))
            ctx.call("pmt_fill_uniform_dyn_f64", C.c_void_p(dv.buf), rows, cols, lda, C.byref(x._seed_word), x.scale)
            x._mailbox_write = None
            x._in_tape = True
        if ps and all(getattr(x, "_in_tape", False) for x in ps):
            self._tape_parameters = ps             # every value enters through the tape: _refresh_parameters' short walk

    def _create_model_run(self, ctx):
        """SMALL models: the per-solve walk behind ONE C call (pmt_model_update, csrc/modelrun.hip — the entry point a Julia / C host uses
        for the same walk): mailboxes of the host-updated Parameters whose value array is a float64 array that stays put (`val=` buffers,
        in-place callbacks) are registered with their strides — the library copies value -> mailbox itself —, the records' constants with
        the function objects' fields are finished there too.  Everything else keeps this host's own writers (dirty byte 0)."""
        fast = getattr(self, "_tape_parameters", None)
        if not getattr(self, "_small", False) or fast is None:
            return
        run = C.c_void_p()
        ctx.call("pmt_model_create", ctx.plan, C.byref(run))
        self._run_slot = []                                   # per Parameter of `fast`: slot in the C model, or -1 (written here)
        for x in fast:
            slot = -1
            mb, dv = getattr(x, "_mailbox", None), x._dev
            val = getattr(x, "val", None)
            if getattr(x, "_mailbox_write", None) is not None and isinstance(val, np.ndarray) and val.dtype == np.float64 and \
                    val.ndim in (1, 2) and all(st % 8 == 0 and st >= 0 for st in val.strides):
                from .device import DMat
                rows, cols = (dv.rows, dv.cols) if isinstance(dv, DMat) else (val.shape[0], 0)
                rs = val.strides[0] // 8
                cs = val.strides[1] // 8 if val.ndim == 2 else 0
                ld = dv.lda if isinstance(dv, DMat) else max(int(mb.size), rows)
                if val.shape == ((rows, cols) if cols else (rows,)):
                    out = C.c_int()
                    ctx.call("pmt_model_add_mailbox", run, C.c_void_p(val.ctypes.data), rows, cols, rs, cs, C.c_void_p(mb.ctypes.data), ld, C.byref(out))
                    slot = out.value
                    x._run_val = val                       # (identity of the array object: `val.ctypes.data` costs a microsecond per call)
            self._run_slot.append(slot)
        # what a record's fetch() would copy out of HBM (a buffer whose device twin IS the host array needs nothing): behind the replay, in C
        for r in self._records:
            for host, key in r.fetch_list():
                if key in r.dev and r.dev[key] != host.ctypes.data:
                    ctx.call("pmt_model_add_fetch", run, C.c_void_p(host.ctypes.data), C.c_void_p(r.dev[key]), host.nbytes)
        self._run_nslots = int(ctx.lib.pmt_model_num_slots(run))
        self._run_mask = (C.c_ubyte * max(self._run_nslots, 1))()
        self._model_run = run

    def _fast_update(self, ctx):
        """update!(model) of a small model, synchronous: callbacks here (they are host functions), everything else in pmt_model_update"""
        mask, slots = self._run_mask, self._run_slot
        synced = False
        for i, x in enumerate(self._tape_parameters):
            val = Parameter.__call__(x)                                  # evalarg(::Parameter) (src/lazyexpression.jl:51)
            slot = slots[i]
            if x._dev_version == x.version:
                if slot >= 0:
                    mask[slot] = 0
                continue
            x._dev_version = x.version
            write = x._mailbox_write
            if write is None:
                x._seed_word.value = x.current_seed() % (1 << 64)
            elif slot >= 0 and val is x._run_val:
                mask[slot] = 1                                           # the library copies value -> mailbox
            else:
                if slot >= 0:
                    mask[slot] = 0
                if not synced and getattr(ctx, "_replay_pending", False):
                    ctx.synchronize()                                   # the previous replay may still be reading the mailboxes
                synced = True
                if val is not None:
                    write(val)
        ctx.call("pmt_model_update", self._model_run, mask, self._run_nslots, 1)
        ctx._replay_pending = False
        for r in self._records:
            r.finish_fetch()

    def _mark_side_lane_parameters(self):
        """Host-updated Parameters that ONLY side-lane records read (and, with a hand-off, only when its launches are side-lane entries too)
        are committed on the side stream (pmt_plan_commit_lane): the upload of a constraint's data then overlaps the contraction of the
        objective instead of standing in front of it.  Every other Parameter keeps the plan's stream."""
        lane_records = getattr(self, "_lane_records", [])
        handoff_ok = self.device_qp is None or getattr(self.device_qp, "_in_tape_lane", None) == "side"
        for x, side_only in self._parameter_readers().values():
            # (a graph replay launches the side-lane entries as nodes of ONE graph on the plan's stream: no side stream to order against)
            x._commit_on_side_lane = bool(side_only and handoff_ok and lane_records and not self._use_graph)

    def _parameter_readers(self):
        """{id(Parameter): [Parameter, read by side-lane records only]}"""
        lane_records = getattr(self, "_lane_records", [])
        readers = {}
        for r in self._records:
            if not isinstance(r.expr, DeviceNode):
                continue
            on_side = any(r is x for x in lane_records)
            for x in schedule([r.expr]):
                if isinstance(x, Parameter):
                    readers.setdefault(id(x), [x, True])[1] &= on_side
        return readers

    def _side_refreshed_parameter_ids(self):
        """Parameters whose values will be produced ON the side stream (committed / regenerated there) once a hand-off recorded on the side
        lane exists: what a front-of-lane transfer may read without waiting for the plan's stream (lane 3)"""
        if self._use_graph or not getattr(self, "_lane_records", []):
            return set()
        return {k for k, (x, side_only) in self._parameter_readers().items() if side_only}

    @staticmethod
    def _side_lane_ok(r):
        from .device import DDenseAff, DSparseAff, DVarsAff
        if not getattr(r, "side_lane_ok", False) or not isinstance(r.expr, DeviceNode):
            return False
        for x in schedule([r.expr]):                # every node below the record is an implicit block: nothing of it is in the tape
            if isinstance(x, DeviceNode) and not (isinstance(x.out, (DDenseAff, DVarsAff, DSparseAff)) and not x.out.need_terms):
                return False
        return True

    def _mapindices(self, indexmap):                                   # src/model.jl:100-107
        for c in self.constraints:
            c.optimizerindex = indexmap["constraints"][c]
        self.model_var_to_optimizer = np.asarray(indexmap["variables"], dtype=np.int64).copy()
        if self._records:
            self.device().upload(self._varmap_buf, self.model_var_to_optimizer)
            for r in self._records:
                for hook in getattr(r, "varmap_hooks", ()):
                    hook(self.model_var_to_optimizer)

    def _refresh_parameters(self):
        ctx = self.device()
        fast = getattr(self, "_tape_parameters", None)
        if fast is not None and not ctx.recording and not any(getattr(x, "_staged_pending", False) for x in fast):
            # a SMALL model: every Parameter's value enters through an entry of the tape (mailbox copy or seeded fill) — the general walk
            # below (device_value_of: staging slots, lanes, uploads) reduces to "evaluate, write the mailbox / the seed word when it changed"
            synced = False
            for x in fast:
                val = Parameter.__call__(x)                              # evalarg(::Parameter) (src/lazyexpression.jl:51)
                if x._dev_version != x.version:
                    write = x._mailbox_write
                    if write is not None:
                        if not synced and getattr(ctx, "_replay_pending", False):
                            ctx.synchronize()                           # the previous replay may still be reading the mailboxes
                        synced = True
                        if val is not None:
                            write(val)
                    else:
                        x._seed_word.value = x.current_seed() % (1 << 64)
                    x._dev_version = x.version
            return
        from .lazyexpression import device_value_of
        ctx._staging_dirty = False
        if ctx._stage_slot != ctx._pending_slot:
            ctx.set_stage_slot(ctx._pending_slot)   # wait / commit / consumed act on the slot the pending values were staged into
        ctx._refreshing = True
        try:
            for x in self._order:
                if isinstance(x, Parameter):
                    device_value_of(x, ctx)
        finally:
            ctx._refreshing = False
        if ctx._staging_dirty:
            ctx.staging_consumed()               # behind the commits: the staging buffers may be overwritten by the next stage_parameters()
            ctx._staging_dirty = False

    def _run_tape(self, fetch=True):
        ctx = self.device()
        if getattr(self, "_fetches_read_parameters", False):
            # host_csc: dense constraint blocks leave straight out of their Parameter buffers (handoff.py): the previous solve's transfers
            # have read them before this solve's callbacks / commits rewrite them (a no-op when the caller has synchronised, as solve! does)
            ctx.fetch_synchronize()
        if fetch and getattr(self, "_model_run", None) is not None and not ctx.recording and \
                not any(getattr(x, "_staged_pending", False) for x in self._tape_parameters):
            self._fast_update(ctx)
            return
        self._refresh_parameters()
        ctx.replay()
        if not fetch:
            return
        for r in self._records:
            r.fetch(ctx)
        if self._overlap_moi:
            ctx.fetch_synchronize()                            # recorded fetches + the delivered quadratic terms have landed
        ctx.synchronize()
        for r in self._records:
            r.finish_fetch()

    def update(self, synchronize=True):                                # src/model.jl:132-143
        for p in self.params:                                          # setdirty!(model) — except what stage_parameters() has already
            if not getattr(p, "_staged_pending", False):               # evaluated for this solve
                p.setdirty()
        if self.device_qp is not None:
            # device hand-off: the MOI buffers never leave HBM; the solver's CSC data is rebuilt right behind the tape
            if self._records:
                self._run_tape(fetch=False)
            self.device_qp.refresh()
            if self.device_qp.host is not None:
                self.device_qp.host.host_copies()                      # blocks the host already has: copied here while the device works
            if synchronize:                                            # synchronize=False: the caller overlaps the next stage_parameters()
                if self.device_qp.host is not None:                    # with this re-evaluation and synchronises later
                    self.device_qp.host.wait()                         # host_csc: the solver's arrays have landed
                else:
                    self.device().synchronize()
            if self.device_qp.host is not None and hasattr(self.optimizer, "set_host_qp"):
                self.optimizer.set_host_qp(self.device_qp.host)        # OSQP: update(Px=, Ax=, q=, l=, u=) from these arrays
            elif hasattr(self.optimizer, "set_device_qp"):
                self.optimizer.set_device_qp(self.device_qp)
            return
        if self._records:
            self._run_tape()
        if not self.objective.isconstant:                              # src/moi_interop.jl:131-137
            self.optimizer.set_objective_function(self.objective.f)
        for c in self.constraints:                                     # src/moi_interop.jl:168-175, order :236-247
            if not c.isconstant:
                self.optimizer.set_constraint_function(c.optimizerindex, c.f)

    def solve(self):                                                   # src/model.jl:151-159
        if not self.initialized:
            self.initialize()
        self.update()
        self.optimizer.optimize()

    # ---- results (src/model.jl:166-194)
    def value(self, x):
        if isinstance(x, (list, tuple, np.ndarray)):
            return np.array([self.value(v) for v in x])
        return self.optimizer.variable_primal(int(self.model_var_to_optimizer[x.index - 1]))

    def objectivevalue(self): return self.optimizer.objective_value()
    def terminationstatus(self): return self.optimizer.termination_status()
    def primalstatus(self): return self.optimizer.primal_status()
    def dualstatus(self): return self.optimizer.dual_status()


# ---- function forms of the reference's exported names
def mock_model(**kw):                                                   # src/mockmodel.jl:3-6
    return Model(MockOptimizer(), **kw)


def setobjective(model, sense, expr): model.setobjective(sense, expr)
def initialize(model): model.initialize()
def update(model): model.update()
def solve(model): model.solve()
def value(model, x): return model.value(x)
def objectivevalue(model): return model.objectivevalue()
def terminationstatus(model): return model.terminationstatus()
def primalstatus(model): return model.primalstatus()
def dualstatus(model): return model.dualstatus()
def setdirty(x): x.setdirty()


def objective(model, sense, expr):
    """@objective(model, sense, expr) (src/model.jl:267-271)."""
    model.setobjective(sense, expr)


INTEGERS = ("ℤ", "Integers")
ZERO_ONE = ("{0, 1}", "ZeroOne")


def constraint(model, rel, op=None, rhs=None):
    """@constraint(model, lhs (>=|<=|==|in) rhs) (src/model.jl:224-249): always lhs - rhs in the matching cone."""
    if not isinstance(rel, Relation):
        if op is None:
            raise ArgumentError("Expected expression of the form `a relation b`")          # src/model.jl:226
        rel = Relation(rel, op, rhs)
    lhs, op, rhs = rel.lhs, rel.op, rel.rhs
    if op in ("in", "∈"):
        if rhs in INTEGERS:
            return model.add_integer_constraint(lhs)
        if rhs in ZERO_ONE:
            return model.add_binary_constraint(lhs)
        raise ArgumentError("'in' only supports ℤ/Integers and {0, 1}/ZeroOne")              # src/model.jl:243
    if op not in (">=", "<=", "=="):
        raise ArgumentError("Relation not recognized")                                        # src/model.jl:246
    expr = lazy("-", lhs, rhs)
    if op == ">=":
        model.add_nonnegative_constraint(expr)
    elif op == "<=":
        model.add_nonpositive_constraint(expr)
    else:
        model.add_zero_constraint(expr)
