"""ctypes front-end for the CPU oracle (oracle/parametron_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package (parametron.jl_amd).

The C file restates the reference's Julia loops (each function cites the reference
file:line); this module only marshals numpy arrays in and out.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libparametron_oracle.so")

# Julia isbits layouts
LT = np.dtype([("coeff", "<f8"), ("var", "<i8")])                     # LinearTerm / MOI.ScalarAffineTerm
QT = np.dtype([("coeff", "<f8"), ("row", "<i8"), ("col", "<i8")])     # QuadraticTerm / MOI.ScalarQuadraticTerm
VAT = np.dtype([("out", "<i8"), ("coeff", "<f8"), ("var", "<i8")])    # MOI.VectorAffineTerm

OK, DIMENSION_MISMATCH, ARGUMENT_ERROR = 0, 1, 2


class DimensionMismatch(Exception):
    """Julia DimensionMismatch (thrown under @boundscheck in src/functions.jl)."""


def build(force=False):
    src = os.path.join(HERE, "parametron_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i64, f64, ci = C.c_void_p, C.c_int64, C.c_double, C.c_int
        sig = {
            "pmo_aff_new": (vp, []), "pmo_aff_free": (None, [vp]),
            "pmo_quad_new": (vp, []), "pmo_quad_free": (None, [vp]),
            "pmo_affvec_new": (vp, [i64]), "pmo_affvec_free": (None, [vp, i64]),
            "pmo_affvec_at": (vp, [vp, i64]),
            "pmo_aff_nterms": (i64, [vp]), "pmo_aff_constant": (f64, [vp]), "pmo_aff_terms": (vp, [vp]),
            "pmo_quad_nterms": (i64, [vp]), "pmo_quad_terms": (vp, [vp]), "pmo_quad_affine": (vp, [vp]),
            "pmo_aff_zero": (None, [vp]), "pmo_quad_zero": (None, [vp]),
            "pmo_aff_push": (None, [vp, f64, i64]), "pmo_aff_set_constant": (None, [vp, f64]),
            "pmo_quad_push": (None, [vp, f64, i64, i64]),
            "pmo_aff_copy_number": (None, [vp, f64]), "pmo_aff_copy_term": (None, [vp, f64, i64]),
            "pmo_aff_copy_var": (None, [vp, i64]), "pmo_aff_copy": (None, [vp, vp]),
            "pmo_quad_copy_aff": (None, [vp, vp]), "pmo_quad_copy": (None, [vp, vp]),
            "pmo_aff_add_number": (None, [vp, f64]), "pmo_aff_add_term": (None, [vp, f64, i64]),
            "pmo_aff_add_var": (None, [vp, i64]), "pmo_aff_add_aff": (None, [vp, vp]),
            "pmo_quad_add_term": (None, [vp, f64, i64, i64]), "pmo_quad_add_quad": (None, [vp, vp]),
            "pmo_aff_sub_number": (None, [vp, f64]), "pmo_aff_sub_term": (None, [vp, f64, i64]),
            "pmo_aff_sub_var": (None, [vp, i64]), "pmo_aff_sub_aff": (None, [vp, vp]),
            "pmo_quad_sub_quad": (None, [vp, vp]),
            "pmo_aff_muladd_aff_number": (None, [vp, vp, f64]),
            "pmo_quad_muladd_quad_number": (None, [vp, vp, f64]),
            "pmo_quad_muladd_aff_term": (None, [vp, vp, f64, i64, ci]),
            "pmo_quad_muladd_aff_aff": (None, [vp, vp, vp]),
            "pmo_aff_mul_aff_number": (None, [vp, vp, f64]),
            "pmo_quad_mul_quad_number": (None, [vp, vp, f64]),
            "pmo_quad_mul_aff_term": (None, [vp, vp, f64, i64, ci]),
            "pmo_quad_mul_aff_aff": (None, [vp, vp, vp]),
            "pmo_vecdot_aff_numbers_affs": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecdot_aff_numbers_vars": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecdot_quad_vars_vars": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecdot_quad_terms_terms": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecdot_quad_affs_vars": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecdot_quad_affs_affs": (ci, [vp, vp, i64, vp, i64]),
            "pmo_vecaddsub_affs_numbers": (ci, [vp, vp, i64, vp, i64, ci]),
            "pmo_vecaddsub_vars_numbers": (ci, [vp, vp, i64, vp, i64, ci]),
            "pmo_vecaddsub_numbers_affs": (ci, [vp, vp, i64, vp, i64, ci]),
            "pmo_vecaddsub_affs_affs": (ci, [vp, vp, i64, vp, i64, ci]),
            "pmo_vecaddsub_affs_vars": (ci, [vp, vp, i64, vp, i64, ci]),
            "pmo_matvecmul_vars": (ci, [vp, i64, vp, i64, i64, vp, i64]),
            "pmo_matvecmul_affs": (ci, [vp, i64, vp, i64, i64, vp, i64]),
            "pmo_bilinearmul": (ci, [vp, vp, i64, i64, vp, i64, vp, i64]),
            "pmo_scale_number_vars": (ci, [vp, i64, f64, vp, i64]),
            "pmo_scale_number_affs": (ci, [vp, i64, f64, vp, i64]),
            "pmo_scale_number_numbers": (ci, [vp, i64, f64, vp, i64]),
            "pmo_vcat": (ci, [vp, i64, vp, vp, i64]),
            "pmo_aff_canonicalize": (None, [vp]), "pmo_quad_canonicalize": (None, [vp]),
            "pmo_aff_prune_zero": (None, [vp, f64]), "pmo_quad_prune_zero": (None, [vp, f64]),
            "pmo_aff_eval": (f64, [vp, vp]), "pmo_quad_eval": (f64, [vp, vp]),
            "pmo_moi_scalar_affine": (None, [vp, vp, vp, vp]),
            "pmo_moi_scalar_quadratic": (None, [vp, vp, vp, vp, vp]),
            "pmo_moi_vector_affine_nterms": (i64, [vp, i64]),
            "pmo_moi_vector_affine": (None, [vp, i64, vp, vp, vp]),
            "pmo_lsq_new": (vp, [i64, i64, i64]), "pmo_lsq_free": (None, [vp]),
            "pmo_lsq_residual": (vp, [vp]), "pmo_lsq_constraint": (vp, [vp]), "pmo_lsq_objective": (vp, [vp]),
            "pmo_lsq_eval_residual": (ci, [vp, vp, vp, vp]),
            "pmo_lsq_eval_constraint": (ci, [vp, vp, vp, vp]),
            "pmo_lsq_eval_objective": (ci, [vp, vp, vp, vp, ci, i64]),
            "pmo_lsq_eval_vecdot": (ci, [vp, i64]),
            "pmo_fill_uniform": (None, [vp, i64, C.c_uint64, f64]),
            "pmo_canonical_quad_samples": (ci, [vp, i64, i64, i64, vp, vp, i64, vp, vp]),
            "pmo_canonical_lin_samples": (ci, [vp, i64, i64, i64, vp, ci, vp, i64, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc == DIMENSION_MISMATCH:
        raise DimensionMismatch()
    if rc != OK:
        raise ValueError("oracle error %d" % rc)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _colmajor(A):
    """Julia Matrix{Float64}: column-major; returns a 1-D buffer in Julia linear-index order."""
    A = np.asarray(A, dtype=np.float64)
    return np.ascontiguousarray(A.T).reshape(-1), A.shape[0], A.shape[1]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _AffView:
    """Non-owning view of a pmo_aff (AffineFunction{Float64})."""

    def __init__(self, handle, owner=None):
        self.h = handle
        self._owner = owner

    @property
    def nterms(self):
        return lib().pmo_aff_nterms(self.h)

    @property
    def constant(self):
        return lib().pmo_aff_constant(self.h)

    def terms(self):
        n = self.nterms
        out = np.empty(n, dtype=LT)
        if n:
            C.memmove(out.ctypes.data, lib().pmo_aff_terms(self.h), n * LT.itemsize)
        return out

    def as_tuple(self):
        """([(coeff, var), ...], constant) — ordered, like Julia's `==` on AffineFunction."""
        t = self.terms()
        return [(float(c), int(v)) for c, v in zip(t["coeff"], t["var"])], float(self.constant)

    # in-place builders (mutate self)
    def zero(self): lib().pmo_aff_zero(self.h); return self
    def push(self, coeff, var): lib().pmo_aff_push(self.h, coeff, var); return self
    def set_constant(self, c): lib().pmo_aff_set_constant(self.h, c); return self
    def copy_from(self, x): lib().pmo_aff_copy(self.h, x.h); return self
    def add_number(self, x): lib().pmo_aff_add_number(self.h, x); return self
    def add_term(self, c, v): lib().pmo_aff_add_term(self.h, c, v); return self
    def add_var(self, v): lib().pmo_aff_add_var(self.h, v); return self
    def add_aff(self, x): lib().pmo_aff_add_aff(self.h, x.h); return self
    def sub_number(self, x): lib().pmo_aff_sub_number(self.h, x); return self
    def sub_term(self, c, v): lib().pmo_aff_sub_term(self.h, c, v); return self
    def sub_var(self, v): lib().pmo_aff_sub_var(self.h, v); return self
    def sub_aff(self, x): lib().pmo_aff_sub_aff(self.h, x.h); return self
    def muladd_aff_number(self, x, y): lib().pmo_aff_muladd_aff_number(self.h, x.h, y); return self
    def mul_aff_number(self, x, y): lib().pmo_aff_mul_aff_number(self.h, x.h, y); return self
    def canonicalize(self): lib().pmo_aff_canonicalize(self.h); return self
    def prune_zero(self, atol=0.0): lib().pmo_aff_prune_zero(self.h, atol); return self

    def eval(self, vals):
        v = _f64(vals)
        return lib().pmo_aff_eval(self.h, _ptr(v))


class Aff(_AffView):
    """Owning AffineFunction{Float64}."""

    def __init__(self, terms=(), constant=0.0):
        super().__init__(lib().pmo_aff_new())
        for c, v in terms:
            self.push(c, v)
        self.set_constant(constant)

    def __del__(self):
        try:
            lib().pmo_aff_free(self.h)
        except Exception:
            pass


class Quad:
    """Owning QuadraticFunction{Float64}."""

    def __init__(self, quad=(), linear=(), constant=0.0):
        self.h = lib().pmo_quad_new()
        for c, r, cl in quad:
            lib().pmo_quad_push(self.h, c, r, cl)
        a = self.affine
        for c, v in linear:
            a.push(c, v)
        a.set_constant(constant)

    def __del__(self):
        try:
            lib().pmo_quad_free(self.h)
        except Exception:
            pass

    @property
    def affine(self):
        return _AffView(lib().pmo_quad_affine(self.h), owner=self)

    @property
    def nterms(self):
        return lib().pmo_quad_nterms(self.h)

    def terms(self):
        n = self.nterms
        out = np.empty(n, dtype=QT)
        if n:
            C.memmove(out.ctypes.data, lib().pmo_quad_terms(self.h), n * QT.itemsize)
        return out

    def as_tuple(self):
        t = self.terms()
        q = [(float(c), int(r), int(cl)) for c, r, cl in zip(t["coeff"], t["row"], t["col"])]
        lin, const = self.affine.as_tuple()
        return q, lin, const

    def zero(self): lib().pmo_quad_zero(self.h); return self
    def copy_from(self, x): lib().pmo_quad_copy(self.h, x.h); return self
    def copy_from_aff(self, x): lib().pmo_quad_copy_aff(self.h, x.h); return self
    def add_term(self, c, r, cl): lib().pmo_quad_add_term(self.h, c, r, cl); return self
    def add_quad(self, x): lib().pmo_quad_add_quad(self.h, x.h); return self
    def sub_quad(self, x): lib().pmo_quad_sub_quad(self.h, x.h); return self
    def muladd_quad_number(self, x, y): lib().pmo_quad_muladd_quad_number(self.h, x.h, y); return self
    def muladd_aff_var(self, x, var): lib().pmo_quad_muladd_aff_term(self.h, x.h, 1.0, var, 1); return self
    def muladd_aff_term(self, x, c, var): lib().pmo_quad_muladd_aff_term(self.h, x.h, c, var, 0); return self
    def muladd_aff_aff(self, x, y): lib().pmo_quad_muladd_aff_aff(self.h, x.h, y.h); return self
    def mul_quad_number(self, x, y): lib().pmo_quad_mul_quad_number(self.h, x.h, y); return self
    def mul_aff_var(self, x, var): lib().pmo_quad_mul_aff_term(self.h, x.h, 1.0, var, 1); return self
    def mul_aff_term(self, x, c, var): lib().pmo_quad_mul_aff_term(self.h, x.h, c, var, 0); return self
    def mul_aff_aff(self, x, y): lib().pmo_quad_mul_aff_aff(self.h, x.h, y.h); return self
    def canonicalize(self): lib().pmo_quad_canonicalize(self.h); return self
    def prune_zero(self, atol=0.0): lib().pmo_quad_prune_zero(self.h, atol); return self

    def eval(self, vals):
        v = _f64(vals)
        return lib().pmo_quad_eval(self.h, _ptr(v))

    # vecdot! forms (dest = self)
    def vecdot_vars_vars(self, x, y):
        x, y = _i64(x), _i64(y)
        _check(lib().pmo_vecdot_quad_vars_vars(self.h, _ptr(x), len(x), _ptr(y), len(y))); return self

    def vecdot_terms_terms(self, x, y):
        """x, y: arrays of (coeff, var); a bare Variable is (1.0, var)."""
        x = np.ascontiguousarray(np.array([tuple(t) for t in x], dtype=LT))
        y = np.ascontiguousarray(np.array([tuple(t) for t in y], dtype=LT))
        _check(lib().pmo_vecdot_quad_terms_terms(self.h, _ptr(x), len(x), _ptr(y), len(y))); return self

    def vecdot_affs_vars(self, xs, y):
        y = _i64(y)
        _check(lib().pmo_vecdot_quad_affs_vars(self.h, xs.h, len(xs), _ptr(y), len(y))); return self

    def vecdot_affs_affs(self, xs, ys):
        _check(lib().pmo_vecdot_quad_affs_affs(self.h, xs.h, len(xs), ys.h, len(ys))); return self

    def bilinearmul(self, Q, x, y):
        q, rows, cols = _colmajor(Q)
        x, y = _i64(x), _i64(y)
        _check(lib().pmo_bilinearmul(self.h, _ptr(q), rows, cols, _ptr(x), len(x), _ptr(y), len(y))); return self

    def moi(self, varmap=None):
        """update!(::MOI.ScalarQuadraticFunction, f, varmap) -> (affine_terms, quadratic_terms, constant)."""
        vmap = None if varmap is None else _i64(varmap)
        at = np.empty(self.affine.nterms, dtype=LT)
        qt = np.empty(self.nterms, dtype=QT)
        const = C.c_double()
        lib().pmo_moi_scalar_quadratic(self.h, None if vmap is None else _ptr(vmap), _ptr(at), _ptr(qt), C.byref(const))
        return at, qt, const.value


class AffVec:
    """Owning Vector{AffineFunction{Float64}}."""

    def __init__(self, n, _handle=None, _own=True):
        self.n = int(n)
        self._own = _own
        self.h = lib().pmo_affvec_new(self.n) if _handle is None else _handle

    def __del__(self):
        try:
            if self._own:
                lib().pmo_affvec_free(self.h, self.n)
        except Exception:
            pass

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError(i)
        return _AffView(lib().pmo_affvec_at(self.h, i), owner=self)

    def as_tuples(self):
        return [self[i].as_tuple() for i in range(self.n)]

    def flat(self):
        """(terms[LT] row-major concatenation, row_ptr, constants)."""
        counts = [self[i].nterms for i in range(self.n)]
        row_ptr = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum(counts, out=row_ptr[1:])
        terms = np.concatenate([self[i].terms() for i in range(self.n)]) if self.n else np.empty(0, dtype=LT)
        consts = np.array([self[i].constant for i in range(self.n)], dtype=np.float64)
        return terms, row_ptr, consts

    # builders (dest = self)
    def matvecmul_vars(self, A, x):
        a, rows, cols = _colmajor(A)
        x = _i64(x)
        _check(lib().pmo_matvecmul_vars(self.h, self.n, _ptr(a), rows, cols, _ptr(x), len(x))); return self

    def matvecmul_affs(self, A, xs):
        a, rows, cols = _colmajor(A)
        _check(lib().pmo_matvecmul_affs(self.h, self.n, _ptr(a), rows, cols, xs.h, len(xs))); return self

    def _resized(self, n):
        # vecadd!/vecsubtract! resize dest to length(x) when it differs (functions.jl:756)
        if n != self.n:
            lib().pmo_affvec_free(self.h, self.n)
            self.n = n
            self.h = lib().pmo_affvec_new(n)

    def vecaddsub(self, x, y, subtract):
        kinds = (_kind(x), _kind(y))
        sub = 1 if subtract else 0
        L = lib()
        nx = len(x)
        if len(y) == nx:
            self._resized(nx)
        if kinds == ("affs", "numbers"):
            y = _f64(y); rc = L.pmo_vecaddsub_affs_numbers(self.h, x.h, nx, _ptr(y), len(y), sub)
        elif kinds == ("vars", "numbers"):
            x = _i64(x); y = _f64(y); rc = L.pmo_vecaddsub_vars_numbers(self.h, _ptr(x), nx, _ptr(y), len(y), sub)
        elif kinds == ("numbers", "affs"):
            x = _f64(x); rc = L.pmo_vecaddsub_numbers_affs(self.h, _ptr(x), nx, y.h, len(y), sub)
        elif kinds == ("affs", "affs"):
            rc = L.pmo_vecaddsub_affs_affs(self.h, x.h, nx, y.h, len(y), sub)
        elif kinds == ("affs", "vars"):
            y = _i64(y); rc = L.pmo_vecaddsub_affs_vars(self.h, x.h, nx, _ptr(y), len(y), sub)
        else:
            raise TypeError(kinds)
        _check(rc); return self

    def vecadd(self, x, y): return self.vecaddsub(x, y, False)
    def vecsubtract(self, x, y): return self.vecaddsub(x, y, True)

    def scale_number_affs(self, s, ys):
        _check(lib().pmo_scale_number_affs(self.h, self.n, s, ys.h, len(ys))); return self

    def vcat(self, *srcs):
        arr = (C.c_void_p * len(srcs))(*[s.h for s in srcs])
        lens = _i64([len(s) for s in srcs])
        _check(lib().pmo_vcat(self.h, self.n, arr, _ptr(lens), len(srcs))); return self

    def moi(self, varmap=None):
        """update!(::MOI.VectorAffineFunction, fs, varmap) -> (terms[VAT], constants)."""
        vmap = None if varmap is None else _i64(varmap)
        nt = lib().pmo_moi_vector_affine_nterms(self.h, self.n)
        terms = np.empty(nt, dtype=VAT)
        consts = np.empty(self.n, dtype=np.float64)
        lib().pmo_moi_vector_affine(self.h, self.n, None if vmap is None else _ptr(vmap), _ptr(terms), _ptr(consts))
        return terms, consts


def _kind(v):
    if isinstance(v, AffVec):
        return "affs"
    a = np.asarray(v)
    return "vars" if a.dtype.kind in "iu" else "numbers"


def aff_moi(f, varmap=None):
    """update!(::MOI.ScalarAffineFunction, f, varmap) -> (terms[LT], constant)."""
    vmap = None if varmap is None else _i64(varmap)
    t = np.empty(f.nterms, dtype=LT)
    const = C.c_double()
    lib().pmo_moi_scalar_affine(f.h, None if vmap is None else _ptr(vmap), _ptr(t), C.byref(const))
    return t, const.value


def vecdot_aff_numbers_vars(x, y):
    dest = Aff()
    x, y = _f64(x), _i64(y)
    _check(lib().pmo_vecdot_aff_numbers_vars(dest.h, _ptr(x), len(x), _ptr(y), len(y)))
    return dest


def vecdot_aff_numbers_affs(x, ys):
    dest = Aff()
    x = _f64(x)
    _check(lib().pmo_vecdot_aff_numbers_affs(dest.h, _ptr(x), len(x), ys.h, len(ys)))
    return dest


def scale_number_vars(s, y):
    y = _i64(y)
    dest = np.empty(len(y), dtype=LT)
    _check(lib().pmo_scale_number_vars(_ptr(dest), len(dest), s, _ptr(y), len(y)))
    return dest


def fill_uniform(n, seed, scale=1.0):
    """Counter-based U[0,1)*scale stream shared bit-for-bit with the device fill kernel."""
    out = np.empty(int(n), dtype=np.float64)
    lib().pmo_fill_uniform(_ptr(out), out.size, C.c_uint64(seed), scale)
    return out


def canonical_quad_samples(A_colmajor, lda, rows, cols, pj, pk):
    """MOI coefficients of the canonical quadratic terms at the sampled 1-based pairs (pj[i] <= pk[i]): (double-precision literal sum,
    long-double sum of the same products) — see pmo_canonical_quad_samples."""
    pj = np.ascontiguousarray(pj, dtype=np.int64)
    pk = np.ascontiguousarray(pk, dtype=np.int64)
    out, out_ld = np.empty(len(pj)), np.empty(len(pj))
    _check(lib().pmo_canonical_quad_samples(_ptr(A_colmajor), int(lda), int(rows), int(cols), _ptr(pj), _ptr(pk), len(pj), _ptr(out), _ptr(out_ld)))
    return out, out_ld


def canonical_lin_samples(A_colmajor, lda, rows, cols, b, sign, pj):
    """MOI coefficients of the canonical affine terms at the sampled 1-based variables: (double sum, long-double sum)."""
    pj = np.ascontiguousarray(pj, dtype=np.int64)
    out, out_ld = np.empty(len(pj)), np.empty(len(pj))
    _check(lib().pmo_canonical_lin_samples(_ptr(A_colmajor), int(lda), int(rows), int(cols), _ptr(b), int(sign), _ptr(pj), len(pj), _ptr(out), _ptr(out_ld)))
    return out, out_ld


class LsqWorkspace:
    """README Example 1 restated end to end: residual = A*x - b; residual . residual; C*x - d.

    (README.md:23-57; call stack SURVEY.md §3.2).  All buffers persist across calls, so
    repeated `update` calls reproduce the reference's zero-allocation steady state.
    """

    def __init__(self, n, r, m):
        self.n, self.r, self.m = int(n), int(r), int(m)
        self.h = lib().pmo_lsq_new(self.n, self.r, self.m)
        self.residual = AffVec(self.r, _handle=lib().pmo_lsq_residual(self.h), _own=False)
        self.constraint = AffVec(self.m, _handle=lib().pmo_lsq_constraint(self.h), _own=False)
        self.objective = Quad.__new__(Quad)
        self.objective.h = lib().pmo_lsq_objective(self.h)
        self.objective.__class__ = _BorrowedQuad

    def __del__(self):
        try:
            lib().pmo_lsq_free(self.h)
        except Exception:
            pass

    def eval_residual(self, A_colmajor, b, xvar):
        _check(lib().pmo_lsq_eval_residual(self.h, _ptr(A_colmajor), _ptr(b), _ptr(xvar)))

    def eval_constraint(self, C_colmajor, d, xvar):
        _check(lib().pmo_lsq_eval_constraint(self.h, _ptr(C_colmajor), _ptr(d), _ptr(xvar)))

    def eval_objective(self, A_colmajor, b, xvar, twice=True, rows_limit=-1):
        _check(lib().pmo_lsq_eval_objective(self.h, _ptr(A_colmajor), _ptr(b), _ptr(xvar), 1 if twice else 0, rows_limit))


    def eval_vecdot(self, rows_limit=-1):
        _check(lib().pmo_lsq_eval_vecdot(self.h, rows_limit))


class _BorrowedQuad(Quad):
    def __del__(self):
        pass
