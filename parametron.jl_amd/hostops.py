"""Constant folding on the host — setup time only.

`optimize_toplevel` evaluates an expression that involves no Parameter and no LazyExpression immediately
(src/lazyexpression.jl:189-192): the result is a plain value that the hot path never recomputes (`isconstant`,
src/moi_interop.jl:123,132).  These are the out-of-place array operations such constants need, restated with the
reference's term order on Python lists of the scalar types of functions.py.  Nothing here runs inside update!().
"""
import numpy as np

from ._lib import ArgumentError, DimensionMismatch
from .functions import AffineFunction, LinearTerm, QuadraticFunction, QuadraticTerm, Variable, _isnum


class Transpose:
    """transpose(x) / x' of a vector of Variables or functions (elements are their own transposes, src/functions.jl:635,645)."""

    def __init__(self, parent):
        self.parent = parent

    def __mul__(self, o):
        from .lazyexpression import lazy
        return lazy("*", self, o)

    __matmul__ = __mul__


def is_vector(v):
    return isinstance(v, (list, tuple)) or (isinstance(v, np.ndarray) and v.ndim == 1)


def elem_kind(v):
    """kind of a vector's elements: 'num' | 'var' | 'lt' | 'qt' | 'aff' | 'quad' | 'empty'."""
    if isinstance(v, np.ndarray):
        if v.dtype == object:
            v = list(v)
        else:
            return "num"
    if len(v) == 0:
        return "empty"
    kinds = set()
    for e in v:
        if _isnum(e):
            kinds.add("num")
        elif isinstance(e, Variable):
            kinds.add("var")
        elif isinstance(e, LinearTerm):
            kinds.add("lt")
        elif isinstance(e, QuadraticTerm):
            kinds.add("qt")
        elif isinstance(e, AffineFunction):
            kinds.add("aff")
        elif isinstance(e, QuadraticFunction):
            kinds.add("quad")
        else:
            raise ArgumentError("unsupported vector element %r" % (type(e).__name__,))
    if len(kinds) == 1:
        return kinds.pop()
    if kinds <= {"var", "lt", "num"}:
        return "lt" if "lt" in kinds else ("var" if kinds == {"var"} else "mixed")
    return "mixed"


def matvecmul(A, x):
    """A * x for a constant matrix: matvecmul! (src/functions.jl:775-822) out of place."""
    A = np.asarray(A, dtype=np.float64)
    rows, cols = A.shape
    if len(x) != cols:
        raise DimensionMismatch("matvecmul!: length(x) != size(A, 2)")
    k = elem_kind(x)
    out = [AffineFunction.zero() for _ in range(rows)]
    if k == "var":
        for r in range(rows):
            out[r].linear = [LinearTerm(float(A[r, c]), x[c]) for c in range(cols)]
    elif k == "aff":
        for c in range(cols):
            for r in range(rows):
                out[r].muladd(x[c], float(A[r, c]))
    else:
        raise ArgumentError("matrix * vector of %s is not supported" % k)
    return out


def vecaddsub(x, y, sign):
    """vecadd!/vecsubtract! (src/functions.jl:751-764) out of place: dest[i] = copyto!(zero, x[i]) (+|-) y[i]."""
    if len(x) != len(y):
        raise DimensionMismatch("vecadd!/vecsubtract!: lengths differ")
    out = []
    for a, b in zip(x, y):
        d = AffineFunction.of(float(a) if _isnum(a) else a)
        d = d.add(float(b) if _isnum(b) else b) if sign > 0 else d.subtract(float(b) if _isnum(b) else b)
        out.append(d)
    return out


def vecdot(x, y):
    """dot(x, y) on constant vectors: the vecdot!/_vecdot! method table (src/functions.jl:665-731, :931-956)."""
    if isinstance(x, np.ndarray) and x.ndim > 1:
        x = x.reshape(-1, order="F")        # column-major linear order (test/functions.jl:201-206)
    if isinstance(y, np.ndarray) and y.ndim > 1:
        y = y.reshape(-1, order="F")
    if len(x) != len(y):
        raise DimensionMismatch("dot: lengths differ")
    kx, ky = elem_kind(x), elem_kind(y)
    if kx == "num" and ky == "num":
        return float(np.dot(np.asarray(x, dtype=float), np.asarray(y, dtype=float)))
    lin = {"var", "lt"}
    if (kx == "num" and ky in lin | {"aff"}) or (ky == "num" and kx in lin | {"aff"}):
        dest = AffineFunction.zero()
        if "aff" in (kx, ky):                                       # :665-674
            for a, b in zip(x, y):
                dest.muladd(a if not _isnum(a) else float(a), b if not _isnum(b) else float(b))
        else:                                                        # :676-687
            dest.linear = [(float(a) * b) if _isnum(a) else (a * float(b)) for a, b in zip(x, y)]
        return dest
    if kx in lin | {"num", "qt"} and ky in lin | {"num", "qt"}:      # :689-700
        dest = QuadraticFunction.zero()
        dest.quadratic = [a * b for a, b in zip(x, y)]
        return dest
    dest = QuadraticFunction.zero()                                   # :702-709
    for a, b in zip(x, y):
        dest.muladd(a, b)
    return dest


def bilinearmul(Q, x, y):
    """transpose(x) * Q * y (src/functions.jl:840-858), including the Q[k]/(row, col) pairing of :849-856."""
    Q = np.asarray(Q, dtype=np.float64)
    if Q.shape != (len(x), len(y)):
        raise DimensionMismatch("bilinearmul!: size(Q) != (length(x), length(y))")
    flat = Q.reshape(-1, order="F")
    dest = QuadraticFunction.zero()
    k = 0
    for xr in x:
        for yc in y:
            dest.quadratic.append(QuadraticTerm(float(flat[k]), xr, yc))
            k += 1
    return dest


def scale(s, v):
    """scale! (src/functions.jl:873-925) out of place."""
    k = elem_kind(v)
    if k == "num":
        return float(s) * np.asarray(v, dtype=float)
    if k == "var":
        return [LinearTerm(float(s), e) for e in v]
    if k == "aff":
        return [AffineFunction.zero().muladd(e, float(s)) for e in v]
    raise ArgumentError("scale! of a vector of %s is not supported" % k)


def vcat(*vs):
    """vcat! (src/functions.jl:969-994) out of place."""
    out = []
    for v in vs:
        out.extend(AffineFunction.of(e) for e in v)
    return out


def _is_numeric(v):
    """a plain number array (matrix or vector of numbers), not a scalar"""
    if isinstance(v, np.ndarray):
        return v.dtype != object and v.ndim >= 1
    return isinstance(v, (list, tuple)) and len(v) > 0 and elem_kind(v) == "num"


def _num_array(v):
    return np.asarray(v, dtype=np.float64)


def _matmul(a, b):
    """Julia's `*` on number arrays: the matrix product, with Julia's errors (MethodError for vector*vector)."""
    if a.ndim == 1 and b.ndim == 1:
        raise ArgumentError("no method matching *(::Vector{Float64}, ::Vector{Float64}); use dot(x, y) or x' * y")
    if a.ndim == 1:                                     # Vector * Matrix: only the outer product with a one-row matrix exists
        if b.shape[0] != 1:
            raise DimensionMismatch("vector * matrix: the matrix must have one row, has %d" % b.shape[0])
        return np.outer(a, b[0])
    if a.shape[1] != b.shape[0]:
        raise DimensionMismatch("A has dimensions %r but B has dimensions %r" % (a.shape, b.shape))
    return a @ b


def apply(f, *args):
    """Evaluate a Parameter-free expression immediately (src/lazyexpression.jl:189-192)."""
    if f == "*":
        if len(args) == 3 and isinstance(args[0], Transpose):
            return bilinearmul(args[1], args[0].parent, args[2])
        a, b = args
        if isinstance(a, Transpose):
            if isinstance(b, np.ndarray) and b.ndim == 2:
                if _is_numeric(a.parent):                     # numbers: x' * Q is the row vector (Q' x)'
                    return Transpose(_matmul(np.ascontiguousarray(b.T), _num_array(a.parent)))
                return _RowTimesMatrix(a.parent, b)
            return vecdot(a.parent, b)
        if isinstance(a, _RowTimesMatrix):
            return bilinearmul(a.Q, a.x, b)
        # a 0-dimensional array is a scalar (Julia: a Number); scalar * numeric array of ANY rank is the elementwise scaling — `2 * A`
        # is valid Julia, the generic rule src/lazyexpression.jl:198 calls `*` out of place
        if isinstance(a, np.ndarray) and a.ndim == 0:
            a = a.item()
        if isinstance(b, np.ndarray) and b.ndim == 0:
            b = b.item()
        na, nb = _is_numeric(a), _is_numeric(b)
        if _isnum(a) and nb and isinstance(b, np.ndarray) and b.ndim >= 2:
            return float(a) * _num_array(b)
        if _isnum(b) and na and isinstance(a, np.ndarray) and a.ndim >= 2:
            return _num_array(a) * float(b)
        if na and nb:
            # plain numbers on both sides: Julia's `*` is the MATRIX product (never numpy's elementwise broadcast) —
            # Matrix*Matrix, Matrix*Vector; Vector*Vector has no method (the generic rule src/lazyexpression.jl:198 would throw)
            return _matmul(_num_array(a), _num_array(b))
        if isinstance(a, np.ndarray) and a.ndim == 2 and is_vector(b):
            return matvecmul(a, b)
        if _isnum(a) and is_vector(b):
            return scale(a, b)
        if _isnum(b) and is_vector(a):
            return scale(b, a)
        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
            raise ArgumentError("no method matching *(%s, %s)" % (type(a).__name__, type(b).__name__))
        return a * b
    if f in ("+", "-"):
        a, b = args
        if is_vector(a) and is_vector(b):
            if elem_kind(a) == "num" and elem_kind(b) == "num":
                return np.asarray(a, dtype=float) + np.asarray(b, dtype=float) if f == "+" else np.asarray(a, dtype=float) - np.asarray(b, dtype=float)
            return vecaddsub(a, b, +1 if f == "+" else -1)
        return a + b if f == "+" else a - b
    if f == "dot":
        a, b = args
        if is_vector(a) or isinstance(a, np.ndarray):
            return vecdot(a, b)
        return a * b
    if f == "vcat":
        return vcat(*args)
    if f == "vect":
        return list(args)
    if f == "adjoint":
        (a,) = args
        if isinstance(a, np.ndarray) and a.ndim == 2:
            return np.ascontiguousarray(a.T)
        return Transpose(a)
    if f == "convert":
        return args[-1]
    if f == "identity":
        return args[0]
    raise ArgumentError("unsupported constant expression %r" % (f,))


class _RowTimesMatrix:
    """transpose(x) * Q waiting for its right factor (Julia parses x' * Q * y as one 3-argument call)."""

    def __init__(self, x, Q):
        self.x, self.Q = x, Q

    def __mul__(self, o):
        from .lazyexpression import lazy
        return lazy("*", self, o)

    __matmul__ = __mul__
