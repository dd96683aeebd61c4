"""Host-side value types of Parametron.Functions (src/functions.jl:37-1001), used at SETUP time only.

The reference builds constant (Parameter-free) pieces of a model with ordinary operator overloading at
macro-expansion time (`optimize_toplevel` evaluates them once, src/lazyexpression.jl:189-192) and never
touches them again on the hot path (`isconstant`, src/moi_interop.jl:123,132).  These classes are that
scalar algebra — Variable, LinearTerm, QuadraticTerm, AffineFunction, QuadraticFunction with the
reference's term ORDER — plus the value types returned when a device node is fetched (`expr()`).
Everything evaluated per solve!() runs in the HIP kernels behind include/parametron_hip.h.
"""
import numbers

import numpy as np

from ._lib import LT, QT, ArgumentError, DimensionMismatch


def _isnum(x):
    return isinstance(x, (numbers.Real, np.floating, np.integer)) and not isinstance(x, bool)


class Variable:
    """A single decision variable (src/functions.jl:96-98); index is 1-based."""

    __slots__ = ("index",)
    __array_priority__ = 1000

    def __init__(self, index):
        if not isinstance(index, (int, np.integer)):
            # Variable(model): src/model.jl:49-53
            index = index._add_variable()
        self.index = int(index)

    def __hash__(self):
        return hash(("Variable", self.index))

    def __eq__(self, other):
        return isinstance(other, Variable) and other.index == self.index

    def __repr__(self):
        return "x%d" % self.index

    def __pos__(self):
        return self

    def __neg__(self):
        return LinearTerm(-1, self)                                  # :122

    def __mul__(self, o):
        if _isnum(o):
            return LinearTerm(o, self)                               # :121
        if isinstance(o, Variable):
            return QuadraticTerm(1, self, o)                         # :148
        if isinstance(o, LinearTerm):
            return QuadraticTerm(o.coeff, self, o.var)               # :146
        if isinstance(o, AffineFunction):
            return QuadraticFunction.zero().muladd(o, self)          # :619 -> :546
        return NotImplemented

    def __rmul__(self, o):
        if _isnum(o):
            return LinearTerm(o, self)                               # :120
        return NotImplemented

    def __add__(self, o):
        return _addsub(self, o, +1)

    def __radd__(self, o):
        return _addsub(o, self, +1)

    def __sub__(self, o):
        return _addsub(self, o, -1)

    def __rsub__(self, o):
        return _addsub(o, self, -1)

    def __pow__(self, p):
        return _power_by_squaring(self, p)


class LinearTerm:
    """coeff * var (src/functions.jl:110-113)."""

    __slots__ = ("coeff", "var")
    __array_priority__ = 1000

    def __init__(self, coeff, var):
        self.coeff = coeff
        self.var = var

    def __eq__(self, o):
        return isinstance(o, LinearTerm) and self.coeff == o.coeff and self.var == o.var

    def __hash__(self):
        return hash((self.coeff, self.var.index))

    def __repr__(self):
        return "%s * x%d" % (_fmt(self.coeff), self.var.index)       # :117

    def __pos__(self):
        return self

    def __neg__(self):
        return LinearTerm(-self.coeff, self.var)                     # :158

    def __mul__(self, o):
        if _isnum(o):
            return LinearTerm(o * self.coeff, self.var)              # :160 -> :159
        if isinstance(o, Variable):
            return QuadraticTerm(self.coeff, self.var, o)            # :147
        if isinstance(o, LinearTerm):
            return QuadraticTerm(self.coeff * o.coeff, self.var, o.var)   # :149
        if isinstance(o, AffineFunction):
            return QuadraticFunction.zero().muladd(o, self)          # :621 -> :546
        return NotImplemented

    def __rmul__(self, o):
        if _isnum(o):
            return LinearTerm(o * self.coeff, self.var)              # :159
        return NotImplemented

    def __add__(self, o):
        return _addsub(self, o, +1)

    def __radd__(self, o):
        return _addsub(o, self, +1)

    def __sub__(self, o):
        return _addsub(self, o, -1)

    def __rsub__(self, o):
        return _addsub(o, self, -1)

    def __pow__(self, p):
        return _power_by_squaring(self, p)


class QuadraticTerm:
    """coeff * rowvar * colvar (src/functions.jl:136-140)."""

    __slots__ = ("coeff", "rowvar", "colvar")
    __array_priority__ = 1000

    def __init__(self, coeff, rowvar, colvar):
        self.coeff, self.rowvar, self.colvar = coeff, rowvar, colvar

    def __eq__(self, o):
        return isinstance(o, QuadraticTerm) and (self.coeff, self.rowvar, self.colvar) == (o.coeff, o.rowvar, o.colvar)

    def __hash__(self):
        return hash((self.coeff, self.rowvar.index, self.colvar.index))

    def __repr__(self):
        return "%s * x%d * x%d" % (_fmt(self.coeff), self.rowvar.index, self.colvar.index)   # :143

    def __pos__(self):
        return self

    def __neg__(self):
        return QuadraticTerm(-self.coeff, self.rowvar, self.colvar)

    def __mul__(self, o):
        if _isnum(o):
            return QuadraticTerm(o * self.coeff, self.rowvar, self.colvar)
        return NotImplemented

    __rmul__ = __mul__

    def __add__(self, o):
        return _addsub(self, o, +1)

    def __radd__(self, o):
        return _addsub(o, self, +1)

    def __sub__(self, o):
        return _addsub(self, o, -1)

    def __rsub__(self, o):
        return _addsub(o, self, -1)

    def canonicalize(self):                                           # :182-184
        a, b = self.rowvar.index, self.colvar.index
        return QuadraticTerm(self.coeff, Variable(min(a, b)), Variable(max(a, b)))


def _fmt(c):
    if isinstance(c, (int, np.integer)):
        return str(int(c))
    return repr(float(c))


class AffineFunction:
    """Sum of LinearTerms plus a constant (src/functions.jl:218-221).  `linear` is an ordered list."""

    __array_priority__ = 1000

    def __init__(self, linear=(), constant=0):
        self.linear = [t if isinstance(t, LinearTerm) else LinearTerm(t[0], t[1] if isinstance(t[1], Variable) else Variable(t[1]))
                       for t in linear]
        self.constant = constant

    @staticmethod
    def zero():
        return AffineFunction([], 0)                                   # :243

    @staticmethod
    def of(x):
        """AffineFunction{T}(x) conversions (src/functions.jl:228-231)."""
        if isinstance(x, AffineFunction):
            return AffineFunction(list(x.linear), x.constant)
        if isinstance(x, LinearTerm):
            return AffineFunction([x], 0)
        if isinstance(x, Variable):
            return AffineFunction([LinearTerm(1, x)], 0)
        if _isnum(x):
            return AffineFunction([], x)
        raise TypeError(type(x))

    @staticmethod
    def from_arrays(terms, constant):
        """terms: numpy LT array fetched from the device."""
        return AffineFunction([LinearTerm(float(c), Variable(int(v))) for c, v in zip(terms["coeff"], terms["var"])], float(constant))

    def to_arrays(self):
        t = np.empty(len(self.linear), dtype=LT)
        for i, term in enumerate(self.linear):
            t[i] = (term.coeff, term.var.index)
        return t, float(self.constant)

    def copy(self):
        return AffineFunction(list(self.linear), self.constant)

    def __eq__(self, o):                                               # :239 (ordered)
        if isinstance(o, (Variable, LinearTerm)) or _isnum(o):
            o = AffineFunction.of(o)
        return isinstance(o, AffineFunction) and self.linear == o.linear and self.constant == o.constant

    __hash__ = None

    def __repr__(self):                                                # :252-257
        return "".join("%r + " % t for t in self.linear) + _fmt(self.constant)

    def __call__(self, vals):                                          # :259-267
        ret = self.constant
        for t in self.linear:
            ret = ret + t.coeff * vals[t.var]
        return ret

    # in-place builders (restated for constants; the per-solve versions are HIP kernels)
    def add(self, x):                                                  # :452-455
        if _isnum(x):
            self.constant = self.constant + x
        elif isinstance(x, Variable):
            self.linear.append(LinearTerm(1, x))
        elif isinstance(x, LinearTerm):
            self.linear.append(x)
        elif isinstance(x, AffineFunction):
            self.linear.extend(list(x.linear))
            self.constant = self.constant + x.constant
        else:
            raise TypeError(type(x))
        return self

    def subtract(self, x):                                             # :474-485
        if _isnum(x):
            self.constant = self.constant - x
        elif isinstance(x, Variable):
            self.linear.append(LinearTerm(-1, x))
        elif isinstance(x, LinearTerm):
            self.linear.append(-x)
        elif isinstance(x, AffineFunction):
            self.linear.extend([-t for t in x.linear])
            self.constant = self.constant - x.constant
        else:
            raise TypeError(type(x))
        return self

    def muladd(self, x, y):                                            # :515-524
        if _isnum(x) and isinstance(y, AffineFunction):
            x, y = y, x
        if not (isinstance(x, AffineFunction) and _isnum(y)):
            raise TypeError("muladd!(::AffineFunction, %s, %s)" % (type(x).__name__, type(y).__name__))
        self.linear.extend([t * y for t in x.linear])
        self.constant = self.constant + x.constant * y
        return self

    def canonicalize(self):                                            # :269-272 + util.jl:9-26
        out = self.copy()
        out.linear = _sort_and_combine(out.linear, key=lambda t: t.var.index,
                                       combine=lambda a, b: LinearTerm(a.coeff + b.coeff, a.var))
        return out

    def prune_zero(self, atol=0):                                      # :294-297
        return AffineFunction([t for t in self.linear if abs(t.coeff) > atol], self.constant)

    def __pos__(self):
        return self

    def __neg__(self):
        return AffineFunction.zero().subtract(self)

    def __add__(self, o):
        return _addsub(self, o, +1)

    def __radd__(self, o):
        return _addsub(o, self, +1)

    def __sub__(self, o):
        return _addsub(self, o, -1)

    def __rsub__(self, o):
        return _addsub(o, self, -1)

    def __mul__(self, o):
        if _isnum(o):
            return AffineFunction.zero().muladd(self, o)               # :623
        if isinstance(o, (Variable, LinearTerm, AffineFunction)):
            return QuadraticFunction.zero().muladd(self, o)            # :618-622
        return NotImplemented

    def __rmul__(self, o):
        if _isnum(o):
            return AffineFunction.zero().muladd(o, self)               # :624
        if isinstance(o, (Variable, LinearTerm)):
            return QuadraticFunction.zero().muladd(o, self)
        return NotImplemented

    def __pow__(self, p):
        return _power_by_squaring(self, p)                             # :630


class QuadraticFunction:
    """Sum of QuadraticTerms plus an AffineFunction (src/functions.jl:326-329)."""

    __array_priority__ = 1000

    def __init__(self, quadratic=(), affine=None):
        self.quadratic = [t if isinstance(t, QuadraticTerm) else
                          QuadraticTerm(t[0], t[1] if isinstance(t[1], Variable) else Variable(t[1]),
                                        t[2] if isinstance(t[2], Variable) else Variable(t[2])) for t in quadratic]
        self.affine = affine if affine is not None else AffineFunction.zero()

    @staticmethod
    def zero():
        return QuadraticFunction([], AffineFunction.zero())              # :354

    @staticmethod
    def of(x):                                                           # :334-346
        if isinstance(x, QuadraticFunction):
            return QuadraticFunction(list(x.quadratic), x.affine.copy())
        if isinstance(x, QuadraticTerm):
            return QuadraticFunction([x], AffineFunction.zero())
        return QuadraticFunction([], AffineFunction.of(x))

    @staticmethod
    def from_arrays(quad, lin, constant):
        q = [QuadraticTerm(float(c), Variable(int(r)), Variable(int(cl))) for c, r, cl in zip(quad["coeff"], quad["row"], quad["col"])]
        return QuadraticFunction(q, AffineFunction.from_arrays(lin, constant))

    def to_arrays(self):
        q = np.empty(len(self.quadratic), dtype=QT)
        for i, t in enumerate(self.quadratic):
            q[i] = (t.coeff, t.rowvar.index, t.colvar.index)
        lin, const = self.affine.to_arrays()
        return q, lin, const

    def copy(self):
        return QuadraticFunction.of(self)

    def __eq__(self, o):                                                 # :350
        if isinstance(o, (Variable, LinearTerm, QuadraticTerm, AffineFunction)) or _isnum(o):
            o = QuadraticFunction.of(o)
        return isinstance(o, QuadraticFunction) and self.quadratic == o.quadratic and self.affine == o.affine

    __hash__ = None

    def __repr__(self):                                                  # :366-371
        return "".join("%r + " % t for t in self.quadratic) + repr(self.affine)

    def __call__(self, vals):                                            # :373-379
        ret = self.affine(vals)
        for t in self.quadratic:
            ret = ret + t.coeff * vals[t.rowvar] * vals[t.colvar]
        return ret

    def add(self, x):                                                    # :457-459
        if isinstance(x, QuadraticTerm):
            self.quadratic.append(x)
        elif isinstance(x, QuadraticFunction):
            self.quadratic.extend(list(x.quadratic))
            self.affine.add(x.affine)
        else:
            self.affine.add(x)
        return self

    def subtract(self, x):                                               # :487-500
        if isinstance(x, QuadraticTerm):
            self.quadratic.append(-x)
        elif isinstance(x, QuadraticFunction):
            self.quadratic.extend([-t for t in x.quadratic])
            self.affine.subtract(x.affine)
        else:
            self.affine.subtract(x)
        return self

    def muladd(self, x, y):                                              # :526-576
        if _isnum(x) and isinstance(y, QuadraticFunction):
            x, y = y, x
        if isinstance(x, QuadraticFunction) and _isnum(y):               # :526-534
            self.quadratic.extend([t * y for t in x.quadratic])
            self.affine.muladd(x.affine, y)
            return self
        if isinstance(x, (Variable, LinearTerm)) and isinstance(y, AffineFunction):
            x, y = y, x                                                  # :546
        if isinstance(x, AffineFunction) and isinstance(y, (Variable, LinearTerm)):   # :537-545
            self.quadratic.extend([t * y for t in x.linear])
            self.affine.add(x.constant * y)
            return self
        if isinstance(x, AffineFunction) and isinstance(y, AffineFunction):            # :548-576
            for tx in x.linear:
                for ty in y.linear:
                    self.quadratic.append(tx * ty)
            self.affine.linear.extend([t * y.constant for t in x.linear])
            self.affine.linear.extend([t * x.constant for t in y.linear])
            self.affine.constant = self.affine.constant + x.constant * y.constant
            return self
        raise TypeError("muladd!(::QuadraticFunction, %s, %s)" % (type(x).__name__, type(y).__name__))

    def canonicalize(self):                                              # :381-386
        out = self.copy()
        out.affine = out.affine.canonicalize()

        def key(t):
            c = t.canonicalize()
            return (c.rowvar.index, c.colvar.index)

        def combine(a, b):                                               # :186-191
            a = a.canonicalize()
            return QuadraticTerm(a.coeff + b.coeff, a.rowvar, a.colvar)

        out.quadratic = _sort_and_combine(out.quadratic, key=key, combine=combine)
        return out

    def prune_zero(self, atol=0):                                        # :409-413 (affine part pruned with the default atol)
        return QuadraticFunction([t for t in self.quadratic if abs(t.coeff) > atol], self.affine.prune_zero())

    def __pos__(self):
        return self

    def __add__(self, o):
        return _addsub(self, o, +1)

    def __radd__(self, o):
        return _addsub(o, self, +1)

    def __sub__(self, o):
        return _addsub(self, o, -1)

    def __rsub__(self, o):
        return _addsub(o, self, -1)

    def __mul__(self, o):
        if _isnum(o):
            return QuadraticFunction.zero().muladd(self, o)              # :625
        return NotImplemented

    __rmul__ = __mul__


def _sort_and_combine(v, key, combine):
    """sort_and_combine! (src/util.jl:9-26).  Python's sort is stable whereas Base.Sort.QuickSort is not, so the
    order in which duplicates are summed may differ from Julia's in the last bits (tolerance-checked)."""
    if not v:
        return v
    v = sorted(v, key=key)
    out = [v[0]]
    for x in v[1:]:
        if key(out[-1]) < key(x):
            out.append(x)
        else:
            out[-1] = combine(out[-1], x)
    return out


_RANK = {Variable: 1, LinearTerm: 1, AffineFunction: 2, QuadraticTerm: 3, QuadraticFunction: 4}


def _addsub(x, y, sign):
    """x + y / x - y following the method table at src/functions.jl:581-616: the result type is the smallest
    function type holding both; the left operand is copied and the right one add!-ed / subtract!-ed."""
    def rank(v):
        if _isnum(v):
            return 0
        r = _RANK.get(type(v))
        if r is None:
            raise TypeError(type(v))
        return r
    try:
        rx, ry = rank(x), rank(y)
    except TypeError:
        return NotImplemented
    if rx == 0 and ry == 0:
        return x + y if sign > 0 else x - y
    top = max(rx, ry)
    dest = AffineFunction.of(x) if top <= 2 else QuadraticFunction.of(x)
    return dest.add(y) if sign > 0 else dest.subtract(y)


def _power_by_squaring(x, p):
    if not isinstance(p, (int, np.integer)) or p < 0:
        raise ArgumentError("only non-negative integer powers are supported")
    if p == 0:
        return 1
    if p == 1:
        return x
    if p == 2:
        return x * x
    raise ArgumentError("powers above 2 leave the quadratic function types")


def _relation(op):
    def method(self, other):
        from .lazyexpression import Relation
        return Relation(self, op, other)
    return method


for _cls in (Variable, LinearTerm, QuadraticTerm, AffineFunction, QuadraticFunction):
    # `x + 2y >= 4` inside constraint(model, ...) (src/model.jl:224-249); `==` keeps Julia's structural meaning,
    # use constraint(model, lhs, "==", rhs) for equality constraints between constants.
    _cls.__le__ = _relation("<=")
    _cls.__ge__ = _relation(">=")


def canonicalize(f):
    return f.canonicalize()


def prune_zero(f, atol=0):
    return f.prune_zero(atol)


def dot_scalar(x, y):
    """LinearAlgebra.dot on two scalar ParametronFunctions = x * y (src/functions.jl:636-638)."""
    return x * y


__all__ = ["Variable", "LinearTerm", "QuadraticTerm", "AffineFunction", "QuadraticFunction", "canonicalize", "prune_zero",
           "DimensionMismatch", "ArgumentError"]
