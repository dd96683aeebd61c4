"""Test infrastructure: a tiny dense QP 'optimizer' implementing the slice of MathOptInterface that Parametron drives
(parametron_jl_amd.AbstractOptimizer).  It stands in for OSQP/GLPK, which the reference's own tests use and which are
third-party solvers outside the scope of the hot path.  Equality-constrained problems are solved exactly through the KKT
system; inequalities go through scipy's SLSQP from the KKT point."""
import numpy as np
from scipy import optimize

import parametron_jl_amd as P
from parametron_jl_amd import moi
from moi_dense import dense_quadratic, dense_scalar_affine, dense_vector_affine


class DenseQPOptimizer(P.AbstractOptimizer):
    def __init__(self, variable_offset=0, permute_seed=None):
        self.offset = variable_offset
        self.permute_seed = permute_seed
        self.x = None
        self.objval = None
        self.status = "OPTIMIZE_NOT_CALLED"
        self.set_calls = 0

    def copy_to(self, backend):
        self.n = backend.nvars
        self.sense = backend.sense
        idx = np.arange(1, self.n + 1, dtype=np.int64)
        if self.permute_seed is not None:
            idx = np.random.default_rng(self.permute_seed).permutation(self.n).astype(np.int64) + 1
        self.varidx = idx + self.offset                     # optimizer index of Variable k
        # MOI.copy_to translates the indices of the functions it copies (src/model.jl:118); functions set later through
        # set_*_function already carry optimizer indices (update! applies model_var_to_optimizer)
        self.objective = self._translated(backend.objective.f)
        self.cons, self.sets = {}, {}
        cmap = {}
        for i, c in enumerate(backend.constraints):
            cmap[c] = i
            self.cons[i], self.sets[i] = self._translated(c.f), c.set
        return {"variables": self.varidx, "constraints": cmap}

    def _translated(self, f):
        import copy
        g = copy.copy(f)
        for attr, fields in (("terms", ["var"]), ("affine_terms", ["var"]), ("quadratic_terms", ["row", "col"])):
            arr = getattr(f, attr, None)
            if arr is not None and len(arr):
                arr = arr.copy()
                for fld in fields:
                    arr[fld] = self.varidx[arr[fld] - 1]
                setattr(g, attr, arr)
        return g

    def set_objective_function(self, f):
        self.objective = f
        self.set_calls += 1

    def set_constraint_function(self, index, f):
        self.cons[index] = f
        self.set_calls += 1

    def _shift(self, arr, fields):
        a = arr.copy()
        for f in fields:
            a[f] = a[f] - self.offset
        return a

    def _dense_objective(self):
        f = self.objective
        n = self.n
        if isinstance(f, moi.ScalarQuadraticFunction):
            return dense_quadratic(self._shift(f.affine_terms, ["var"]), self._shift(f.quadratic_terms, ["row", "col"]), f.constant, n)
        a, c = dense_scalar_affine(self._shift(f.terms, ["var"]), f.constant, n)
        return np.zeros((n, n)), a, c

    def _rows(self):
        """all constraints as rows g(x) = M x + k with sense in {'==', '>=', '<='} (0 on the right-hand side)"""
        Ms, ks, senses = [], [], []
        for i, f in self.cons.items():
            s = self.sets[i]
            if isinstance(f, moi.VectorAffineFunction):
                M, k = dense_vector_affine(self._shift(f.terms, ["var"]), f.constants, self.n)
                sense = {moi.Zeros: "==", moi.Nonnegatives: ">=", moi.Nonpositives: "<="}[type(s)]
            elif isinstance(f, moi.ScalarAffineFunction):
                a, c = dense_scalar_affine(self._shift(f.terms, ["var"]), f.constant, self.n)
                M, k = a[None, :], np.array([c - (s.value or 0.0)])
                sense = {moi.EqualTo: "==", moi.GreaterThan: ">=", moi.LessThan: "<="}[type(s)]
            else:
                raise NotImplementedError(type(f))
            Ms.append(M); ks.append(k); senses += [sense] * len(k)
        if not Ms:
            return np.zeros((0, self.n)), np.zeros(0), []
        return np.vstack(Ms), np.concatenate(ks), senses

    def optimize(self):
        Q, a, c = self._dense_objective()
        sgn = 1.0 if self.sense == P.Minimize else -1.0
        Q, a = sgn * Q, sgn * a
        M, k, senses = self._rows()
        eq = np.array([s == "==" for s in senses], dtype=bool)
        n = self.n
        Aeq, beq = M[eq], -k[eq]
        K = np.block([[Q, Aeq.T], [Aeq, np.zeros((Aeq.shape[0],) * 2)]])
        x = np.linalg.lstsq(K, np.concatenate([-a, beq]), rcond=None)[0][:n]
        if not eq.all():
            cons = []
            if eq.any():
                cons.append({"type": "eq", "fun": lambda v: Aeq @ v - beq, "jac": lambda v: Aeq})
            for sense, sg in ((">=", 1.0), ("<=", -1.0)):
                sel = np.array([s == sense for s in senses], dtype=bool)
                if sel.any():
                    Mi, ki = sg * M[sel], sg * k[sel]
                    cons.append({"type": "ineq", "fun": lambda v, Mi=Mi, ki=ki: Mi @ v + ki, "jac": lambda v, Mi=Mi: Mi})
            res = optimize.minimize(lambda v: 0.5 * v @ Q @ v + a @ v, x, jac=lambda v: Q @ v + a, constraints=cons, method="SLSQP",
                                    options={"ftol": 1e-14, "maxiter": 500})
            x = res.x
        self.x = x
        self.objval = sgn * (0.5 * x @ Q @ x + a @ x) + c
        self.status = "OPTIMAL"

    def variable_primal(self, index):
        return float(self.x[index - self.offset - 1])          # dense column of optimizer index `index`

    def objective_value(self): return float(self.objval)
    def termination_status(self): return self.status
    def primal_status(self): return "FEASIBLE_POINT" if self.x is not None else "NO_SOLUTION"
    def dual_status(self): return "FEASIBLE_POINT" if self.x is not None else "NO_SOLUTION"


class TinyMIPOptimizer(DenseQPOptimizer):
    """Stands in for GLPK in the reference's boolean / integer tests (test/model.jl:222-267): variables with a ZeroOne or Integer
    SingleVariable constraint are ENUMERATED (binary: {0, 1}; integer: -8..8), the others must not exist — a linear objective over a
    handful of integer points, exactly what those tests need.  Third-party solver stand-in, test infrastructure only."""

    def copy_to(self, backend):
        self.domains = {}
        kept = []
        for c in backend.constraints:
            if isinstance(c.f, moi.SingleVariable):
                self.domains[c.f.variable.index] = (0, 1) if isinstance(c.set, moi.ZeroOne) else tuple(range(-8, 9))
            else:
                kept.append(c)
        import types
        out = super().copy_to(types.SimpleNamespace(nvars=backend.nvars, sense=backend.sense, objective=backend.objective, constraints=kept))
        for c in backend.constraints:
            if isinstance(c.f, moi.SingleVariable):
                out["constraints"][c] = -1
        return out

    def optimize(self):
        import itertools
        Q, a, c = self._dense_objective()
        assert not Q.any() and len(self.domains) == self.n, "TinyMIPOptimizer: linear objective, every variable integer or binary"
        M, k, senses = self._rows()
        # model Variable j (1-based) sits in dense column (varidx[j - 1] - offset - 1)
        cols = [int(self.varidx[j - 1] - self.offset - 1) for j in sorted(self.domains)]
        best, bestx = None, None
        for point in itertools.product(*[self.domains[j] for j in sorted(self.domains)]):
            x = np.zeros(self.n)
            x[cols] = point
            g = M @ x + k
            ok = all((s == "==" and abs(v) < 1e-9) or (s == ">=" and v >= -1e-9) or (s == "<=" and v <= 1e-9) for s, v in zip(senses, g))
            if not ok:
                continue
            val = a @ x + c
            if best is None or (val < best if self.sense == P.Minimize else val > best):
                best, bestx = val, x
        self.x, self.objval = bestx, best
        self.status = "OPTIMAL" if bestx is not None else "INFEASIBLE"
