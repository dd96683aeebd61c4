"""Test helper: interpret MOI function buffers the way MathOptInterface 0.8 does.

ScalarQuadraticFunction = 1/2 x'Qx + a'x + c with Q symmetric; duplicate entries are summed
and mirrored (i,j)/(j,i) entries are duplicates of each other, so an off-diagonal term with
coefficient c sets Q[i,j] = Q[j,i] += c and a diagonal term sets Q[i,i] += c (hence the
reference's diagonal doubling, src/moi_interop.jl:58).
"""
import numpy as np


def dense_quadratic(affine_terms, quadratic_terms, constant, nvars):
    Q = np.zeros((nvars, nvars))
    a = np.zeros(nvars)
    for c, i, j in zip(quadratic_terms["coeff"], quadratic_terms["row"], quadratic_terms["col"]):
        if i == j:
            Q[i - 1, i - 1] += c
        else:
            Q[i - 1, j - 1] += c
            Q[j - 1, i - 1] += c
    np.add.at(a, affine_terms["var"] - 1, affine_terms["coeff"])
    return Q, a, float(constant)


def dense_vector_affine(terms, constants, nvars):
    m = len(constants)
    M = np.zeros((m, nvars))
    np.add.at(M, (terms["out"] - 1, terms["var"] - 1), terms["coeff"])
    return M, np.asarray(constants, dtype=float)


def dense_scalar_affine(terms, constant, nvars):
    a = np.zeros(nvars)
    np.add.at(a, terms["var"] - 1, terms["coeff"])
    return a, float(constant)


def solve_eq_qp(Q, a, Aeq, beq):
    """min 1/2 x'Qx + a'x  s.t. Aeq x = beq  (KKT solve)."""
    n, m = Q.shape[0], Aeq.shape[0]
    K = np.block([[Q, Aeq.T], [Aeq, np.zeros((m, m))]])
    rhs = np.concatenate([-a, beq])
    sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
    return sol[:n]
