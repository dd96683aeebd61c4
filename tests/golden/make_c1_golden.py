"""Generates tests/golden/c1_readme_example1.npz: BASELINE config 1 (README Example 1 of the reference: n = 8 variables, r = 8 residual
rows, m = 2 equality constraints), computed by a THIRD, independent implementation — plain Python floats and loops written from the
output specification in SURVEY.md Appendix A (A.1-A.6), not from the C oracle and not from the device code.  The oracle
(tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_golden.py) must both reproduce these arrays: indices exactly,
literal-mode coefficients bit for bit (one rounding per coefficient, Appendix A.2), canonical-mode coefficients within 1e-12.

Inputs are the counter-based stream of SURVEY.md §8(d) (make_fill_golden.fill): A seed 1, b seed 2, C seed 3, d seed 4 (x2), all
column-major; the bilinear matrix Q uses seed 5.  varmap is a fixed permutation of 1..n plus an offset."""
import os

import numpy as np

from make_fill_golden import fill

N, R, M_ROWS = 8, 8, 2
VARMAP = [13, 11, 18, 12, 15, 17, 14, 16]          # model_var_to_optimizer: Variable k -> optimizer index VARMAP[k-1]


def main():
    n, r, m = N, R, M_ROWS
    A = [float(v) for v in fill(r * n, 1)]          # A[i][j] at j*r + i (column-major)
    b = [float(v) for v in fill(r, 2)]
    Cm = [float(v) for v in fill(m * n, 3)]
    d = [float(v) for v in fill(m, 4, 2.0)]
    Q = [float(v) for v in fill(n * n, 5)]
    a = lambda i, j: A[j * r + i]
    c = lambda i, j: Cm[j * m + i]
    nb = [0.0 - b[i] for i in range(r)]
    vm = VARMAP
    # A.1 residual (native Vector{AffineFunction})
    res_coeff = [[a(i, j) for j in range(n)] for i in range(r)]
    res_var = [[j + 1 for j in range(n)] for i in range(r)]
    # A.2 literal objective -> MOI.ScalarQuadraticFunction
    quad = []
    for i in range(r):
        for j in range(n):
            for k in range(n):
                coeff = a(i, j) * a(i, k)
                if j == k:
                    coeff = 2 * coeff
                quad.append((coeff, vm[j], vm[k]))
    aff = []
    for i in range(r):
        for j in range(n):
            aff.append((nb[i] * a(i, j), vm[j]))
        for k in range(n):
            aff.append((nb[i] * a(i, k), vm[k]))
    const = 0.0
    for i in range(r):
        const = const + nb[i] * nb[i]
    # A.3 canonical objective (reference summation order is unspecified: plain left-to-right here, compared with tolerance)
    cquad = []
    for j in range(n):
        for k in range(j, n):
            s = 0.0
            for i in range(r):
                s = s + a(i, j) * a(i, k)
            cquad.append((2 * s, vm[j], vm[k]))
    caff = []
    for j in range(n):
        s = 0.0
        for i in range(r):
            s = s + nb[i] * a(i, j)
        caff.append((2 * s, vm[j]))
    # A.4 constraint C*x == d -> MOI.VectorAffineFunction
    vat = [(row + 1, c(row, col), vm[col]) for row in range(m) for col in range(n)]
    vconst = [0.0 - d[row] for row in range(m)]
    # A.5 bounds x - l with l = first n values of b's stream negated
    lows = [0.0 - b[i] for i in range(n)]
    bvat = [(i + 1, 1.0, vm[i]) for i in range(n)]
    bconst = [0.0 - lows[i] for i in range(n)]
    # A.6 bilinear x'Qx -> quadratic[k] = (Q[k] linear column-major, x[row], x[col]), row = k div n, col = k mod n; MOI doubling on the diagonal
    bil = []
    for k in range(n * n):
        row, col = k // n, k % n
        coeff = Q[k]
        if row == col:
            coeff = 2 * coeff
        bil.append((coeff, vm[row], vm[col]))
    QT = np.dtype([("coeff", "<f8"), ("row", "<i8"), ("col", "<i8")])
    LT = np.dtype([("coeff", "<f8"), ("var", "<i8")])
    VAT = np.dtype([("out", "<i8"), ("coeff", "<f8"), ("var", "<i8")])
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_readme_example1.npz"),
             n=n, r=r, m=m, varmap=np.array(vm, dtype=np.int64), A=np.array(A), b=np.array(b), C=np.array(Cm), d=np.array(d), Q=np.array(Q),
             residual_coeff=np.array(res_coeff), residual_var=np.array(res_var, dtype=np.int64), residual_const=np.array(nb),
             literal_quad=np.array(quad, dtype=QT), literal_aff=np.array(aff, dtype=LT), const=np.array([const]),
             canonical_quad=np.array(cquad, dtype=QT), canonical_aff=np.array(caff, dtype=LT),
             constraint_terms=np.array(vat, dtype=VAT), constraint_consts=np.array(vconst),
             lows=np.array(lows), bounds_terms=np.array(bvat, dtype=VAT), bounds_consts=np.array(bconst),
             bilinear_quad=np.array(bil, dtype=QT))


if __name__ == "__main__":
    main()
