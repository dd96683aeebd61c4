/*
 * parametron_oracle.c — CPU restatement of Parametron.jl's parameter-update hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (parametron.jl_amd + libparametron_hip.so) never links, loads or calls it.
 *
 * The reference (tkoolen/Parametron.jl v0.9.1) is pure Julia and Julia is not installed in
 * this image, so the reference itself can be neither compiled nor imported.  This file
 * restates, loop for loop, the reference functions on the update!() path.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Parity pin: the restatement is checked in tests/test_oracle_golden.py against every
 * known-answer test the reference holds for this path (test/functions.jl, test/util.jl,
 * closed forms of test/model.jl) — see SURVEY.md §8(c).
 *
 * Third-party algorithm restated: canonicalize! sorts with Base.Sort.QuickSort
 * (src/functions.jl:270,384).  Julia Base is not part of /root/reference; the algorithm
 * below (median-of-three pivot, Hoare partition, insertion sort for spans <= 20, recurse
 * on the smaller half) is Julia 1.0's base/sort.jl `sort!(v, lo, hi, ::QuickSortAlg, o)`,
 * restated from its published source.  It only determines the ORDER in which duplicate
 * terms are summed, which the parity tests treat with a 1e-12 relative tolerance.
 *
 * Arithmetic: fp64, no FMA contraction (build with -ffp-contract=off), one rounding per
 * product exactly as Julia (which never contracts a*b+c).
 *
 * Layouts (Julia isbits structs):
 *   LinearTerm{Float64}      = { double coeff; int64 var; }                    16 B
 *   QuadraticTerm{Float64}   = { double coeff; int64 rowvar; int64 colvar; }   24 B
 *   MOI.ScalarAffineTerm     = { double coefficient; int64 variable_index; }   16 B
 *   MOI.ScalarQuadraticTerm  = { double; int64; int64; }                       24 B
 *   MOI.VectorAffineTerm     = { int64 output_index; double; int64; }          24 B
 * Variable indices are 1-based (Julia Variable.index).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { double coeff; int64_t var; } pmo_lt;              /* functions.jl:110-113 */
typedef struct { double coeff; int64_t row; int64_t col; } pmo_qt; /* functions.jl:136-140 */
typedef struct { int64_t out; double coeff; int64_t var; } pmo_vat;/* moi_interop.jl:75     */

/* AffineFunction{Float64}: Vector{LinearTerm} + RefValue constant — functions.jl:218-221 */
typedef struct { pmo_lt *linear; int64_t n; int64_t cap; double constant; } pmo_aff;
/* QuadraticFunction{Float64}: Vector{QuadraticTerm} + AffineFunction — functions.jl:326-329 */
typedef struct { pmo_qt *quad; int64_t n; int64_t cap; pmo_aff affine; } pmo_quad;

#define PMO_OK 0
#define PMO_DIMENSION_MISMATCH 1   /* Julia DimensionMismatch */
#define PMO_ARGUMENT_ERROR 2       /* Julia ArgumentError     */

/* ------------------------------------------------------------------------------------ */
/* storage helpers: Julia resize!/empty!/push! keep capacity => zero steady-state allocs  */

static void aff_reserve(pmo_aff *f, int64_t n) {
    if (n > f->cap) {
        int64_t cap = f->cap ? f->cap : 4;
        while (cap < n) cap *= 2;
        f->linear = (pmo_lt *)realloc(f->linear, (size_t)cap * sizeof(pmo_lt));
        f->cap = cap;
    }
}
static void aff_resize(pmo_aff *f, int64_t n) { aff_reserve(f, n); f->n = n; }
static void quad_resize(pmo_quad *f, int64_t n) {
    if (n > f->cap) {
        int64_t cap = f->cap ? f->cap : 4;
        while (cap < n) cap *= 2;
        f->quad = (pmo_qt *)realloc(f->quad, (size_t)cap * sizeof(pmo_qt));
        f->cap = cap;
    }
    f->n = n;
}

pmo_aff *pmo_aff_new(void) { return (pmo_aff *)calloc(1, sizeof(pmo_aff)); }
void pmo_aff_free(pmo_aff *f) { if (f) { free(f->linear); free(f); } }
pmo_quad *pmo_quad_new(void) { return (pmo_quad *)calloc(1, sizeof(pmo_quad)); }
void pmo_quad_free(pmo_quad *f) { if (f) { free(f->quad); free(f->affine.linear); free(f); } }
/* a Vector{AffineFunction} is an array of pmo_aff structs */
pmo_aff *pmo_affvec_new(int64_t n) { return (pmo_aff *)calloc((size_t)(n > 0 ? n : 1), sizeof(pmo_aff)); }
void pmo_affvec_free(pmo_aff *v, int64_t n) {
    if (!v) return;
    for (int64_t i = 0; i < n; i++) free(v[i].linear);
    free(v);
}
pmo_aff *pmo_affvec_at(pmo_aff *v, int64_t i) { return &v[i]; }

int64_t pmo_aff_nterms(const pmo_aff *f) { return f->n; }
double pmo_aff_constant(const pmo_aff *f) { return f->constant; }
const pmo_lt *pmo_aff_terms(const pmo_aff *f) { return f->linear; }
int64_t pmo_quad_nterms(const pmo_quad *f) { return f->n; }
const pmo_qt *pmo_quad_terms(const pmo_quad *f) { return f->quad; }
pmo_aff *pmo_quad_affine(pmo_quad *f) { return &f->affine; }

/* ------------------------------------------------------------------------------------ */
/* zero!  — functions.jl:244 (affine), :355 (quadratic)                                   */
void pmo_aff_zero(pmo_aff *f) { f->n = 0; f->constant = 0; }
void pmo_quad_zero(pmo_quad *f) { f->n = 0; pmo_aff_zero(&f->affine); }

/* term algebra — functions.jl:114 (LinearTerm{T}(var) = one(T)*var), :158-160 (-, c*term),
 * :146-149 (products giving QuadraticTerm) */
static inline pmo_lt lt_of_var(int64_t var) { pmo_lt t = {1.0, var}; return t; }
static inline pmo_lt lt_neg(pmo_lt t) { t.coeff = -t.coeff; return t; }
static inline pmo_lt lt_scale(double c, pmo_lt t) { t.coeff = c * t.coeff; return t; }   /* :159 */
static inline pmo_qt qt_neg(pmo_qt t) { t.coeff = -t.coeff; return t; }
static inline pmo_qt qt_scale(double c, pmo_qt t) { t.coeff = c * t.coeff; return t; }
static inline pmo_qt qt_lt_lt(pmo_lt x, pmo_lt y) { pmo_qt q = {x.coeff * y.coeff, x.var, y.var}; return q; } /* :149 */
static inline pmo_qt qt_lt_var(pmo_lt y, int64_t x) { pmo_qt q = {y.coeff, y.var, x}; return q; }            /* :147 */

/* push helpers used by tests to build inputs (push!(f.linear, term)) */
void pmo_aff_push(pmo_aff *f, double coeff, int64_t var) {
    aff_reserve(f, f->n + 1); f->linear[f->n].coeff = coeff; f->linear[f->n].var = var; f->n++;
}
void pmo_aff_set_constant(pmo_aff *f, double c) { f->constant = c; }
void pmo_quad_push(pmo_quad *f, double coeff, int64_t row, int64_t col) {
    quad_resize(f, f->n + 1); pmo_qt t = {coeff, row, col}; f->quad[f->n - 1] = t;
}

/* ------------------------------------------------------------------------------------ */
/* copyto!  — functions.jl:419-439                                                       */
void pmo_aff_copy_number(pmo_aff *f, double x) { pmo_aff_zero(f); f->constant = x; }            /* :419 */
void pmo_aff_copy_term(pmo_aff *f, double coeff, int64_t var) {                                  /* :420 */
    pmo_aff_zero(f); pmo_aff_push(f, coeff, var);
}
void pmo_aff_copy_var(pmo_aff *f, int64_t var) { pmo_aff_copy_term(f, 1.0, var); }              /* :421 */
void pmo_aff_copy(pmo_aff *f, const pmo_aff *x) {                                                /* :422-427 */
    aff_resize(f, x->n);
    if (x->n) memmove(f->linear, x->linear, (size_t)x->n * sizeof(pmo_lt));
    f->constant = x->constant;
}
void pmo_quad_copy_aff(pmo_quad *f, const pmo_aff *x) { f->n = 0; pmo_aff_copy(&f->affine, x); } /* :428-432 */
void pmo_quad_copy(pmo_quad *f, const pmo_quad *x) {                                             /* :434-439 */
    quad_resize(f, x->n);
    if (x->n) memmove(f->quad, x->quad, (size_t)x->n * sizeof(pmo_qt));
    pmo_aff_copy(&f->affine, &x->affine);
}

/* ------------------------------------------------------------------------------------ */
/* add!  — functions.jl:452-461                                                          */
void pmo_aff_add_number(pmo_aff *f, double x) { f->constant += x; }                             /* :452 */
void pmo_aff_add_term(pmo_aff *f, double coeff, int64_t var) { pmo_aff_push(f, coeff, var); }   /* :454 */
void pmo_aff_add_var(pmo_aff *f, int64_t var) { pmo_aff_push(f, 1.0, var); }                    /* :453 */
void pmo_aff_add_aff(pmo_aff *f, const pmo_aff *x) {                                             /* :455 */
    int64_t off = f->n, xn = x->n;
    aff_resize(f, off + xn);
    memmove(f->linear + off, x->linear, (size_t)xn * sizeof(pmo_lt)); /* append! (x may alias f) */
    f->constant += x->constant;
}
void pmo_quad_add_term(pmo_quad *f, double c, int64_t r, int64_t cl) { pmo_quad_push(f, c, r, cl); } /* :458 */
void pmo_quad_add_quad(pmo_quad *f, const pmo_quad *x) {                                         /* :459 */
    int64_t off = f->n, xn = x->n;
    quad_resize(f, off + xn);
    memmove(f->quad + off, x->quad, (size_t)xn * sizeof(pmo_qt));
    pmo_aff_add_aff(&f->affine, &x->affine);
}

/* subtract!  — functions.jl:474-500                                                     */
void pmo_aff_sub_number(pmo_aff *f, double x) { f->constant -= x; }                             /* :474 */
void pmo_aff_sub_term(pmo_aff *f, double coeff, int64_t var) { pmo_aff_push(f, -coeff, var); }  /* :476 */
void pmo_aff_sub_var(pmo_aff *f, int64_t var) { pmo_aff_push(f, -1.0, var); }                   /* :475 */
void pmo_aff_sub_aff(pmo_aff *f, const pmo_aff *x) {                                             /* :477-485 */
    int64_t off = f->n, xn = x->n;
    aff_resize(f, off + xn);
    const pmo_lt *xl = x->linear;
    for (int64_t i = 0; i < xn; i++) f->linear[off + i] = lt_neg(xl[i]);
    f->constant -= x->constant;
}
void pmo_quad_sub_quad(pmo_quad *f, const pmo_quad *x) {                                         /* :492-500 */
    int64_t off = f->n, xn = x->n;
    quad_resize(f, off + xn);
    for (int64_t i = 0; i < xn; i++) f->quad[off + i] = qt_neg(x->quad[i]);
    pmo_aff_sub_aff(&f->affine, &x->affine);
}

/* ------------------------------------------------------------------------------------ */
/* muladd!  — functions.jl:515-576                                                       */
void pmo_aff_muladd_aff_number(pmo_aff *dest, const pmo_aff *x, double y) {                     /* :515-523 */
    int64_t off = dest->n, xn = x->n;
    aff_resize(dest, off + xn);
    for (int64_t i = 0; i < xn; i++) dest->linear[off + i] = lt_scale(y, x->linear[i]);
    dest->constant += x->constant * y;
}
void pmo_quad_muladd_quad_number(pmo_quad *dest, const pmo_quad *x, double y) {                 /* :526-534 */
    int64_t off = dest->n, xn = x->n;
    quad_resize(dest, off + xn);
    for (int64_t i = 0; i < xn; i++) dest->quad[off + i] = qt_scale(y, x->quad[i]);
    pmo_aff_muladd_aff_number(&dest->affine, &x->affine, y);
}
/* affine x (Variable | LinearTerm): y given as (ycoeff, yvar); a bare Variable is (1.0, var)
 * but note :541 `x.linear[i] * y` with y::Variable keeps the term coefficient unchanged
 * (functions.jl:147) whereas y::LinearTerm multiplies coefficients (:149); `is_var` selects. */
void pmo_quad_muladd_aff_term(pmo_quad *dest, const pmo_aff *x, double ycoeff, int64_t yvar, int is_var) { /* :537-545 */
    int64_t off = dest->n, xn = x->n;
    quad_resize(dest, off + xn);
    pmo_lt y = {ycoeff, yvar};
    for (int64_t i = 0; i < xn; i++)
        dest->quad[off + i] = is_var ? qt_lt_var(x->linear[i], yvar) : qt_lt_lt(x->linear[i], y);
    /* add!(dest.affine, x.constant[] * y): Number*Variable = LinearTerm(c, var) (:120);
     * Number*LinearTerm = c * coeff (:159) */
    double c = is_var ? x->constant : x->constant * ycoeff;
    pmo_aff_push(&dest->affine, c, yvar);
}
void pmo_quad_muladd_aff_aff(pmo_quad *dest, const pmo_aff *x, const pmo_aff *y) {              /* :548-576 */
    const pmo_lt *xl = x->linear, *yl = y->linear;
    int64_t xn = x->n, yn = y->n;
    int64_t quadoffset = dest->n;
    quad_resize(dest, quadoffset + xn * yn);
    int64_t k = 0;
    for (int64_t i = 0; i < xn; i++)
        for (int64_t j = 0; j < yn; j++)
            dest->quad[quadoffset + k++] = qt_lt_lt(xl[i], yl[j]);
    pmo_aff *da = &dest->affine;
    double xconst = x->constant, yconst = y->constant;
    int64_t linoffset = da->n;
    aff_resize(da, linoffset + xn + yn);
    k = linoffset;
    for (int64_t i = 0; i < xn; i++) da->linear[k++] = lt_scale(yconst, xl[i]);   /* :567 term*c -> c*coeff */
    for (int64_t i = 0; i < yn; i++) da->linear[k++] = lt_scale(xconst, yl[i]);   /* :571 */
    da->constant += xconst * yconst;                                              /* :574 */
}
/* mul!(dest, x, y) = zero!(dest); muladd!(dest, x, y) — functions.jl:578 */
void pmo_aff_mul_aff_number(pmo_aff *d, const pmo_aff *x, double y) { pmo_aff_zero(d); pmo_aff_muladd_aff_number(d, x, y); }
void pmo_quad_mul_quad_number(pmo_quad *d, const pmo_quad *x, double y) { pmo_quad_zero(d); pmo_quad_muladd_quad_number(d, x, y); }
void pmo_quad_mul_aff_term(pmo_quad *d, const pmo_aff *x, double yc, int64_t yv, int is_var) { pmo_quad_zero(d); pmo_quad_muladd_aff_term(d, x, yc, yv, is_var); }
void pmo_quad_mul_aff_aff(pmo_quad *d, const pmo_aff *x, const pmo_aff *y) { pmo_quad_zero(d); pmo_quad_muladd_aff_aff(d, x, y); }

/* ------------------------------------------------------------------------------------ */
/* vecdot!  — functions.jl:665-731                                                       */
/* Number[] . AffineFunction[]  (:665-674 via muladd! :524 -> :515) */
int pmo_vecdot_aff_numbers_affs(pmo_aff *dest, const double *x, int64_t nx, const pmo_aff *y, int64_t ny) {
    pmo_aff_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) pmo_aff_muladd_aff_number(dest, &y[i], x[i]);
    return PMO_OK;
}
/* Number[] . Variable[]  (:676-687): linear[i] = x[i]*y[i] (order of the operands does not
 * matter for the result: Number*Variable == Variable*Number, :120-121) */
int pmo_vecdot_aff_numbers_vars(pmo_aff *dest, const double *x, int64_t nx, const int64_t *y, int64_t ny) {
    pmo_aff_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    aff_resize(dest, nx);
    for (int64_t i = 0; i < nx; i++) { dest->linear[i].coeff = x[i]; dest->linear[i].var = y[i]; }
    return PMO_OK;
}
/* Variable[] . Variable[]  (:689-700): quadratic[i] = x[i]*y[i] = QuadraticTerm(1, x, y) (:148) */
int pmo_vecdot_quad_vars_vars(pmo_quad *dest, const int64_t *x, int64_t nx, const int64_t *y, int64_t ny) {
    pmo_quad_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    quad_resize(dest, nx);
    for (int64_t i = 0; i < nx; i++) { pmo_qt t = {1.0, x[i], y[i]}; dest->quad[i] = t; }
    return PMO_OK;
}
/* (Variable | LinearTerm)[] . (Variable | LinearTerm)[]  (:689-700): quadratic[i] = x[i]*y[i].
 * A Variable is passed as LinearTerm(1.0, var): Variable*LinearTerm = QuadraticTerm(y.coeff, x, y.var)
 * (:146), LinearTerm*Variable = QuadraticTerm(x.coeff, x.var, y) (:147), LinearTerm*LinearTerm =
 * x.coeff*y.coeff (:149); multiplying by the exact 1.0 reproduces the first two bit for bit. */
int pmo_vecdot_quad_terms_terms(pmo_quad *dest, const pmo_lt *x, int64_t nx, const pmo_lt *y, int64_t ny) {
    pmo_quad_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    quad_resize(dest, nx);
    for (int64_t i = 0; i < nx; i++) dest->quad[i] = qt_lt_lt(x[i], y[i]);
    return PMO_OK;
}
/* AffineFunction[] . Variable[] (either order: :546 swaps)  (:702-709 -> :537-545) */
int pmo_vecdot_quad_affs_vars(pmo_quad *dest, const pmo_aff *x, int64_t nx, const int64_t *y, int64_t ny) {
    pmo_quad_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) pmo_quad_muladd_aff_term(dest, &x[i], 1.0, y[i], 1);
    return PMO_OK;
}
/* AffineFunction[] . AffineFunction[]  (:702-709 -> :548-576) — HOT LOOP 3 */
int pmo_vecdot_quad_affs_affs(pmo_quad *dest, const pmo_aff *x, int64_t nx, const pmo_aff *y, int64_t ny) {
    pmo_quad_zero(dest);
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) pmo_quad_muladd_aff_aff(dest, &x[i], &y[i]);
    return PMO_OK;
}

/* ------------------------------------------------------------------------------------ */
/* vecadd! / vecsubtract!  — functions.jl:751-764 ; scalar add!/subtract!(dest,x,y) :461,:502
 * dest[i] = zero!; copyto!(dest[i], x[i]); (add|subtract)!(dest[i], y[i])               */
int pmo_vecaddsub_affs_numbers(pmo_aff *dest, const pmo_aff *x, int64_t nx, const double *y, int64_t ny, int subtract) {
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) {
        pmo_aff_zero(&dest[i]);
        pmo_aff_copy(&dest[i], &x[i]);
        if (subtract) pmo_aff_sub_number(&dest[i], y[i]); else pmo_aff_add_number(&dest[i], y[i]);
    }
    return PMO_OK;
}
int pmo_vecaddsub_vars_numbers(pmo_aff *dest, const int64_t *x, int64_t nx, const double *y, int64_t ny, int subtract) {
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) {
        pmo_aff_zero(&dest[i]);
        pmo_aff_copy_var(&dest[i], x[i]);
        if (subtract) pmo_aff_sub_number(&dest[i], y[i]); else pmo_aff_add_number(&dest[i], y[i]);
    }
    return PMO_OK;
}
int pmo_vecaddsub_numbers_affs(pmo_aff *dest, const double *x, int64_t nx, const pmo_aff *y, int64_t ny, int subtract) {
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) {
        pmo_aff_zero(&dest[i]);
        pmo_aff_copy_number(&dest[i], x[i]);
        if (subtract) pmo_aff_sub_aff(&dest[i], &y[i]); else pmo_aff_add_aff(&dest[i], &y[i]);
    }
    return PMO_OK;
}
int pmo_vecaddsub_affs_affs(pmo_aff *dest, const pmo_aff *x, int64_t nx, const pmo_aff *y, int64_t ny, int subtract) {
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) {
        pmo_aff_zero(&dest[i]);
        pmo_aff_copy(&dest[i], &x[i]);
        if (subtract) pmo_aff_sub_aff(&dest[i], &y[i]); else pmo_aff_add_aff(&dest[i], &y[i]);
    }
    return PMO_OK;
}
int pmo_vecaddsub_affs_vars(pmo_aff *dest, const pmo_aff *x, int64_t nx, const int64_t *y, int64_t ny, int subtract) {
    if (nx != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nx; i++) {
        pmo_aff_zero(&dest[i]);
        pmo_aff_copy(&dest[i], &x[i]);
        if (subtract) pmo_aff_sub_var(&dest[i], y[i]); else pmo_aff_add_var(&dest[i], y[i]);
    }
    return PMO_OK;
}

/* ------------------------------------------------------------------------------------ */
/* matvecmul!(y, A, x::Vector{Variable})  — functions.jl:775-798 — HOT LOOP 1
 * A is rows x cols column-major; linear index i runs rows-fastest (:790-796).           */
int pmo_matvecmul_vars(pmo_aff *y, int64_t ny, const double *A, int64_t rows, int64_t cols,
                       const int64_t *x, int64_t nx) {
    if (ny != rows) return PMO_DIMENSION_MISMATCH;   /* :780 */
    if (nx != cols) return PMO_DIMENSION_MISMATCH;   /* :781 */
    for (int64_t row = 0; row < rows; row++) { pmo_aff_zero(&y[row]); aff_resize(&y[row], cols); }
    int64_t i = 0;
    for (int64_t col = 0; col < cols; col++)
        for (int64_t row = 0; row < rows; row++) {
            y[row].linear[col].coeff = A[i];      /* A[i] * x[col]  (:793, Number*Variable :120) */
            y[row].linear[col].var = x[col];
            i++;
        }
    return PMO_OK;
}
/* matvecmul!(y, A, x::Vector{AffineFunction})  — functions.jl:800-822 */
int pmo_matvecmul_affs(pmo_aff *y, int64_t ny, const double *A, int64_t rows, int64_t cols,
                       const pmo_aff *x, int64_t nx) {
    if (ny != rows) return PMO_DIMENSION_MISMATCH;
    if (nx != cols) return PMO_DIMENSION_MISMATCH;
    for (int64_t row = 0; row < rows; row++) pmo_aff_zero(&y[row]);
    int64_t i = 0;
    for (int64_t col = 0; col < cols; col++)
        for (int64_t row = 0; row < rows; row++) {
            pmo_aff_muladd_aff_number(&y[row], &x[col], A[i]);   /* :817 -> :524 -> :515 */
            i++;
        }
    return PMO_OK;
}

/* bilinearmul!(dest, Q, x', y)  — functions.jl:840-858.  NB: Q[k] is the column-major linear
 * index while (row, col) advance row-major: term k pairs Q[k] with
 * (x[k / ny], y[k % ny]). */
int pmo_bilinearmul(pmo_quad *dest, const double *Q, int64_t qrows, int64_t qcols,
                    const int64_t *x, int64_t nx, const int64_t *y, int64_t ny) {
    if (qrows != nx || qcols != ny) return PMO_DIMENSION_MISMATCH;   /* :845 */
    pmo_quad_zero(dest);
    quad_resize(dest, qrows * qcols);
    int64_t k = 0;
    for (int64_t row = 0; row < nx; row++)
        for (int64_t col = 0; col < ny; col++) {
            pmo_qt t = {Q[k], x[row], y[col]};
            dest->quad[k] = t;
            k++;
        }
    return PMO_OK;
}

/* scale!  — functions.jl:873-925 */
int pmo_scale_number_vars(pmo_lt *dest, int64_t nd, double x, const int64_t *y, int64_t ny) {    /* :873-893 */
    if (nd != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nd; i++) { dest[i].coeff = x; dest[i].var = y[i]; }
    return PMO_OK;
}
int pmo_scale_number_affs(pmo_aff *dest, int64_t nd, double x, const pmo_aff *y, int64_t ny) {   /* :895-915 */
    if (nd != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nd; i++) pmo_aff_mul_aff_number(&dest[i], &y[i], x);
    return PMO_OK;
}
int pmo_scale_number_numbers(double *dest, int64_t nd, double x, const double *y, int64_t ny) {  /* :917-925 */
    if (nd != ny) return PMO_DIMENSION_MISMATCH;
    for (int64_t i = 0; i < nd; i++) dest[i] = x * y[i];
    return PMO_OK;
}

/* vcat!  — functions.jl:969-994.  `srcs` = array of nsrc vectors, lens[k] their lengths. */
int pmo_vcat(pmo_aff *y, int64_t ny, const pmo_aff *const *srcs, const int64_t *lens, int64_t nsrc) {
    for (int64_t i = 0; i < ny; i++) pmo_aff_zero(&y[i]);                /* :989-991 */
    int64_t i = 0;                                                        /* 0-based cursor */
    for (int64_t k = 0; k < nsrc; k++) {
        if (!(i + lens[k] - 1 <= ny - 1)) return PMO_DIMENSION_MISMATCH;  /* :979 */
        for (int64_t s = 0; s < lens[k]; s++) { pmo_aff_copy(&y[i], &srcs[k][s]); i++; }
    }
    if (i != ny) return PMO_DIMENSION_MISMATCH;                           /* :970 */
    return PMO_OK;
}

/* ------------------------------------------------------------------------------------ */
/* canonical form — util.jl:9-26, functions.jl:182-191,269-272,381-386,294-297,409-413   */

#define SMALL_THRESHOLD 20   /* Julia base/sort.jl */

/* sort keys */
static inline int lt_key_less(const pmo_lt *a, const pmo_lt *b) { return a->var < b->var; }
static inline void qt_key(const pmo_qt *t, int64_t *r, int64_t *c) {   /* canonicalize(term) :182-184 */
    if (t->row <= t->col) { *r = t->row; *c = t->col; } else { *r = t->col; *c = t->row; }
}
static inline int qt_key_less(const pmo_qt *a, const pmo_qt *b) {
    int64_t ar, ac, br, bc; qt_key(a, &ar, &ac); qt_key(b, &br, &bc);
    return ar < br || (ar == br && ac < bc);   /* isless on tuples */
}

#define DEFINE_JULIA_QUICKSORT(NAME, T, LESS)                                            \
static void NAME##_insertion(T *v, int64_t lo, int64_t hi) {                             \
    for (int64_t i = lo + 1; i <= hi; i++) {                                             \
        int64_t j = i; T x = v[i];                                                       \
        while (j > lo) { if (LESS(&x, &v[j - 1])) { v[j] = v[j - 1]; j--; continue; } break; } \
        v[j] = x;                                                                        \
    }                                                                                    \
}                                                                                        \
static int64_t NAME##_partition(T *v, int64_t lo, int64_t hi) {                          \
    int64_t mi = (int64_t)(((uint64_t)(lo + hi)) >> 1);                                  \
    T t;                                                                                 \
    if (LESS(&v[lo], &v[mi])) { t = v[mi]; v[mi] = v[lo]; v[lo] = t; }                   \
    if (LESS(&v[hi], &v[lo])) {                                                          \
        if (LESS(&v[hi], &v[mi])) { T a = v[lo], b = v[mi], c = v[hi]; v[hi] = a; v[lo] = b; v[mi] = c; } \
        else { t = v[hi]; v[hi] = v[lo]; v[lo] = t; }                                    \
    }                                                                                    \
    T pivot = v[lo];                                                                     \
    int64_t i = lo, j = hi;                                                              \
    for (;;) {                                                                           \
        i++; j--;                                                                        \
        while (LESS(&v[i], &pivot)) i++;                                                 \
        while (LESS(&pivot, &v[j])) j--;                                                 \
        if (i >= j) break;                                                               \
        t = v[i]; v[i] = v[j]; v[j] = t;                                                 \
    }                                                                                    \
    v[lo] = v[j]; v[j] = pivot;                                                          \
    return j;                                                                            \
}                                                                                        \
static void NAME(T *v, int64_t lo, int64_t hi) {                                         \
    while (lo < hi) {                                                                    \
        if (hi - lo <= SMALL_THRESHOLD) { NAME##_insertion(v, lo, hi); return; }         \
        int64_t j = NAME##_partition(v, lo, hi);                                         \
        if (j - lo < hi - j) { if (lo < j - 1) NAME(v, lo, j - 1); lo = j + 1; }         \
        else { if (j + 1 < hi) NAME(v, j + 1, hi); hi = j - 1; }                         \
    }                                                                                    \
}
DEFINE_JULIA_QUICKSORT(lt_quicksort, pmo_lt, lt_key_less)
DEFINE_JULIA_QUICKSORT(qt_quicksort, pmo_qt, qt_key_less)

/* sort_and_combine! on linear terms — util.jl:9-26 with by = term.var.index,
 * combine = functions.jl:126-129 */
void pmo_aff_canonicalize(pmo_aff *f) {                                   /* functions.jl:269-272 */
    if (f->n == 0) return;                                                /* util.jl:12 */
    lt_quicksort(f->linear, 0, f->n - 1);
    int64_t j = 0;
    for (int64_t i = 1; i < f->n; i++) {
        pmo_lt x = f->linear[i];
        if (lt_key_less(&f->linear[j], &x)) { j++; f->linear[j] = x; }
        else { f->linear[j].coeff = f->linear[j].coeff + x.coeff; }       /* combine: t1.coeff + t2.coeff, var of t1 */
    }
    f->n = j + 1;
}
void pmo_quad_canonicalize(pmo_quad *f) {                                 /* functions.jl:381-386 */
    pmo_aff_canonicalize(&f->affine);
    if (f->n == 0) return;
    qt_quicksort(f->quad, 0, f->n - 1);
    int64_t j = 0;
    for (int64_t i = 1; i < f->n; i++) {
        pmo_qt x = f->quad[i];
        if (qt_key_less(&f->quad[j], &x)) { j++; f->quad[j] = x; }        /* NB: stored un-canonicalized (util.jl:18-19) */
        else {                                                            /* combine :186-191 canonicalizes both */
            int64_t r, c; qt_key(&f->quad[j], &r, &c);
            pmo_qt t = {f->quad[j].coeff + x.coeff, r, c};
            f->quad[j] = t;
        }
    }
    f->n = j + 1;
}
/* prune_zero!  — functions.jl:294-297 (affine), :409-413 (quadratic; NB the affine part is
 * pruned with the DEFAULT atol = 0, the kwarg is not forwarded, :410) */
void pmo_aff_prune_zero(pmo_aff *f, double atol) {
    int64_t j = 0;
    for (int64_t i = 0; i < f->n; i++) if (fabs(f->linear[i].coeff) > atol) f->linear[j++] = f->linear[i];
    f->n = j;
}
void pmo_quad_prune_zero(pmo_quad *f, double atol) {
    pmo_aff_prune_zero(&f->affine, 0.0);
    int64_t j = 0;
    for (int64_t i = 0; i < f->n; i++) if (fabs(f->quad[i].coeff) > atol) f->quad[j++] = f->quad[i];
    f->n = j;
}

/* evaluation at a point — functions.jl:259-267, :373-379 ; vals indexed by var (1-based) */
double pmo_aff_eval(const pmo_aff *f, const double *vals) {
    double ret = f->constant;
    for (int64_t i = 0; i < f->n; i++) ret += f->linear[i].coeff * vals[f->linear[i].var - 1];
    return ret;
}
double pmo_quad_eval(const pmo_quad *f, const double *vals) {
    double ret = pmo_aff_eval(&f->affine, vals);
    for (int64_t i = 0; i < f->n; i++) ret += f->quad[i].coeff * vals[f->quad[i].row - 1] * vals[f->quad[i].col - 1];
    return ret;
}

/* ------------------------------------------------------------------------------------ */
/* MOI copies — moi_interop.jl:35-81.  varmap[k-1] = model_var_to_optimizer[k].value
 * (model.jl:100-107); varmap == NULL is IdentityVarMap (moi_interop.jl:32-33).          */
static inline int64_t vm(const int64_t *varmap, int64_t var) { return varmap ? varmap[var - 1] : var; }

/* update!(::MOI.ScalarAffineFunction, ::AffineFunction, varmap) — :35-43 */
void pmo_moi_scalar_affine(const pmo_aff *f, const int64_t *varmap, pmo_lt *terms, double *constant) {
    *constant = f->constant;
    for (int64_t i = 0; i < f->n; i++) { terms[i].coeff = f->linear[i].coeff; terms[i].var = vm(varmap, f->linear[i].var); }
}
/* update!(::MOI.ScalarQuadraticFunction, ::QuadraticFunction, varmap) — :45-62 — HOT LOOP 4 */
void pmo_moi_scalar_quadratic(const pmo_quad *f, const int64_t *varmap,
                              pmo_lt *affine_terms, pmo_qt *quadratic_terms, double *constant) {
    const pmo_aff *a = &f->affine;
    *constant = a->constant;
    for (int64_t i = 0; i < a->n; i++) { affine_terms[i].coeff = a->linear[i].coeff; affine_terms[i].var = vm(varmap, a->linear[i].var); }
    for (int64_t i = 0; i < f->n; i++) {
        pmo_qt t = f->quad[i];
        quadratic_terms[i].row = vm(varmap, t.row);
        quadratic_terms[i].col = vm(varmap, t.col);
        quadratic_terms[i].coeff = (t.row == t.col) ? 2 * t.coeff : t.coeff;   /* :58 */
    }
}
/* update!(::MOI.VectorAffineFunction, ::Vector{AffineFunction}, varmap) — :64-81 — HOT LOOP 5 */
int64_t pmo_moi_vector_affine_nterms(const pmo_aff *fs, int64_t n) {
    int64_t s = 0; for (int64_t i = 0; i < n; i++) s += fs[i].n; return s;   /* :66-69 */
}
void pmo_moi_vector_affine(const pmo_aff *fs, int64_t n, const int64_t *varmap, pmo_vat *terms, double *constants) {
    int64_t i = 0;
    for (int64_t row = 0; row < n; row++) {
        const pmo_aff *f = &fs[row];
        for (int64_t k = 0; k < f->n; k++) {
            terms[i].out = row + 1;                                   /* Int64(row), 1-based */
            terms[i].coeff = f->linear[k].coeff;
            terms[i].var = vm(varmap, f->linear[k].var);
            i++;
        }
        constants[row] = f->constant;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Whole-path restatement for README Example 1 (README.md:23-57; call stack SURVEY §3.2):
 *     residual = A*x - b ; objective residual . residual ; constraint C*x - d in Zeros
 * One call = one update!(model) (model.jl:132-143) minus the third-party MOI.set calls.
 * As in the reference the residual expression is evaluated twice (no memoisation,
 * lazyexpression.jl:53-61) when `evaluate_residual_twice` != 0.
 * Scratch objects are owned by the caller-visible workspace so that steady-state calls do
 * not allocate (the reference's @allocated == 0 contract).                               */
typedef struct {
    int64_t n, r, m;
    pmo_aff *Ax, *residual;       /* r rows  (matvecmul! dest, vecsubtract! dest) */
    pmo_aff *Cx, *cres;           /* m rows */
    pmo_quad *objective;          /* vecdot! dest */
} pmo_lsq_workspace;

pmo_lsq_workspace *pmo_lsq_new(int64_t n, int64_t r, int64_t m) {
    pmo_lsq_workspace *w = (pmo_lsq_workspace *)calloc(1, sizeof(*w));
    w->n = n; w->r = r; w->m = m;
    w->Ax = pmo_affvec_new(r); w->residual = pmo_affvec_new(r);
    w->Cx = pmo_affvec_new(m); w->cres = pmo_affvec_new(m);
    w->objective = pmo_quad_new();
    return w;
}
void pmo_lsq_free(pmo_lsq_workspace *w) {
    if (!w) return;
    pmo_affvec_free(w->Ax, w->r); pmo_affvec_free(w->residual, w->r);
    pmo_affvec_free(w->Cx, w->m); pmo_affvec_free(w->cres, w->m);
    pmo_quad_free(w->objective); free(w);
}
pmo_aff *pmo_lsq_residual(pmo_lsq_workspace *w) { return w->residual; }
pmo_aff *pmo_lsq_constraint(pmo_lsq_workspace *w) { return w->cres; }
pmo_quad *pmo_lsq_objective(pmo_lsq_workspace *w) { return w->objective; }

/* the affine sub-DAG  (A*x - b): matvecmul! then vecsubtract! */
int pmo_lsq_eval_residual(pmo_lsq_workspace *w, const double *A, const double *b, const int64_t *xvar) {
    int rc = pmo_matvecmul_vars(w->Ax, w->r, A, w->r, w->n, xvar, w->n);
    if (rc) return rc;
    return pmo_vecaddsub_affs_numbers(w->residual, w->Ax, w->r, b, w->r, 1);
}
int pmo_lsq_eval_constraint(pmo_lsq_workspace *w, const double *C, const double *d, const int64_t *xvar) {
    int rc = pmo_matvecmul_vars(w->Cx, w->m, C, w->m, w->n, xvar, w->n);
    if (rc) return rc;
    return pmo_vecaddsub_affs_numbers(w->cres, w->Cx, w->m, d, w->m, 1);
}
/* rows_limit < 0: all rows.  rows_limit >= 0 restricts the objective expansion (HOT LOOPS
 * 3+4) to the first rows_limit rows of the residual — used ONLY by bench.py's bounded CPU
 * baseline sample (the full n = 4096 expansion is 1.65 TB). */
int pmo_lsq_eval_objective(pmo_lsq_workspace *w, const double *A, const double *b, const int64_t *xvar,
                           int evaluate_residual_twice, int64_t rows_limit) {
    int rc = pmo_lsq_eval_residual(w, A, b, xvar);
    if (rc) return rc;
    if (evaluate_residual_twice) { rc = pmo_lsq_eval_residual(w, A, b, xvar); if (rc) return rc; }
    int64_t rows = (rows_limit < 0 || rows_limit > w->r) ? w->r : rows_limit;
    return pmo_vecdot_quad_affs_affs(w->objective, w->residual, rows, w->residual, rows);
}

/* only the vecdot! node (HOT LOOP 3) on the already evaluated residual — lets bench.py time the nodes separately */
int pmo_lsq_eval_vecdot(pmo_lsq_workspace *w, int64_t rows_limit) {
    int64_t rows = (rows_limit < 0 || rows_limit > w->r) ? w->r : rows_limit;
    return pmo_vecdot_quad_affs_affs(w->objective, w->residual, rows, w->residual, rows);
}

/* ------------------------------------------------------------------------------------ */
/* synthetic inputs: counter-based RNG shared (bit-for-bit) with the device fill kernel
 * (SURVEY §8d): u = splitmix64(seed * 0x9E3779B97F4A7C15 + index) >> 11 * 2^-53 in [0,1) */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void pmo_fill_uniform(double *dst, int64_t n, uint64_t seed, double scale) {
    uint64_t base = seed * 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < n; i++)
        dst[i] = scale * ((double)(splitmix64(base + (uint64_t)i) >> 11) * 0x1.0p-53);
}

/* ------------------------------------------------------------------------------------ */
/* Canonical MOI coefficients of SAMPLED terms of residual . residual at sizes where the literal objective cannot be
 * materialised (n = r = 4096: 1.65 TB).  For residual = A*x (+|-) b the literal function (functions.jl:702-709 over
 * :548-576) holds, for every row i, the quadratic terms coeff = A[i,j]*A[i,k] at (j,k) for ALL ordered pairs and the
 * affine terms (c_i*A[i,j], x_j) twice (:566-573), c_i = 0.0 (+|-) b[i].  canonicalize! (:381-386, util.jl:9-26) adds up
 * the terms that share an unordered pair {j,k} (resp. a variable) in QuickSort order — an order the reference does not
 * specify — and the MOI copy doubles the diagonal (moi_interop.jl:58).  Restated per sampled pair, one product per
 * literal term, summed in row order, the (j,k) term of a row before its (k,j) term; no FMA (-ffp-contract=off):
 *     off-diagonal (j < k):  sum_i [ A[i,j]*A[i,k] ] + [ A[i,k]*A[i,j] ]  (2r terms)
 *     diagonal     (j = k):  2 * sum_i A[i,j]*A[i,j]                      (r terms, then the MOI doubling)
 *     affine       j:        sum_i [ c_i*A[i,j] ] + [ c_i*A[i,j] ]        (2r terms)
 * `out` gets that double-precision sum; `out_ld` (optional) the same sum of the SAME rounded products accumulated in
 * long double — the reference point for "how far is any summation order from the exact sum of the literal terms".
 * A is column-major with leading dimension lda; pairs are 1-based positions (j <= k). */
int pmo_canonical_quad_samples(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *pj, const int64_t *pk,
                               int64_t npairs, double *out, double *out_ld) {
    if (rows < 0 || cols < 0 || lda < rows) return PMO_DIMENSION_MISMATCH;
    for (int64_t p = 0; p < npairs; p++) {
        int64_t j = pj[p] - 1, k = pk[p] - 1;
        if (j < 0 || k < j || k >= cols) return PMO_ARGUMENT_ERROR;
        const double *aj = A + j * lda, *ak = A + k * lda;
        double s = 0.0;
        long double sl = 0.0L;
        if (j == k) {
            for (int64_t i = 0; i < rows; i++) { double pr = aj[i] * aj[i]; s = s + pr; sl += (long double)pr; }
            s = 2 * s; sl = 2 * sl;
        } else {
            for (int64_t i = 0; i < rows; i++) {
                double p1 = aj[i] * ak[i], p2 = ak[i] * aj[i];
                s = s + p1; s = s + p2;
                sl += (long double)p1; sl += (long double)p2;
            }
        }
        out[p] = s;
        if (out_ld) out_ld[p] = (double)sl;
    }
    return PMO_OK;
}

int pmo_canonical_lin_samples(const double *A, int64_t lda, int64_t rows, int64_t cols, const double *b, int sign, const int64_t *pj,
                              int64_t n, double *out, double *out_ld) {
    if (rows < 0 || cols < 0 || lda < rows) return PMO_DIMENSION_MISMATCH;
    for (int64_t p = 0; p < n; p++) {
        int64_t j = pj[p] - 1;
        if (j < 0 || j >= cols) return PMO_ARGUMENT_ERROR;
        const double *aj = A + j * lda;
        double s = 0.0;
        long double sl = 0.0L;
        for (int64_t i = 0; i < rows; i++) {
            double c = sign > 0 ? 0.0 + b[i] : (sign < 0 ? 0.0 - b[i] : 0.0);
            double pr = c * aj[i];
            s = s + pr; s = s + pr;
            sl += (long double)pr; sl += (long double)pr;
        }
        out[p] = s;
        if (out_ld) out_ld[p] = (double)sl;
    }
    return PMO_OK;
}
