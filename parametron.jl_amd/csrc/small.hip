// Small plans: a run of consecutive small tape entries executed by ONE launch.
//
// The reference evaluates a model's lazy-expression DAG by chasing C function pointers, nanoseconds per hop (src/lazyexpression.jl:50-61,
// src/FunctionWrappersQuickFix.jl:108-126): README Example 1 (n = 8, m = 2) re-evaluates in ~15 us on one CPU core (README.md:132-136,
// solve! 51.863 us including OSQP).  On the device every hop of the tape is a launch of ~5 us whatever its size, so the same model took
// 48.5 us (5 launches + the four Parameter callbacks).  Here the tape's nodes are DATA: one 1024-thread workgroup walks the node table in
// tape order — a grid-stride loop per node, a workgroup barrier between nodes (a later node reads what an earlier one wrote through the
// CU's own L1 / L2: workgroup-scope visibility is enough) — and one launch replaces the run.  Every element is computed by the same
// expression as in the node's stand-alone kernel (affine.hip, quad.hip, rng.hip): outputs are bit-identical, tests/test_gpu_small_plan.py.
#include <vector>

#include "common.h"
#include <cstring>

namespace pmt {

typedef unsigned long long u64;

__device__ __forceinline__ uint64_t sp_splitmix64(uint64_t z) {           // rng.hip
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double sp_uniform_at(uint64_t base, uint64_t i, double scale) {
    return scale * ((double)(sp_splitmix64(base + i) >> 11) * 0x1.0p-53);
}

struct SmallDyn { uint64_t v[SMALL_MAX_DYN]; };

// the device view of a node (the host-only tail of SmallNode is not uploaded)
struct SmallDev {
    int op, sign, moi, dyn;
    int sync, narrow;               // narrow != 0: a node of at most 256 elements — inside a phase such nodes are dealt out to the 16 waves (each
                                    // runs its node with its 64 lanes) instead of being walked one after the other by all 1024 threads: the
                                    // four Parameter callbacks of README Example 1 are four concurrent waves.  sync != 0: a workgroup barrier in front of this node (it touches what a node since the last barrier wrote, or
                                    // writes what one read); independent nodes — the four Parameter callbacks, the objective's and a constraint's
                                    // chains — share a phase
    int64_t d[4];
    const void *in[8];
    void *out[3];
    double scale;
    uint64_t seed;
};

// element indices and extents inside a small node are far below 2^31 (SMALL_NODE_WORK_MAX elements per node): 32-bit unsigned division — the
// 64-bit one is a ~200-instruction routine per element on this hardware, and most nodes divide once or twice per element
__device__ __forceinline__ int64_t qdiv(int64_t e, int64_t d) { return (int64_t)((uint32_t)e / (uint32_t)d); }

__device__ __forceinline__ void sp_node(const SmallDev &n, const uint64_t *dyn, int tid, int nt) {
    switch (n.op) {
    case SOP_FILL: {                       // rng.hip: dst[c * lda + i] = scale * U(seed, c * rows + i)
        const int64_t rows = n.d[0], cols = n.d[1], lda = n.d[2];
        const uint64_t base = (n.dyn >= 0 ? dyn[n.dyn] : n.seed) * 0x9E3779B97F4A7C15ull;
        double *dst = static_cast<double *>(n.out[0]);
        for (int64_t e = tid; e < rows * cols; e += nt) {
            const int64_t c = qdiv(e, rows), i = e - c * rows;
            dst[c * lda + i] = sp_uniform_at(base, (uint64_t)e, n.scale);
        }
        break;
    }
    case SOP_AFFINE_LT:
    case SOP_AFFINE_VAT: {                 // affine.hip: matvecmul! + vecadd!/vecsubtract! (+ update!(::MOI.VectorAffineFunction))
        const int64_t lda = n.d[0], rows = n.d[1], cols = n.d[2], row_offset = n.d[3];
        const double *A = static_cast<const double *>(n.in[0]);
        const int64_t *xvar = static_cast<const int64_t *>(n.in[1]);
        const double *b = static_cast<const double *>(n.in[2]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[3]);
        double *consts = static_cast<double *>(n.out[1]);
        for (int64_t e = tid; e < rows * cols; e += nt) {
            const int64_t c = qdiv(e, rows), r = e - c * rows;              // walk A column-major (coalesced reads)
            const double v = A[c * lda + r];
            if (n.op == SOP_AFFINE_LT) {
                LT t; t.coeff = v; t.var = xvar[c];
                static_cast<LT *>(n.out[0])[r * cols + c] = t;
            } else {
                VAT t; t.output_index = row_offset + r + 1; t.coeff = v; t.var = map_var(varmap, xvar[c]);
                static_cast<VAT *>(n.out[0])[r * cols + c] = t;
            }
        }
        if (consts)
            for (int64_t r = tid; r < rows; r += nt) consts[r] = signed_const(b ? b[r] : 0.0, b ? n.sign : 0);
        break;
    }
    case SOP_QUAD_EXPAND: {                // quad.hip: literal _vecdot!/muladd! (+ update!(::MOI.ScalarQuadraticFunction))
        const int64_t rows = n.d[0], nx = n.d[1], ny = n.d[2];
        const LT *x = static_cast<const LT *>(n.in[0]);
        const double *xc = static_cast<const double *>(n.in[1]);
        const LT *y = static_cast<const LT *>(n.in[2]);
        const double *yc = static_cast<const double *>(n.in[3]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[4]);
        QT *oq = static_cast<QT *>(n.out[0]);
        LT *ol = static_cast<LT *>(n.out[1]);
        for (int64_t e = tid; e < rows * nx * ny; e += nt) {
            const int64_t ia = qdiv(e, ny), k = e - ia * ny, i = qdiv(ia, nx);
            const LT xa = x[ia], yk = y[i * ny + k];
            double c = xa.coeff * yk.coeff;                                   // functions.jl:149
            if (n.moi && xa.var == yk.var) c = 2 * c;                        // moi_interop.jl:58
            QT t; t.coeff = c; t.row = n.moi ? map_var(varmap, xa.var) : xa.var; t.col = n.moi ? map_var(varmap, yk.var) : yk.var;
            oq[e] = t;
        }
        const int64_t w = nx + ny;
        for (int64_t e = tid; e < rows * w; e += nt) {
            const int64_t i = qdiv(e, w), k = e - i * w;
            LT t; double c;
            if (k < nx) { t = x[i * nx + k]; c = yc[i]; }                    // xlinear[i] * yconst  (:567)
            else { t = y[i * ny + (k - nx)]; c = xc[i]; }                    // ylinear[i] * xconst  (:571)
            LT o; o.coeff = c * t.coeff; o.var = n.moi ? map_var(varmap, t.var) : t.var;
            ol[e] = o;
        }
        if (tid == 0) {                                                      // strictly left to right (:574)
            double acc = 0.0;
            for (int64_t i = 0; i < rows; ++i) acc = acc + xc[i] * yc[i];
            *static_cast<double *>(n.out[2]) = acc;
        }
        break;
    }
    case SOP_VARS_ADDSUB: {                // affine.hip: x (+|-) v for x::Vector{Variable}
        const int64_t cnt = n.d[0], row_offset = n.d[1];
        const int64_t *xvar = static_cast<const int64_t *>(n.in[0]);
        const double *v = static_cast<const double *>(n.in[1]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[2]);
        LT *olt = static_cast<LT *>(n.out[0]);
        VAT *ovat = static_cast<VAT *>(n.out[1]);
        double *oc = static_cast<double *>(n.out[2]);
        for (int64_t i = tid; i < cnt; i += nt) {
            const int64_t var = xvar[i];
            if (olt) { LT t; t.coeff = 1.0; t.var = var; olt[i] = t; }
            if (ovat) { VAT t; t.output_index = row_offset + i + 1; t.coeff = 1.0; t.var = map_var(varmap, var); ovat[i] = t; }
            if (oc) oc[i] = signed_const(v ? v[i] : 0.0, v ? n.sign : 0);
        }
        break;
    }
    case SOP_CONSTS: {
        const double *d = static_cast<const double *>(n.in[0]);
        double *o = static_cast<double *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) o[i] = signed_const(d[i], n.sign);
        break;
    }
    case SOP_PACK_SA: {                    // moi_interop.jl:35-43
        const LT *in = static_cast<const LT *>(n.in[0]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[1]);
        LT *o = static_cast<LT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) { LT t = in[i]; t.var = map_var(varmap, t.var); o[i] = t; }
        break;
    }
    case SOP_PACK_SQ: {                    // moi_interop.jl:45-62
        const QT *in = static_cast<const QT *>(n.in[0]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[1]);
        QT *o = static_cast<QT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) {
            const QT t = in[i];
            QT r; r.coeff = (t.row == t.col) ? 2 * t.coeff : t.coeff; r.row = map_var(varmap, t.row); r.col = map_var(varmap, t.col);
            o[i] = r;
        }
        break;
    }
    case SOP_PACK_VA: {                    // moi_interop.jl:64-81; uniform rows (row_len) or ragged (row_ptr)
        const LT *in = static_cast<const LT *>(n.in[0]);
        const int64_t *row_ptr = static_cast<const int64_t *>(n.in[1]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[2]);
        const int64_t rows = n.d[0], row_len = n.d[1], row_offset = n.d[2];
        VAT *o = static_cast<VAT *>(n.out[0]);
        if (!row_ptr) {
            for (int64_t e = tid; e < rows * row_len; e += nt) {
                const LT t = in[e];
                VAT r; r.output_index = row_offset + qdiv(e, row_len) + 1; r.coeff = t.coeff; r.var = map_var(varmap, t.var);
                o[e] = r;
            }
        } else {
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;     // (scalar: a wave BRANCHES around the nodes of others)
            for (int64_t row = wave; row < rows; row += nw)
                for (int64_t e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += 64) {
                    const LT t = in[e];
                    VAT r; r.output_index = row_offset + row + 1; r.coeff = t.coeff; r.var = map_var(varmap, t.var);
                    o[e] = r;
                }
        }
        break;
    }
    case SOP_GRAM: {                       // gram.hip for tiny shapes: canonical least-squares objective, every sum in ROW order (the order of
                                           // the reference's literal expansion, src/functions.jl:705-707, before canonicalize! re-sorts it)
        const int64_t lda = n.d[0], rows = n.d[1], cols = n.d[2];
        const double *A = static_cast<const double *>(n.in[0]);
        const int64_t *xvar = static_cast<const int64_t *>(n.in[1]);
        const double *b = static_cast<const double *>(n.in[2]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[3]);
        QT *oq = static_cast<QT *>(n.out[0]);
        LT *ol = static_cast<LT *>(n.out[1]);
        const int64_t nq = cols * (cols + 1) / 2;
        for (int64_t e = tid; e < nq; e += nt) {
            // e -> (j, k), j <= k, row-major upper triangle
            int64_t j = (int64_t)((2.0 * cols + 1.0 - sqrt((2.0 * cols + 1.0) * (2.0 * cols + 1.0) - 8.0 * (double)e)) * 0.5);
            if (j < 0) j = 0;
            if (j > cols - 1) j = cols - 1;
            while (j > 0 && j * cols - j * (j - 1) / 2 > e) --j;
            while (j + 1 < cols && (j + 1) * cols - (j + 1) * j / 2 <= e) ++j;
            const int64_t k = j + (e - (j * cols - j * (j - 1) / 2));
            double acc = 0.0;
            for (int64_t i = 0; i < rows; ++i) acc = acc + A[j * lda + i] * A[k * lda + i];
            double c = acc;
            if (n.moi || j != k) c = 2 * c;
            const int64_t jv = xvar[j], kv = xvar[k];
            QT t; t.coeff = c; t.row = n.moi ? map_var(varmap, jv) : jv; t.col = n.moi ? map_var(varmap, kv) : kv;
            oq[e] = t;
        }
        for (int64_t j = tid; j < cols; j += nt) {
            double acc = 0.0;
            if (b && n.sign)
                for (int64_t i = 0; i < rows; ++i) acc = acc + signed_const(b[i], n.sign) * A[j * lda + i];
            const int64_t v = xvar[j];
            LT t; t.coeff = 2 * acc; t.var = n.moi ? map_var(varmap, v) : v;
            ol[j] = t;
        }
        if (tid == 0) {
            double acc = 0.0;
            if (b && n.sign)
                for (int64_t i = 0; i < rows; ++i) { const double c = signed_const(b[i], n.sign); acc = acc + c * c; }
            *static_cast<double *>(n.out[2]) = acc;
        }
        break;
    }
    case SOP_AFFVEC_COMBINE: {            // terms.hip: copyto!/add!/subtract!/vcat! on affine vectors, UNIFORM rows (functions.jl:422-427,455,477-485,969-994)
        const int64_t rows = n.d[0], la = n.d[1], lb = n.d[2], lo = n.d[3];
        const LT *xa = static_cast<const LT *>(n.in[0]);
        const double *ca = static_cast<const double *>(n.in[1]);
        const LT *xb = static_cast<const LT *>(n.in[2]);
        const double *cb = static_cast<const double *>(n.in[3]);
        LT *o = static_cast<LT *>(n.out[0]);
        double *oc = static_cast<double *>(n.out[1]);
        const int64_t na = xa ? la : 0, nb = xb ? lb : 0;
        for (int64_t e = tid; e < rows * lo; e += nt) {
            const int64_t row = qdiv(e, lo), k = e - row * lo;
            if (k < na) o[e] = xa[row * la + k];
            else if (k - na < nb) {
                LT t = xb[row * lb + (k - na)];
                if (n.sign < 0) t.coeff = -t.coeff;                          // -x.linear[i]  (functions.jl:481, :158)
                o[e] = t;
            }
        }
        if (oc)
            for (int64_t row = tid; row < rows; row += nt) {
                double c = ca ? ca[row] : 0.0;
                if (cb) c = n.sign < 0 ? c - cb[row] : c + cb[row];
                oc[row] = c;
            }
        break;
    }
    case SOP_AFFVEC_SCALE: {               // terms.hip: scale!/mul! of an affine vector by a number (functions.jl:515-523)
        const int64_t rows = n.d[0], nterms = n.d[1];
        const LT *y = static_cast<const LT *>(n.in[0]);
        const double *yc = static_cast<const double *>(n.in[1]);
        const double *sdev = static_cast<const double *>(n.in[2]);
        const double sc = sdev ? *sdev : n.scale;
        LT *o = static_cast<LT *>(n.out[0]);
        double *oc = static_cast<double *>(n.out[1]);
        for (int64_t i = tid; i < nterms + rows; i += nt) {
            if (i < nterms) { LT t = y[i]; t.coeff = sc * t.coeff; o[i] = t; }
            else { const int64_t r = i - nterms; oc[r] = 0.0 + yc[r] * sc; }
        }
        break;
    }
    case SOP_MATVEC_AFFS: {                // terms.hip: matvecmul!(y, A, ::Vector{AffineFunction}) (functions.jl:800-822)
        const int64_t lda = n.d[0], rows = n.d[1], cols = n.d[2], L = n.d[3];
        const double *A = static_cast<const double *>(n.in[0]);
        const LT *x = static_cast<const LT *>(n.in[1]);
        const double *xc = static_cast<const double *>(n.in[2]);
        LT *o = static_cast<LT *>(n.out[0]);
        double *oc = static_cast<double *>(n.out[1]);
        const int64_t per_row = cols * L;
        for (int64_t e = tid; e < rows * per_row; e += nt) {
            const int64_t row = qdiv(e, per_row), rem = e - row * per_row, col = qdiv(rem, L);
            LT t = x[rem];
            t.coeff = A[col * lda + row] * t.coeff;
            o[e] = t;
        }
        for (int64_t row = tid; row < rows; row += nt) {
            double acc = 0.0;
            for (int64_t col = 0; col < cols; ++col) { const double p = xc[col] * A[col * lda + row]; acc = acc + p; }
            oc[row] = acc;
        }
        break;
    }
    case SOP_VECDOT_NUM_VARS: {            // terms.hip: numbers . Vector{Variable} (functions.jl:684)
        const double *v = static_cast<const double *>(n.in[0]);
        const int64_t *xvar = static_cast<const int64_t *>(n.in[1]);
        LT *o = static_cast<LT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) { LT t; t.coeff = v[i]; t.var = xvar[i]; o[i] = t; }
        if (tid == 0 && n.out[1]) *static_cast<double *>(n.out[1]) = 0.0;
        break;
    }
    case SOP_VECDOT_NUM_AFFS: {            // terms.hip: numbers . Vector{AffineFunction} (functions.jl:524 -> :519, :521)
        const int64_t cnt = n.d[0], L = n.d[1];
        const double *v = static_cast<const double *>(n.in[0]);
        const LT *x = static_cast<const LT *>(n.in[1]);
        const double *xc = static_cast<const double *>(n.in[2]);
        LT *o = static_cast<LT *>(n.out[0]);
        for (int64_t e = tid; e < cnt * L; e += nt) { LT t = x[e]; t.coeff = v[qdiv(e, L)] * t.coeff; o[e] = t; }
        if (tid == 0) {
            double acc = 0.0;
            for (int64_t i = 0; i < cnt; ++i) acc = acc + xc[i] * v[i];        // launch_seq_dot(x_consts, v): left to right
            *static_cast<double *>(n.out[1]) = acc;
        }
        break;
    }
    case SOP_TRANSPOSE: {                  // terms.hip: the adjoint rule's closure, dest[j, i] = A[i, j] (lazyexpression.jl:206-217)
        const int64_t lds_ = n.d[0], rows = n.d[1], cols = n.d[2], ldd = n.d[3];
        const double *src = static_cast<const double *>(n.in[0]);
        double *dst = static_cast<double *>(n.out[0]);
        for (int64_t e = tid; e < rows * cols; e += nt) {
            const int64_t c = qdiv(e, rows), r = e - c * rows;
            dst[r * ldd + c] = src[c * lds_ + r];
        }
        break;
    }
    case SOP_QUAD_COMBINE: {               // terms.hip: [qa ; sb * qb] (functions.jl:434-439,459,492-500)
        const int64_t na = n.d[0], nb = n.d[1];
        const QT *qa = static_cast<const QT *>(n.in[0]);
        const QT *qb = static_cast<const QT *>(n.in[1]);
        QT *o = static_cast<QT *>(n.out[0]);
        for (int64_t i = tid; i < na + nb; i += nt) {
            QT t = i < na ? qa[i] : qb[i - na];
            if (i >= na && n.sign < 0) t.coeff = -t.coeff;
            o[i] = t;
        }
        break;
    }
    case SOP_QUAD_SCALE: {                 // terms.hip: s * q (functions.jl:526-534)
        const QT *q = static_cast<const QT *>(n.in[0]);
        const double *sdev = static_cast<const double *>(n.in[1]);
        const double sc = sdev ? *sdev : n.scale;
        QT *o = static_cast<QT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) { QT t = q[i]; t.coeff = sc * t.coeff; o[i] = t; }
        break;
    }
    case SOP_SCALE_VARS: {                 // terms.hip: scale!(dest::Vector{LinearTerm}, x, y::Vector{Variable}) (functions.jl:873-893)
        const int64_t *yvar = static_cast<const int64_t *>(n.in[0]);
        const double *sdev = static_cast<const double *>(n.in[1]);
        const double sc = sdev ? *sdev : n.scale;
        LT *o = static_cast<LT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) { LT t; t.coeff = sc; t.var = yvar[i]; o[i] = t; }
        break;
    }
    case SOP_SCALE_NUMBERS: {              // terms.hip: dest .= x .* y (functions.jl:917-925)
        const double *y = static_cast<const double *>(n.in[0]);
        const double *sdev = static_cast<const double *>(n.in[1]);
        const double sc = sdev ? *sdev : n.scale;
        double *o = static_cast<double *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) o[i] = sc * y[i];
        break;
    }
    case SOP_BILINEAR: {                   // quad.hip: bilinearmul! (functions.jl:840-858): term r * ny + k = (Q[lin], x[r], y[k]), lin = r * ny + k column-major
        const int64_t ldq = n.d[0], nxr = n.d[1], ny = n.d[2];
        const double *Q = static_cast<const double *>(n.in[0]);
        const int64_t *xvar = static_cast<const int64_t *>(n.in[1]);
        const int64_t *yvar = static_cast<const int64_t *>(n.in[2]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[3]);
        QT *o = static_cast<QT *>(n.out[0]);
        for (int64_t e = tid; e < nxr * ny; e += nt) {
            const int64_t r = qdiv(e, ny), k = e - r * ny;
            const int64_t qcol = qdiv(e, nxr), qrow = e - qcol * nxr;                   // column-major linear index of the nxr x ny matrix (:853)
            double c = Q[qcol * ldq + qrow];
            const int64_t xv = xvar[r], yv = yvar[k];
            if (n.moi && xv == yv) c = 2 * c;
            QT t; t.coeff = c; t.row = n.moi ? map_var(varmap, xv) : xv; t.col = n.moi ? map_var(varmap, yv) : yv;
            o[e] = t;
        }
        break;
    }
    case SOP_VECDOT_TERMS: {               // quad.hip: Variable/LinearTerm . Variable/LinearTerm (functions.jl:689-700, :146-149)
        const double *xc = static_cast<const double *>(n.in[0]);
        const int64_t *xvar = static_cast<const int64_t *>(n.in[1]);
        const double *yc = static_cast<const double *>(n.in[2]);
        const int64_t *yvar = static_cast<const int64_t *>(n.in[3]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[4]);
        QT *o = static_cast<QT *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) {
            const double a = xc ? xc[i] : 1.0, b = yc ? yc[i] : 1.0;
            double c = a * b;
            const int64_t xv = xvar[i], yv = yvar[i];
            if (n.moi && xv == yv) c = 2 * c;
            QT t; t.coeff = c; t.row = n.moi ? map_var(varmap, xv) : xv; t.col = n.moi ? map_var(varmap, yv) : yv;
            o[i] = t;
        }
        break;
    }
    case SOP_VECDOT_AFFS_VARS: {           // quad.hip: Vector{AffineFunction} . Vector{Variable} (functions.jl:702-709 over :537-546)
        const int64_t rows = n.d[0], L = n.d[1];
        const LT *x = static_cast<const LT *>(n.in[0]);
        const double *xc = static_cast<const double *>(n.in[1]);
        const int64_t *yvar = static_cast<const int64_t *>(n.in[2]);
        const int64_t *varmap = static_cast<const int64_t *>(n.in[3]);
        QT *oq = static_cast<QT *>(n.out[0]);
        LT *ol = static_cast<LT *>(n.out[1]);
        for (int64_t e = tid; e < rows * L + rows; e += nt) {
            if (e < rows * L) {
                const int64_t i = qdiv(e, L);
                const LT t = x[e];
                const int64_t yv = yvar[i];
                QT q; q.coeff = (n.moi && t.var == yv) ? 2 * t.coeff : t.coeff;
                q.row = n.moi ? map_var(varmap, t.var) : t.var; q.col = n.moi ? map_var(varmap, yv) : yv;
                oq[e] = q;
            } else {
                const int64_t i = e - rows * L;
                LT l; l.coeff = xc[i]; l.var = n.moi ? map_var(varmap, yvar[i]) : yvar[i];
                ol[i] = l;
            }
        }
        break;
    }
    case SOP_COPY8: {
        const u64 *src = static_cast<const u64 *>(n.in[0]);
        u64 *dst = static_cast<u64 *>(n.out[0]);
        for (int64_t i = tid; i < n.d[0]; i += nt) dst[i] = src[i];
        break;
    }
    default: break;
    }
}

constexpr int SMALL_MAX_NODES = 48;          // nodes of one launch (their descriptions are fetched into LDS in one round trip)

// syncmask / narrowmask: bit k = node k's `sync` / `narrow` flag, as kernel arguments (scalar registers): a wave skips the nodes that are
// not its own without touching their descriptions (an LDS read and a wait per skipped node was ~0.3 us)
// SEVERAL workgroups (plans whose phases hold tens of thousands of elements: a model of ~100 variables kept ONE CU busy for 21 us): every
// node's element loop strides over all of them, narrow nodes are dealt to the waves of all, and the barrier in front of a dependent node
// becomes a GRID barrier — one monotonic counter per plan (`bar`, zeroed when the table is uploaded; the launches of a plan are
// serialised on its stream, launch e waits for the values base + W, base + 2 W, .. with base = e * barriers * W): workgroup barrier,
// lane 0 releases the workgroup's writes at agent scope and arrives, polls (bounded), acquires; every wave acquires behind the closing
// workgroup barrier.  At most 32 workgroups: co-resident on any MI355X partition, so the poll cannot starve its own grid.
__device__ __forceinline__ void small_grid_barrier(unsigned long long *bar, unsigned long long target, int *error, long long bound, int tid) {
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > bound) {                // 2 s of 100 MHz ticks: report, do not hang
                if (error) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (pmt_plan_synchronize reports it)
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// syncmask / narrowmask: bit k = node k's `sync` / `narrow` flag, as kernel arguments (scalar registers): a wave skips the nodes that are
// not its own without touching their descriptions (an LDS read and a wait per skipped node was ~0.3 us)
__global__ __launch_bounds__(1024) void small_plan_kernel(const SmallDev *__restrict__ table, int count, SmallDyn dyn, unsigned long long syncmask,
                                                          unsigned long long narrowmask, unsigned long long *bar, unsigned long long base, int *error, long long bound) {
    __shared__ SmallDev nodes[SMALL_MAX_NODES];
    __shared__ uint64_t sdyn[SMALL_MAX_DYN];
    const int tid = threadIdx.x, nt = blockDim.x;
    // (static indices: a kernel argument array indexed by a register would be copied to scratch)
#pragma unroll
    for (int i = 0; i < SMALL_MAX_DYN; ++i)
        if (tid == i) sdyn[i] = dyn.v[i];
    // every node description in ONE round trip (a description per node and barrier was a dependent global load per hop: 9 hops of README
    // Example 1 took 10.6 us of kernel time for ~700 terms of work)
    constexpr int W8 = (int)(sizeof(SmallDev) / 8);
    for (int i = tid; i < count * W8; i += nt) reinterpret_cast<u64 *>(nodes)[i] = reinterpret_cast<const u64 *>(table)[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;     // (scalar: a wave BRANCHES around the nodes of others)
    const int wgs = (int)gridDim.x;
    const int nw = 16 * wgs, gwave = (int)blockIdx.x * 16 + wave;                    // (a workgroup is always 1024 threads)
    const int gtid = (int)blockIdx.x * nt + tid, gnt = wgs * nt;
    int slot = 0;                                  // narrow nodes of the current phase seen so far
    unsigned long long target = base;
    for (int k = 0; k < count; ++k) {
        if ((syncmask >> k) & 1) {                 // (uniform: what the previous phase wrote is visible before this one reads it)
            if (wgs == 1) __syncthreads();
            else { target += (unsigned long long)wgs; small_grid_barrier(bar, target, error, bound, tid); }
            slot = 0;
        }
        if ((narrowmask >> k) & 1) {
            if (slot % nw == gwave) sp_node(nodes[k], sdyn, lane, 64);
            ++slot;
        } else {
            sp_node(nodes[k], sdyn, gtid, gnt);
        }
    }
}

int small_max_nodes() { return SMALL_MAX_NODES; }

// ---- host: which nodes need a barrier in front of them -----------------------------------------------------------------------------------
// Byte ranges a node reads / writes, from its shapes (the static index arrays — xvar, varmap, row_ptr — are never written by a node and are
// left out).  An op this function does not know makes every later node synchronise (conservative).
struct SmallRange { const char *p; size_t n; };
static bool small_node_io(const SmallNode &s, std::vector<SmallRange> &rd, std::vector<SmallRange> &wr) {
    auto R = [&](const void *p, int64_t bytes) { if (p && bytes > 0) rd.push_back({static_cast<const char *>(p), (size_t)bytes}); };
    auto W = [&](void *p, int64_t bytes) { if (p && bytes > 0) wr.push_back({static_cast<const char *>(p), (size_t)bytes}); };
    auto mat = [](int64_t ld, int64_t rows, int64_t cols) { return cols > 0 ? ((cols - 1) * ld + rows) * 8 : 0; };
    const int64_t *d = s.d;
    switch (s.op) {
    case SOP_FILL: W(s.out[0], mat(d[2], d[0], d[1])); return true;
    case SOP_AFFINE_LT: case SOP_AFFINE_VAT:
        R(s.in[0], mat(d[0], d[1], d[2])); R(s.in[2], d[1] * 8);
        W(s.out[0], d[1] * d[2] * (s.op == SOP_AFFINE_LT ? 16 : 24)); W(s.out[1], d[1] * 8); return true;
    case SOP_QUAD_EXPAND:
        R(s.in[0], d[0] * d[1] * 16); R(s.in[1], d[0] * 8); R(s.in[2], d[0] * d[2] * 16); R(s.in[3], d[0] * 8);
        W(s.out[0], d[0] * d[1] * d[2] * 24); W(s.out[1], d[0] * (d[1] + d[2]) * 16); W(s.out[2], 8); return true;
    case SOP_VARS_ADDSUB: R(s.in[1], d[0] * 8); W(s.out[0], d[0] * 16); W(s.out[1], d[0] * 24); W(s.out[2], d[0] * 8); return true;
    case SOP_CONSTS: R(s.in[0], d[0] * 8); W(s.out[0], d[0] * 8); return true;
    case SOP_PACK_SA: R(s.in[0], d[0] * 16); W(s.out[0], d[0] * 16); return true;
    case SOP_PACK_SQ: R(s.in[0], d[0] * 24); W(s.out[0], d[0] * 24); return true;
    case SOP_PACK_VA: if (s.in[1]) return false; R(s.in[0], d[0] * d[1] * 16); W(s.out[0], d[0] * d[1] * 24); return true;
    case SOP_COPY8: R(s.in[0], d[0] * 8); W(s.out[0], d[0] * 8); return true;
    case SOP_GRAM: R(s.in[0], mat(d[0], d[1], d[2])); R(s.in[2], d[1] * 8);
        W(s.out[0], d[2] * (d[2] + 1) / 2 * 24); W(s.out[1], d[2] * 16); W(s.out[2], 8); return true;
    case SOP_AFFVEC_COMBINE: R(s.in[0], d[0] * d[1] * 16); R(s.in[1], d[0] * 8); R(s.in[2], d[0] * d[2] * 16); R(s.in[3], d[0] * 8);
        W(s.out[0], d[0] * d[3] * 16); W(s.out[1], d[0] * 8); return true;
    case SOP_AFFVEC_SCALE: R(s.in[0], d[1] * 16); R(s.in[1], d[0] * 8); R(s.in[2], 8); W(s.out[0], d[1] * 16); W(s.out[1], d[0] * 8); return true;
    case SOP_MATVEC_AFFS: R(s.in[0], mat(d[0], d[1], d[2])); R(s.in[1], d[2] * d[3] * 16); R(s.in[2], d[2] * 8);
        W(s.out[0], d[1] * d[2] * d[3] * 16); W(s.out[1], d[1] * 8); return true;
    case SOP_VECDOT_NUM_VARS: R(s.in[0], d[0] * 8); W(s.out[0], d[0] * 16); W(s.out[1], 8); return true;
    case SOP_VECDOT_NUM_AFFS: R(s.in[0], d[0] * 8); R(s.in[1], d[0] * d[1] * 16); R(s.in[2], d[0] * 8); W(s.out[0], d[0] * d[1] * 16); W(s.out[1], 8); return true;
    case SOP_TRANSPOSE: R(s.in[0], mat(d[0], d[1], d[2])); W(s.out[0], mat(d[3], d[2], d[1])); return true;
    case SOP_QUAD_COMBINE: R(s.in[0], d[0] * 24); R(s.in[1], d[1] * 24); W(s.out[0], (d[0] + d[1]) * 24); return true;
    case SOP_QUAD_SCALE: R(s.in[0], d[0] * 24); R(s.in[1], 8); W(s.out[0], d[0] * 24); return true;
    case SOP_SCALE_VARS: R(s.in[1], 8); W(s.out[0], d[0] * 16); return true;
    case SOP_SCALE_NUMBERS: R(s.in[0], d[0] * 8); R(s.in[1], 8); W(s.out[0], d[0] * 8); return true;
    case SOP_BILINEAR: R(s.in[0], mat(d[0], d[1], d[2])); W(s.out[0], d[1] * d[2] * 24); return true;
    case SOP_VECDOT_TERMS: R(s.in[0], d[0] * 8); R(s.in[2], d[0] * 8); W(s.out[0], d[0] * 24); return true;
    case SOP_VECDOT_AFFS_VARS: R(s.in[0], d[0] * d[1] * 16); R(s.in[1], d[0] * 8); W(s.out[0], d[0] * d[1] * 24); W(s.out[1], d[0] * 16); return true;
    default: return false;
    }
}

// sets nodes[k].sync: 0 for the first node and for a node that neither touches what a node of the current phase wrote nor writes what one
// read; 1 otherwise (and a new phase begins).  Returns the number of phases.
int small_plan_phases(SmallNode *nodes, int count) {
    std::vector<SmallRange> prd, pwr;          // reads / writes of the current phase
    auto overlap = [](const SmallRange &a, const SmallRange &b) { return a.p < b.p + b.n && b.p < a.p + a.n; };
    auto any = [&](const std::vector<SmallRange> &x, const std::vector<SmallRange> &y) {
        for (auto &a : x) for (auto &b : y) if (overlap(a, b)) return true;
        return false;
    };
    int phases = count > 0 ? 1 : 0;
    for (int k = 0; k < count; ++k) {
        std::vector<SmallRange> rd, wr;
        const bool known = small_node_io(nodes[k], rd, wr);
        const bool dep = !known || any(rd, pwr) || any(wr, pwr) || any(wr, prd);
        nodes[k].sync = (k > 0 && dep) ? 1 : 0;
        if (nodes[k].sync) { prd.clear(); pwr.clear(); ++phases; }
        if (!known) { nodes[k].sync = k > 0 ? 1 : 0; prd.push_back({nullptr, ~(size_t)0}); pwr.push_back({nullptr, ~(size_t)0}); }   // everything after it synchronises too
        prd.insert(prd.end(), rd.begin(), rd.end());
        pwr.insert(pwr.end(), wr.begin(), wr.end());
    }
    return phases;
}

// the flag masks of a group (after small_plan_phases)
void small_plan_masks(const SmallNode *nodes, int count, unsigned long long *syncmask, unsigned long long *narrowmask) {
    *syncmask = *narrowmask = 0;
    for (int k = 0; k < count && k < 64; ++k) {
        if (nodes[k].sync) *syncmask |= 1ull << k;
        if (nodes[k].work <= 256) *narrowmask |= 1ull << k;
    }
}

size_t small_table_bytes(int count) { return sizeof(SmallDev) * (size_t)count; }

// host: the device image of `count` nodes (the caller uploads it once)
void small_table_image(const SmallNode *nodes, int count, void *image) {
    SmallDev *d = static_cast<SmallDev *>(image);
    for (int i = 0; i < count; ++i) {
        const SmallNode &s = nodes[i];
        d[i].op = s.op; d[i].sign = s.sign; d[i].moi = s.moi; d[i].dyn = s.dyn; d[i].sync = s.sync; d[i].narrow = (s.work <= 256) ? 1 : 0;
        for (int k = 0; k < 4; ++k) d[i].d[k] = s.d[k];
        for (int k = 0; k < 8; ++k) d[i].in[k] = s.in[k];
        for (int k = 0; k < 3; ++k) d[i].out[k] = s.out[k];
        d[i].scale = s.scale; d[i].seed = s.seed;
    }
}

// ONE node by itself, its description a kernel argument (no table): the stand-alone form of a node that has no kernel of its own at its
// size — the tiny canonical Gram objective (gram.hip), whose other form, the stream-K node, is three kernels and a side-stream fork.
struct SmallRaw { u64 w[sizeof(SmallDev) / 8]; };      // (a description as plain words: a kernel argument struct that holds pointers sends this compiler's infer-address-spaces pass into a crash)
__global__ __launch_bounds__(1024) void small_one_kernel(SmallRaw d) {
    __shared__ SmallDev node;                          // (sp_node indexes the description's arrays with registers: not from kernel arguments)
    constexpr int W8 = (int)(sizeof(SmallDev) / 8);
#pragma unroll
    for (int i = 0; i < W8; ++i)
        if ((int)threadIdx.x == i) reinterpret_cast<u64 *>(&node)[i] = d.w[i];
    __shared__ uint64_t sdyn[SMALL_MAX_DYN];           // (no host-word seeds here; sp_node wants the array)
    if (threadIdx.x < SMALL_MAX_DYN) sdyn[threadIdx.x] = 0;
    __syncthreads();
    sp_node(node, sdyn, (int)threadIdx.x, (int)blockDim.x);
}

int launch_small_one(const SmallNode &nd, hipStream_t s) {
    SmallNode one = nd;
    one.dyn = -1; one.sync = 0;
    SmallDev d;
    small_table_image(&one, 1, &d);
    SmallRaw raw;
    std::memcpy(&raw, &d, sizeof raw);
    PMT_LAUNCH(small_one_kernel, dim3(1), dim3(1024), 0, s, raw);
    return check_launch("small_one_kernel");
}

int launch_small_plan(const void *device_table, int count, const uint64_t *const *seed_words, int ndyn, unsigned long long syncmask,
                      unsigned long long narrowmask, int workgroups, unsigned long long *barrier_word, unsigned long long barrier_base, int *barrier_error,
                      long long barrier_bound, hipStream_t s) {
    SmallDyn dyn;
    for (int i = 0; i < SMALL_MAX_DYN; ++i) dyn.v[i] = (i < ndyn && seed_words[i]) ? *seed_words[i] : 0;
    PMT_LAUNCH(small_plan_kernel, dim3((unsigned)std::max(1, workgroups)), dim3(1024), 0, s, static_cast<const SmallDev *>(device_table), count, dyn, syncmask,
               narrowmask, barrier_word, barrier_base, barrier_error, barrier_bound);
    return check_launch("small_plan_kernel");
}

// workgroups of a fused run: one up to 8192 elements of work (README Example 1: ~700; its launch is then the single-workgroup kernel of
// before, bit for bit and cycle for cycle), one more per 4096 beyond, 32 at most
// workgroups of small_plan_kernel one CU holds (the occupancy calculator's answer for 1024 threads and the kernel's LDS / registers)
int small_plan_occupancy() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, small_plan_kernel, 1024, 0) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return std::max(1, n);
}

int small_plan_workgroups(int64_t work) { return (int)std::min<int64_t>(32, std::max<int64_t>(1, work / 4096)); }

}  // namespace pmt
