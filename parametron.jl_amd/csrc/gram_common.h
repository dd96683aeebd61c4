// Geometry of the canonical least-squares contraction (gram_sk.hip), kept apart from the kernel bodies: argument block, tile geometry,
// work-unit numbering, the XCD-aware tile order, the accumulator lane map, the term store.
#pragma once
#include "common.h"

namespace pmt {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int ST = 128;            // output tile edge
constexpr int SKC = 256;           // contraction depth per work unit (the most; SKArgs::skc)
constexpr int MAXG = 512;          // upper bound on persistent workgroups
constexpr int SLOT = ST * ST;      // doubles per partial-tile slot
constexpr int MAXGROUPS = 16;     // stages (band ranges) of a host delivery

struct SKArgs {
    const double *A; int64_t lda, rows, cols;
    const int64_t *xvar; const int64_t *varmap; int moi;
    QT *out_quad;     // MOI / native QuadraticTerms at the canonical row-major upper-triangular position, or null
    double *out_csc;  // solver form: alpha * (MOI coefficient) at k(k+1)/2 + j (CSC of the upper triangle, values only), or null
    double alpha;
    int ntiles, nchunk, G;
    int skc;          // rows per (tile, chunk) work unit: SKC, or less where SKC-row units would leave most of the chip idle (launch_gram_sk)
    int tfull;        // whole tiles per workgroup (phase A); tiles [tfull*G, T) are split along the contraction (phase B)
    int64_t U;        // number of (tile, chunk) units of phase B = (T - tfull*G) * nchunk
    int vec_in;
    double *ws;
    // Tile order: 0 = super-rows of 4 tile rows (sk_seq_unrank); w > 0 = super-columns of w tile columns (sk_colseq_unrank): the column
    // bands of the output complete in ascending order, which is what a solver hand-off in CSC order wants to ship first.
    int order_w;
    int seq_begin;    // this launch covers the tiles seq_begin, seq_begin + seq_step, .. (T' of them) of the sequence (a staged host delivery
                      // launches the contraction band range by band range; 0 and all tiles otherwise)
    int seq_step;     // +1, or -1: the sequence is walked from its end (column bands complete in DESCENDING order: the small ones last)
    unsigned *pair_flags; unsigned epoch;      // pair fold of a ranged launch (gram_sk.hip), or null
    unsigned flag_value;                       // what a first half stores into its tile's flag: `epoch` (anything else only under fault injection)
    int *error;                                // page-locked error word: a second half whose partner never showed up stores 2 here
    long long pair_timeout;                    // bound of the pair wait in 100 MHz ticks
    int strict;                                // ranged launches: the sequence enumerates the STRICTLY upper tiles (jb < kb) only — the diagonal
                                               // tiles of a wide tall matrix are computed by gram_tall.hip
};

// TN = 16-column MFMA tiles per wave along N (4: 64x64 wave tile, 4 waves; 2: 64x32 wave tile, 8 waves)
template <int TN>
struct Cfg {
    static constexpr int NW = (TN == 4) ? 4 : 8;
    static constexpr int NT = NW * 64;
    static constexpr int WCOLS = 16 * TN;             // columns per wave
    static constexpr int NWC = ST / WCOLS;            // waves along N
    static constexpr int NACC = 4 * TN * 4;           // fp64 accumulators per lane
    static constexpr int NLD = (ST * 8) / NT;         // 16-byte pieces per thread per panel per 16 rows
};

__device__ __forceinline__ void sk_tri_unrank(int t, int nt, int &jb, int &kb) {
    int j = (int)((2.0 * nt + 1.0 - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)t)) * 0.5);
    if (j < 0) j = 0;
    if (j > nt - 1) j = nt - 1;
    while (j > 0 && (j * nt - j * (j - 1) / 2) > t) --j;
    while (j + 1 < nt && ((j + 1) * nt - (j + 1) * j / 2) <= t) ++j;
    jb = j;
    kb = j + (t - (j * nt - j * (j - 1) / 2));
}

// L2-friendly enumeration of the upper-triangular tile grid: super-rows of 4 tile rows, each traversed column by column
// (kb-major), so 32 consecutive tiles form a ~4 x 8 block that needs only ~12 distinct column panels.  Workgroups round-robin
// over the 8 XCDs (bid % 8), each XCD with its own L2: at every time step the 32 workgroups of one XCD get 32 CONSECUTIVE tiles
// of this sequence, so most panel reads hit the XCD's L2 instead of going to the fabric.
__device__ __forceinline__ void sk_seq_unrank(int idx, int nt, int &jb, int &kb) {
    int base = 0;
    for (int R = 0; R * 4 < nt; ++R) {
        const int j0 = R * 4;
        const int h = min(4, nt - j0);                  // tile rows in this super-row
        const int W = nt - j0;                          // columns (kb = j0 .. nt-1)
        const int tri = h * (h + 1) / 2;                // the first h columns hold 1, 2, .., h tiles
        const int count = tri + (W - h) * h;
        if (idx < base + count) {
            int p = idx - base;
            if (p < tri) {
                int c = 0;
                while (p >= c + 1) { p -= c + 1; ++c; }
                kb = j0 + c; jb = j0 + p;
            } else {
                p -= tri;
                kb = j0 + h + p / h; jb = j0 + p % h;
            }
            return;
        }
        base += count;
    }
    jb = kb = nt - 1;
}

// Column-band-major enumeration: super-columns of w tile columns, each traversed row by row (jb-major), so 32 consecutive tiles form a
// (32/w) x w block and every column band kb is complete once the sequence has passed its super-column.
__device__ __forceinline__ void sk_colseq_unrank(int idx, int nt, int w, int &jb, int &kb) {
    int base = 0;
    for (int c0 = 0; c0 < nt; c0 += w) {
        const int h = min(w, nt - c0);                  // tile columns in this super-column
        const int rect = c0 * h;                        // rows 0 .. c0-1 hold h tiles each; rows c0 .. c0+h-1 hold h, h-1, .., 1
        const int count = rect + h * (h + 1) / 2;
        if (idx < base + count) {
            int p = idx - base;
            if (p < rect) {
                jb = p / h; kb = c0 + p % h;
            } else {
                p -= rect;
                int r = 0;
                while (p >= h - r) { p -= h - r; ++r; }
                jb = c0 + r; kb = c0 + r + p;
            }
            return;
        }
        base += count;
    }
    jb = kb = nt - 1;
}

// idx: position in the tile sequence (a ranged launch adds its seq_begin itself)
__device__ __forceinline__ void sk_tile_unrank(const SKArgs &g, int idx, int &jb, int &kb) {
    if (g.order_w > 0) sk_colseq_unrank(idx, g.ntiles, g.order_w, jb, kb);
    else sk_seq_unrank(idx, g.ntiles, jb, kb);
}

// strictly upper tiles of the nt x nt grid = the upper-inclusive triangle of size nt - 1 shifted one tile column to the right (the same
// L2-friendly super-row order)
__device__ __forceinline__ void sk_tile_unrank_strict(const SKArgs &g, int idx, int &jb, int &kb) {
    sk_seq_unrank(idx, g.ntiles - 1, jb, kb);
    kb += 1;
}

// sequence index of the t-th whole tile of workgroup bid (phase A)
__device__ __forceinline__ int sk_phase_a_index(const SKArgs &g, int bid, int t) {
    if ((g.G & 7) == 0) return (t * 8 + (bid & 7)) * (g.G >> 3) + (bid >> 3);
    return bid * g.tfull + t;
}

__device__ __forceinline__ int64_t sk_unit_begin(const SKArgs &g, int b) { return (int64_t)b * g.U / g.G; }

// (row, col) inside the 128x128 tile of accumulator r = (tm*TN + tn)*4 + s of thread tid
template <int TN>
__device__ __forceinline__ void sk_acc_pos(int tid, int r, int &row, int &col) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / Cfg<TN>::NWC, wc = wave % Cfg<TN>::NWC;
    const int tm = r / (4 * TN), tn = (r / 4) % TN, s = r & 3;
    const int i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
    row = wr * 64 + tm * 16 + 4 * b + i;
    col = wc * Cfg<TN>::WCOLS + tn * 16 + 4 * ((b + s) & 3) + j;
}

// 2*acc -> QuadraticTerm at the canonical upper-triangular position
__device__ __forceinline__ void sk_store_term(const SKArgs &g, int jb, int kb, int row, int col, double v) {
    const int64_t n = g.cols;
    const int64_t j = (int64_t)jb * ST + row, k = (int64_t)kb * ST + col;
    if (k >= n || j >= n || j > k) return;
    double c = v;
    if (g.moi || j != k) c = 2 * c;          // off-diagonal: (j,k)+(k,j) combined; diagonal: MOI doubling (moi_interop.jl:58)
    if (g.out_csc) g.out_csc[k * (k + 1) / 2 + j] = g.alpha * c;
    if (!g.out_quad) return;
    const int64_t jv = g.xvar[j], kv = g.xvar[k];
    const int64_t pos = j * n - (j * (j - 1)) / 2 + (k - j);
    u64 *p = reinterpret_cast<u64 *>(g.out_quad) + pos * 3;
    p[0] = (u64)__double_as_longlong(c);
    p[1] = (u64)(g.moi ? map_var(g.varmap, jv) : jv);
    p[2] = (u64)(g.moi ? map_var(g.varmap, kv) : kv);
}

}  // namespace pmt
