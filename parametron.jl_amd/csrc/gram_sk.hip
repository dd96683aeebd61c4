// Canonical least-squares objective, stream-K form: the upper triangle of 2*A'A on v_mfma_f64_4x4x4_4b_f64.
//
// Why this shape (measured on MI355X, profiles/r01a_*, r01b_*):
//   * v_mfma_f64_4x4x4_4b_f64 issues every 16-17 cycles (72.5 TFLOP/s chip-wide), v_mfma_f64_16x16x4_f64 every 138
//     (34.9 TFLOP/s) — the 4-block form is the full-rate f64 matrix instruction on gfx950.
//   * 528 tiles of 128x128 on 256 CUs quantise to 3 rounds of 2.06 needed; a persistent grid that splits the
//     contraction (stream-K) gives every workgroup the same number of (tile, K-chunk) units.
//   * PMC (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) showed the matrix pipe idle whenever the waves sharing a SIMD sit
//     at the same barrier; variant "wg256" therefore runs TWO independent 4-wave workgroups per CU (one wave per SIMD
//     each, 64x64 wave tiles) so that one workgroup's load/store/barrier phase overlaps the other's MFMA phase, while
//     "wg512" runs one 8-wave workgroup (64x32 wave tiles).
// Work unit = (tile, K-chunk of KC rows).  Units are numbered tile-major and dealt out in contiguous, equal ranges to
// G persistent workgroups.  A workgroup that covers all chunks of a tile writes the QuadraticTerms directly (fused
// epilogue: x2, canonical upper-triangular position, varmap); otherwise it stores its partial accumulators in a
// workspace slot and `gram_sk_fixup_kernel` adds the partials of each split tile in ascending workgroup order
// (deterministic) and writes the terms.  At the BASELINE size (n = r = 4096: 528 tiles x 16 chunks = 8448 units, 33 or
// 16.5 per workgroup) every tile is touched by at most two (three) workgroups.
//
// Operand lane maps of the 4x4x4_4b form were probed on hardware (tools/mfma_probe.hip):
//   A: lane = i + 4b + 16k    B: lane = j + 4b + 16k    D: lane = j + 4b + 16i     (block b, 4x4 tile, k = 0..3)
// i.e. A/B registers look like those of the 16x16x4 form (row|col = lane & 15, k = lane >> 4); block b pairs row group b
// with column group b, so the 16 (row group, column group) pairs of a 16x16 tile take 4 instructions whose B operand is read
// from LDS with the column groups rotated by s = 0..3 (cbsz/abid broadcast is ignored for f64: tools/mfma_probe2.hip).
// acc[tm][tn][s] of lane l = C[16tm + 4b + i][16tn + 4((b+s)&3) + j], i = l>>4, b = (l>>2)&3, j = l&3.
#include "dma.h"
#include "gram_common.h"

// The pair fold's producer side (below) publishes a partial tile with write-through (sc1) stores + s_waitcnt vmcnt(0) + a relaxed flag
// store: the hand-off cdna_hip_programming.md describes for gfx942 / gfx950 as its second, cheaper recipe ("sc1 slab stores -> every wave
// s_waitcnt vmcnt(0) -> __syncthreads() -> relaxed agent-scope flag; the reducer reads behind an acquire fence") — a property of THESE
// targets' write-through stores, not of the HIP memory model.  Any other target gets the model's own release edge (the FORMAL variant,
// +2.6 % per host_csc solve in profiles/r05_pair_fold.txt); -DPMT_SK_PAIR_FORMAL=1 forces it here too.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(PMT_SK_PAIR_FORMAL)
#define PMT_SK_PAIR_FORMAL 1
#endif

// Codegen knobs.  The compiler's schedule of the stage loop moves by +-10 % with source changes that do not touch the loop.  Rounds 1-2
// shipped the luckiest draw of a sweep (246 VGPRs, 1.177 ms at n = r = 4096) — an allocation that only came out that way while a second,
// unrelated instantiation shared the translation unit (alone: 256 + 7 spilled).  Round 3 takes the lottery out of it in two steps
// (profiles/r03_gram_codegen.txt): (1) the epilogue derives its lane-dependent offsets from a FRESH copy of threadIdx.x (sk_fresh_tid),
// so nothing of it is live across the stage loop: the kernel's natural register need drops from ~263 to 183-231; (2) the register
// budget is stated in the source (amdgpu_num_vgpr) and the scheduler works towards it.  Measured over budget x LOADKS x ORDER:
//   budget 248/232: 1.38 ms;  budget 216 and below (183-187 VGPRs allocated): 1.185-1.20 ms;  spills: none in any of them
//   PMT_SK_ORDER 0   MFMA issue order (tn, tm, r)   (1: (tn, r, tm), consecutive MFMAs share the B operand: 1.198 ms)
//   PMT_SK_LOADKS 1  the next stage's global loads are issued after the first k-step   (0: 1.26, 2: 1.23, 3: 1.26 ms)
// gram_sk_kernel at n = r = 4096: 1.186 ms with 183 VGPRs — two waves per SIMD leave 146 of its 512 registers to co-resident kernels.
#ifndef PMT_GRAM_SK_STAGGER
#define PMT_GRAM_SK_STAGGER 0
#endif
#ifndef PMT_SK_LOADKS
#define PMT_SK_LOADKS 1
#endif
#ifndef PMT_SK_ORDER
#define PMT_SK_ORDER 0
#endif
#ifndef PMT_SK_STOREKS
#define PMT_SK_STOREKS -1      // k-step in front of which the next stage's panels go from registers to LDS (-1: behind the last k-step)
#endif
#ifndef PMT_SK_STOREFENCE
#define PMT_SK_STOREFENCE 0    // 1: a scheduling barrier behind those LDS stores (they may not sink to the end of the stage)
#endif
#ifndef PMT_SK_BK
#define PMT_SK_BK 16          // rows per stage (one barrier per stage)
#endif
#ifndef PMT_SK_WPS
#define PMT_SK_WPS 2          // __launch_bounds__ waves-per-SIMD hint of the shipped instantiation
#endif

namespace pmt {

// Tile epilogue through LDS: accumulators -> smem[64][129] (half a tile at a time) -> row-contiguous QuadraticTerm runs.
// For output row j the tile's entries k = max(j, k0) .. k0+127 are consecutive terms of the canonical upper triangle, so
// each wave writes whole row segments as 16-byte chunks (global_store_dwordx4) instead of three scattered 8-byte stores
// per element; the mapped variable indices of the tile's 128 columns / 64 rows are gathered once into LDS.
// smem must hold 64*129 + 192 doubles; all threads of the workgroup call this together.
constexpr int EPITCH = 129;
constexpr int EPI_DOUBLES = 64 * EPITCH + 192;

// threadIdx.x through an opaque move: everything derived from the returned value is computed AFTER this point.  The epilogue needs a dozen
// lane-dependent offsets; derived from the kernel's one `tid` they are hoisted above the stage loop and held (or spilled) across it.
__device__ __forceinline__ int sk_fresh_tid() {
    int t;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"((int)threadIdx.x));
    return t;
}

// hsel >= 0: only the row half hsel of the tile is written (balanced pair fold: each of the two workgroups of a split tile finishes one half)
template <int TN>
__device__ __forceinline__ void sk_epilogue(const SKArgs &g, int jb, int kb, const double (&acc)[Cfg<TN>::NACC], double *smem, int, int hsel = -1) {
    using C = Cfg<TN>;
#if defined(PMT_SK_EPI_ABL) && PMT_SK_EPI_ABL == 3
    {   // ablation: no epilogue at all (the accumulators stay live through a store that never happens): what a FREE write-out would give
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < C::NACC; ++r) sum += acc[r];
        if (sum == 1.2345e300 && g.out_quad) reinterpret_cast<double *>(g.out_quad)[threadIdx.x] = sum;
        return;
    }
#endif
    const int tid = sk_fresh_tid();
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const int64_t n = g.cols;
    const int64_t j0 = (int64_t)jb * ST, k0 = (int64_t)kb * ST;
    double *tile = smem;
    u64 *cmap = reinterpret_cast<u64 *>(smem + 64 * EPITCH);
    u64 *rmap = cmap + 128;
    const int wave = tid >> 6, lane = tid & 63;
    const int wr = wave / C::NWC;
    u64 *out = reinterpret_cast<u64 *>(g.out_quad);
    for (int h = 0; h < 2; ++h) {
        if (hsel >= 0 && h != hsel) continue;
        __syncthreads();
        if (g.out_quad && (h == 0 || hsel >= 0) && tid < 128) {
            const int64_t k = k0 + tid;
            const int64_t kv = k < n ? g.xvar[k] : 1;
            cmap[tid] = (u64)(g.moi ? map_var(g.varmap, kv) : kv);
        }
        if (g.out_quad && tid >= 128 && tid < 192) {
            const int64_t j = j0 + h * 64 + (tid - 128);
            const int64_t jv = j < n ? g.xvar[j] : 1;
            rmap[tid - 128] = (u64)(g.moi ? map_var(g.varmap, jv) : jv);
        }
        if (wr == h) {
#pragma unroll
            for (int r = 0; r < C::NACC; ++r) {
                int row, col;
                sk_acc_pos<TN>(tid, r, row, col);
                double c = acc[r];
                if (g.moi || (j0 + row) != (k0 + col)) c = 2 * c;   // off-diagonal: (j,k)+(k,j) combined; diagonal: MOI doubling
                tile[(row - h * 64) * EPITCH + col] = c;
            }
        }
        __syncthreads();
        if (g.out_csc) {
            // column k of the tile holds rows j0+64h .. of CSC column k: 64 lanes = 64 consecutive doubles (LDS pitch 129: conflict-free)
            const int64_t j = j0 + h * 64 + lane;
            for (int col = wave; col < ST; col += C::NW) {
                const int64_t k = k0 + col;
                if (k >= n) break;
                if (j <= k) {
                    const double v = g.alpha * tile[lane * EPITCH + col];
                    g.out_csc[k * (k + 1) / 2 + j] = v;
                }
            }
        }
        for (int row = wave; g.out_quad && row < 64; row += C::NW) {
            const int64_t j = j0 + h * 64 + row;
            if (j >= n) break;
            const int64_t kstart = j > k0 ? j : k0;
            const int64_t kend = (k0 + ST < n) ? k0 + ST : n;
            const int nterms = (int)(kend - kstart);
            if (nterms <= 0) continue;
            const int coff = (int)(kstart - k0);
            const int64_t term0 = j * n - (j * (j - 1)) / 2 + (kstart - j);
            u64 *seg = out + term0 * 3;
            const int nwords = nterms * 3;
            const int lead = (int)((reinterpret_cast<uintptr_t>(seg) >> 3) & 1);
            const double *trow = tile + row * EPITCH + coff;
            const u64 rv = rmap[row];
            auto word = [&](int q) -> u64 {
#if defined(PMT_SK_EPI_ABL) && PMT_SK_EPI_ABL == 1
                return (u64)q + rv;                          // ablation: no LDS reads, no selects
#endif
                const int t = q / 3, f = q - 3 * t;
                return f == 0 ? (u64)__double_as_longlong(trow[t]) : (f == 1 ? rv : cmap[coff + t]);
            };
            if (lead && lane == 0) seg[0] = word(0);
            for (int c = lane; lead + 2 * c < nwords; c += 64) {
                const int q0 = lead + 2 * c;
                if (q0 + 1 < nwords) {
                    u64x2 v;
                    v.x = word(q0);
                    v.y = word(q0 + 1);
#if defined(PMT_SK_EPI_ABL) && PMT_SK_EPI_ABL == 2
                    if (v.x == 0x7ff8dead7ff8deadull) seg[q0] = v.y;          // ablation: (practically) no global stores
#else
#if defined(PMT_SK_EPI_NT) && PMT_SK_EPI_NT
                    __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(seg + q0));      // tuning: the term array is written once and never re-read here
#else
                    *reinterpret_cast<u64x2 *>(seg + q0) = v;
#endif
#endif
                } else {
                    seg[q0] = word(q0);
                }
            }
        }
    }
}

template <int TN, int BK, bool fast>
__device__ __forceinline__ void sk_load_panel(const SKArgs &g, int64_t c0, int64_t i0, int64_t iend,
                                              f64x2 (&reg)[Cfg<TN>::NLD * (BK / 16)], int tid) {
    constexpr int KP = BK / 2;                         // 16-byte pieces per column
    constexpr int CPP = Cfg<TN>::NT / KP;              // columns covered per pass
    const int kp = tid % KP, cc = tid / KP;
#pragma unroll
    for (int p = 0; p < ST / CPP; ++p) {
        const int64_t col = c0 + cc + CPP * p;
        const int64_t row = i0 + 2 * kp;
        const double *src = g.A + col * g.lda + row;
        f64x2 v;
        if (fast) {
            v = *reinterpret_cast<const f64x2 *>(src);
        } else {
            v.x = 0.0; v.y = 0.0;
            if (col < g.cols) {
                if (g.vec_in && row + 1 < iend) v = *reinterpret_cast<const f64x2 *>(src);
                else {
                    if (row < iend) v.x = src[0];
                    if (row + 1 < iend) v.y = src[1];
                }
            }
        }
        reg[p] = v;
    }
}
// The LDS pitch BK+1 is odd so that the 16 columns a half-wave reads for one k land on 16 distinct bank pairs
// (ds_read_b64 / ds_read2_b64); the price is 8-byte instead of 16-byte LDS stores.
template <int TN, int BK>
__device__ __forceinline__ void sk_store_panel(double *panel, const f64x2 (&reg)[Cfg<TN>::NLD * (BK / 16)], int tid) {
    constexpr int KP = BK / 2;
    constexpr int CPP = Cfg<TN>::NT / KP;
    const int kp = tid % KP, cc = tid / KP;
#pragma unroll
    for (int p = 0; p < ST / CPP; ++p) {
        double *d = panel + (cc + CPP * p) * (BK + 1) + 2 * kp;
        d[0] = reg[p].x;
        d[1] = reg[p].y;
    }
}

// lane i of every 16-lane row receives the value of lane (i - N) mod 16 of the same row (DPP row_ror:N), both halves of the double
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// acc = sum over rows [ibeg, iend) of A[i, j0 + .]' * A[i, k0 + .] for this thread's accumulators of the 128x128 tile.
// K-contiguous column panels go global -> registers -> LDS (double buffered, one barrier per BK rows).
// ABL: ablation switch for profiling only (0 = the kernel; 1 = no LDS operand reads; 2 = no global loads / LDS stores).
// FAST: whole 128-column panels, aligned 16-byte loads, a multiple of BK rows — no bounds checks and no conditionals, so the stage
// body is ONE basic block and the compiler interleaves the global loads and LDS traffic with the MFMA stream and sinks half of a
// stage's MFMAs below the barrier (1.29 -> 1.20 ms at n = r = 4096; with the bounds-checked loads the body was ~40 blocks).
template <int TN, int BK, int ABL, bool FAST>
__device__ __forceinline__ void sk_accumulate_impl(const SKArgs &g, int64_t j0, int64_t k0, bool diag, int64_t ibeg, int64_t iend,
                                                   double (&acc)[Cfg<TN>::NACC], double (&lds)[2][2][ST * (BK + 1)], int tid) {
    using C = Cfg<TN>;
    constexpr int GP = BK + 1;
    constexpr int NREG = C::NLD * (BK / 16);
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / C::NWC, wc = wave % C::NWC;
    const int lm = lane & 15, lk = lane >> 4;
    (void)diag;   // diagonal tiles load their one panel twice (6 % of the tiles, L2 hits) rather than branch inside the stage loop
#pragma unroll
    for (int r = 0; r < C::NACC; ++r) acc[r] = 0.0;

    const int nstage = (int)((iend - ibeg + BK - 1) / BK);
    // (A per-workgroup stagger of the contraction order and a padded lda were both tried against channel hot-spotting of the
    // 32 KiB column stride: neither changes this kernel's time once the GPU is warm — profiles/r01c_lda_padding.txt.)
    auto stage_row = [&](int s) { return ibeg + (int64_t)s * BK; };
    f64x2 rj[NREG], rk[NREG];
    __syncthreads();                                   // previous users of the LDS buffers are done
    if (nstage > 0) {
        sk_load_panel<TN, BK, FAST>(g, j0, stage_row(0), iend, rj, tid);
        sk_load_panel<TN, BK, FAST>(g, k0, stage_row(0), iend, rk, tid);
        sk_store_panel<TN, BK>(lds[0][0], rj, tid);
        sk_store_panel<TN, BK>(lds[0][1], rk, tid);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        const double *pj = lds[cur][0] + (wr * 64 + lm) * GP + lk;
        const double *pk = lds[cur][1] + (wc * C::WCOLS) * GP + lk;
        // TN == 4 (128 accumulator VGPRs, 256-VGPR budget): keep the k-step loop rolled so operand reads are not hoisted
        // a whole stage ahead (fully unrolled it spills ~100 VGPRs)
#pragma unroll(TN == 4 ? 1 : BK / 4)
        for (int ks = 0; ks < BK / 4; ++ks) {
            // The global loads of the NEXT stage are issued after the first k-step's MFMAs are queued, not at the top of the
            // stage: right after the barrier both waves of a SIMD would otherwise spend ~450 cycles issuing loads with the
            // matrix pipe idle (in-kernel s_memtime stamps, profiles/r01c_gram_phases.txt).
            // (FAST: unconditional — the last stage re-loads itself — so the stage body stays one basic block)
            if (ks == (BK / 4 > PMT_SK_LOADKS ? PMT_SK_LOADKS : 0) && ABL != 2 && (FAST || s + 1 < nstage)) {
                const int64_t inext = stage_row(FAST ? min(s + 1, nstage - 1) : s + 1);
                sk_load_panel<TN, BK, FAST>(g, j0, inext, iend, rj, tid);
                sk_load_panel<TN, BK, FAST>(g, k0, inext, iend, rk, tid);
            }
            if (PMT_SK_STOREKS >= 0 && ks == PMT_SK_STOREKS && ABL != 2 && (FAST || s + 1 < nstage)) {
                // the next stage's panels go to LDS HERE, a k-step or two before the barrier: their write latency and the wait for the global
                // loads are then covered by this stage's remaining MFMAs instead of sitting between the last MFMA and the barrier
                sk_store_panel<TN, BK>(lds[cur ^ 1][0], rj, tid);
                sk_store_panel<TN, BK>(lds[cur ^ 1][1], rk, tid);
                if (PMT_SK_STOREFENCE) __builtin_amdgcn_sched_barrier(0);
            }
            double a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = (ABL == 1) ? (double)(tid + t) : pj[t * 16 * GP + ks * 4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                double b[4];
                if (ABL == 3) {
                    // one LDS read + three in-register rotations of the 16-lane rows by 4/8/12 lanes (DPP row_ror) instead of four reads
                    b[0] = pk[(tn * 16 + lm) * GP + ks * 4];
                    b[1] = dpp_row_ror<12>(b[0]);
                    b[2] = dpp_row_ror<8>(b[0]);
                    b[3] = dpp_row_ror<4>(b[0]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rc = ((((lm >> 2) + r) & 3) << 2) | (lm & 3);      // column group rotated by r blocks
                        b[r] = (ABL == 1) ? (double)(rc + r) : pk[(tn * 16 + rc) * GP + ks * 4];
                    }
                }
#if PMT_SK_ORDER == 0
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[(tm * TN + tn) * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm], b[r], acc[(tm * TN + tn) * 4 + r], 0, 0, 0);
#else
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
                        acc[(tm * TN + tn) * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm], b[r], acc[(tm * TN + tn) * 4 + r], 0, 0, 0);
#endif
            }
        }
        if (PMT_SK_STOREKS < 0 && ABL != 2 && (FAST || s + 1 < nstage)) {
            sk_store_panel<TN, BK>(lds[cur ^ 1][0], rj, tid);
            sk_store_panel<TN, BK>(lds[cur ^ 1][1], rk, tid);
        }
        __syncthreads();
    }
}

template <int TN, int BK, int ABL>
__device__ __forceinline__ void sk_accumulate(const SKArgs &g, int64_t j0, int64_t k0, bool diag, int64_t ibeg, int64_t iend,
                                              double (&acc)[Cfg<TN>::NACC], double (&lds)[2][2][ST * (BK + 1)], int tid) {
    const bool fast = g.vec_in && (k0 + ST <= g.cols) && ((iend - ibeg) % BK == 0);   // j0 <= k0: panel J is in range too
    if (fast) sk_accumulate_impl<TN, BK, ABL, true>(g, j0, k0, diag, ibeg, iend, acc, lds, tid);
    else sk_accumulate_impl<TN, BK, ABL, false>(g, j0, k0, diag, ibeg, iend, acc, lds, tid);
}

// ---- batched instances (BASELINE config 4): one workgroup per (instance, tile), coefficient-only output ----------------
// out[inst*out_stride + tri(j,k)] = 2 * sum_i A_inst[i,j] * A_inst[i,k]   (the MOI coefficient of the canonical term (j,k); the
// index arrays are identical for every instance and are not rewritten).
struct BatchGramArgs {
    const double *A; int64_t lda, rows, cols, strideA;
    double *out; int64_t out_stride;
    int ntiles, vec_in;
};

__global__ __launch_bounds__(Cfg<2>::NT, 2) void batch_gram_kernel(BatchGramArgs bg) {
    constexpr int TN = 2, BK = 16;
    using C = Cfg<TN>;
    __shared__ double lds[2][2][ST * (BK + 1)];
    const int tid = threadIdx.x;
    const int64_t inst = blockIdx.y;
    SKArgs g;
    g.A = bg.A + inst * bg.strideA; g.lda = bg.lda; g.rows = bg.rows; g.cols = bg.cols; g.vec_in = bg.vec_in;
    int jb, kb;
    sk_tri_unrank((int)blockIdx.x, bg.ntiles, jb, kb);
    double acc[C::NACC];
    sk_accumulate<TN, BK, 0>(g, (int64_t)jb * ST, (int64_t)kb * ST, jb == kb, 0, bg.rows, acc, lds, tid);
    // stage through LDS so that each output row segment is written contiguously (8 bytes per entry)
    double *tile = &lds[0][0][0];
    const int wave = tid >> 6, lane = tid & 63, wr = wave / C::NWC;
    const int64_t n = bg.cols, j0 = (int64_t)jb * ST, k0 = (int64_t)kb * ST;
    double *out = bg.out + inst * bg.out_stride;
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wr == h) {
#pragma unroll
            for (int r = 0; r < C::NACC; ++r) {
                int row, col;
                sk_acc_pos<TN>(tid, r, row, col);
                tile[(row - h * 64) * EPITCH + col] = 2 * acc[r];
            }
        }
        __syncthreads();
        for (int row = wave; row < 64; row += C::NW) {
            const int64_t j = j0 + h * 64 + row;
            if (j >= n) break;
            const int64_t kstart = j > k0 ? j : k0;
            const int64_t kend = (k0 + ST < n) ? k0 + ST : n;
            const int64_t term0 = j * n - (j * (j - 1)) / 2 + (kstart - j);
            for (int64_t k = kstart + lane; k < kend; k += 64) out[term0 + (k - kstart)] = tile[row * EPITCH + (k - k0)];
        }
    }
}

// out[inst*out_stride + j] = 2 * sum_i c_i * A_inst[i,j], c_i = 0.0 (+|-) b_inst[i]; one wave per (instance, column)
__global__ __launch_bounds__(256) void batch_gram_linear_kernel(const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                                                int64_t strideA, const double *__restrict__ b, int64_t strideb, int sign,
                                                                double *__restrict__ out, int64_t out_stride) {
    const int64_t inst = blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const int lane = threadIdx.x & 63;
    const double *a = A + inst * strideA + col * lda;
    const double *bb = b + inst * strideb;
    double acc = 0.0;
    for (int64_t i = lane; i < rows; i += 64) acc += signed_const(bb[i], sign) * a[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) out[inst * out_stride + col] = 2 * acc;
}

// out[inst*out_stride] = ((0 + c_0^2) + c_1^2) + ... left to right (src/functions.jl:574); one thread per instance
__global__ void batch_const_kernel(const double *__restrict__ b, int64_t strideb, int64_t rows, int sign, int64_t B,
                                   double *__restrict__ out, int64_t out_stride) {
    const int64_t inst = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= B) return;
    const double *bb = b + inst * strideb;
    double acc = 0.0;
    for (int64_t i = 0; i < rows; ++i) {
        const double c = signed_const(bb[i], sign);
        const double p = c * c;
        acc = acc + p;
    }
    out[inst * out_stride] = acc;
}

int launch_batch_gram(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                      int64_t B, double *out_q, double *out_lin, double *out_const, int64_t out_stride, hipStream_t s) {
    // (instances of at most 128 columns take batch_small.hip; the caller decides)
    BatchGramArgs bg;
    bg.A = A; bg.lda = lda; bg.rows = rows; bg.cols = cols; bg.strideA = strideA; bg.out = out_q; bg.out_stride = out_stride;
    bg.ntiles = (int)cdiv(cols, ST);
    bg.vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0 && (strideA & 1) == 0) ? 1 : 0;
    const int T = bg.ntiles * (bg.ntiles + 1) / 2;
    for (int64_t i0 = 0; i0 < B; i0 += 65535) {                 // gridDim.y limit
        const unsigned nb = (unsigned)std::min<int64_t>(65535, B - i0);
        BatchGramArgs part = bg;
        part.A = A + i0 * strideA; part.out = out_q + i0 * out_stride;
        PMT_LAUNCH(batch_gram_kernel, dim3((unsigned)T, nb), dim3(Cfg<2>::NT), 0, s, part);
        PMT_LAUNCH(batch_gram_linear_kernel, dim3((unsigned)cdiv(cols, 4), nb), dim3(256), 0, s, A + i0 * strideA, lda, rows, cols, strideA,
                   b + i0 * strideb, strideb, sign, out_lin + i0 * out_stride, out_stride);
    }
    PMT_LAUNCH(batch_const_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, s, b, strideb, rows, sign, B, out_const, out_stride);
    return check_launch("batch_gram");
}

// ABL: ablation switch for profiling only; results are wrong for ABL != 0; selected with PMT_GRAM_SK_ABLATE.
// amdgpu_num_vgpr(108): on gfx90a+ the attribute counts in pairs of the unified file, i.e. a budget of 216 registers.  The budget is part
// of the source (and checked again by tools/kernel_resources.py at build time), not a by-product of which other instantiations share the
// translation unit; the scheduler settles at 183-187 registers under it (see the knobs at the top of this file).
// RANGED: the launch covers the tiles from g.seq_begin on (a stage of a host delivery).  A separate instantiation: the one extra add in the
// tile numbering of the plain kernel moved its schedule by 1 % (1.192 -> 1.204 ms at n = r = 4096).
template <int TN, int BK, int WPS, int ABL, bool RANGED>
__global__ __launch_bounds__(Cfg<TN>::NT, WPS) __attribute__((amdgpu_num_vgpr(108))) void gram_sk_kernel(SKArgs g) {
    using C = Cfg<TN>;
    constexpr int GP = BK + 1;
    __shared__ double lds[2][2][ST * GP];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;

    // Optional (PMT_GRAM_SK_STAGGER): half of the XCDs (workgroup b runs on XCD b % 8) do their stream-K share FIRST and their
    // whole tiles afterwards, so the two halves of the chip reach their tile epilogues ~1/16 of a tile apart and the 200 MB of output
    // are not written in two chip-wide bursts.  Measured: no gain (profiles/r01d_side_stream.txt); off.
    const bool b_first = PMT_GRAM_SK_STAGGER && (bid & 4);
    auto phase_b = [&]() __attribute__((always_inline)) {
        // phase B: the remaining tiles (fewer than G) are split along the contraction: stream-K over their (tile, chunk) units
        const int64_t u0 = sk_unit_begin(g, bid), u1 = sk_unit_begin(g, bid + 1);
        for (int64_t u = u0; u < u1;) {
            const int rtile = (int)(u / g.nchunk);                       // index among the remainder tiles
            const int tile = g.tfull * g.G + rtile;
            const int c0 = (int)(u - (int64_t)rtile * g.nchunk);
            const int c1 = (int)min((int64_t)g.nchunk, (int64_t)c0 + (u1 - u));
            int jb, kb;
            if (RANGED && g.strict) sk_tile_unrank_strict(g, g.seq_begin + g.seq_step * tile, jb, kb);
            else sk_tile_unrank(g, RANGED ? g.seq_begin + g.seq_step * tile : tile, jb, kb);
            const int64_t j0 = (int64_t)jb * ST, k0 = (int64_t)kb * ST;
            const bool diag = (jb == kb);
            const int64_t ibeg = (int64_t)c0 * g.skc, iend = min(g.rows, (int64_t)c1 * g.skc);

            double acc[C::NACC];
            sk_accumulate<TN, BK, ABL>(g, j0, k0, diag, ibeg, iend, acc, lds, tid);

            bool whole = c0 == 0 && c1 == g.nchunk;
            const bool pair = RANGED && g.pair_flags != nullptr && !whole;
            int hsel = -1;                                       // row half the epilogue below writes (-1: both)
            if (pair) {
                // BALANCED PAIR FOLD (stages of a host delivery whose tiles are split exactly in two, launcher).  The two workgroups of a tile —
                // block ids 2t (rows [0, r/2) of the contraction) and 2t + 1, on the chip at the same time, one per CU — each FINISH one row half
                // of the tile: the one with the first half keeps its wave rows 0 (tile rows 0..63) and hands its partial of wave rows 1 to the
                // partner through the workspace (agent-scope write-through stores, then its flag), the other one the other way round; each waits
                // for the partner's flag, adds the 64 KB it was handed to what it kept (first + second: a sum of two, the same bits whoever
                // adds) and writes its 64 rows through the coalescing epilogue.  No fix-up launch, and half the partial traffic and half the
                // write-out on each workgroup's path (round 3's fold had the second workgroup read 128 KB and write the whole tile while the
                // first sat idle: ~10 us per stage).  The wait is bounded; a half whose partner never showed up is written as NaN and the
                // launch's error word is raised (pmt_plan_fetch_synchronize returns PMT_HIP_ERROR, gram.hip).
                const bool first = c0 == 0;
                const int keep = first ? 0 : 1;
                const int ftid = sk_fresh_tid();                                 // (fresh: nothing of this is live across the stage loop)
                const int mywr = __builtin_amdgcn_readfirstlane(ftid >> 6) / C::NWC;          // wave-uniform: the two roles are scalar branches
#if defined(PMT_SK_PAIR_RELAXED) && PMT_SK_PAIR_RELAXED      // round 4's form, kept for the A/B (profiles/r05_pair_fold.txt)
                if (mywr != keep) {
                    double *w = g.ws + (int64_t)(2 * bid) * SLOT + ftid;
    #pragma unroll
                    for (int r = 0; r < C::NACC; ++r) __hip_atomic_store(&w[r * C::NT], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_store(&g.pair_flags[(first ? 0 : 512) + rtile], g.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned *other = &g.pair_flags[(first ? 512 : 0) + rtile];
                    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                    double late = 0.0;
                    while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > g.pair_timeout) { late = 1.0; break; }   // 2 s (100 MHz ticks)
                    }
                    if (late != 0.0 && g.error) __hip_atomic_store(g.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    lds[0][0][0] = late;                         // (the panels are idle between the stage loop and the epilogue)
                }
                __syncthreads();
                const bool late = lds[0][0][0] != 0.0;
                if (mywr == keep) {
                    const double *w = g.ws + (int64_t)(2 * (bid ^ 1)) * SLOT + ftid;
    #pragma unroll
                    for (int r0 = 0; r0 < C::NACC; r0 += 4) {      // four loads in flight at a time (register budget)
                        double other[4];
    #pragma unroll
                        for (int r = 0; r < 4; ++r) other[r] = __hip_atomic_load(&w[(r0 + r) * C::NT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    #pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r0 + r] = late ? __builtin_nan("") : other[r] + acc[r0 + r];
                        asm volatile("" ::: "memory");
                    }
                }
#else
                // The hand-off, measured in three forms (profiles/r05_pair_fold.txt; host_csc per solve, alternating runs on one box):
                //   round 4   relaxed agent-scope atomic stores / loads of partial and flag + s_waitcnt               1.713-1.728 ms
                //   FORMAL    plain stores, barrier, agent-scope RELEASE flag store; relaxed spin, agent-scope ACQUIRE fence, barrier, plain loads
                //             (-DPMT_SK_PAIR_FORMAL=1; the textbook form of MI355X_MICROARCH.md)                         1.762-1.769 ms (+2.6 %)
                //   shipped   producer as in round 4, consumer as in FORMAL                                              1.726-1.733 ms (+0.5 %)
                // The release is what costs: `buffer_wbl2 sc1` writes back the XCD's whole L2, which at that moment holds the dirty lines of
                // the 31 other CUs' tile epilogues; 1.5 % per solve for an ordering the write-through (sc1) stores + s_waitcnt vmcnt(0) of the
                // producer already give on this hardware (the guide's second recipe: sc1 stores, then the flag).  The consumer side IS the
                // model's: one acquire fence by the thread that saw the flag (it invalidates the CU's L1 and the non-local L2 lines), the
                // workgroup barrier, then plain loads by everybody — no stale line can be read.  The partial and the flag are atomic objects
                // on the producer side, so there is no data race in the language model either; what the shipped form does not have is the
                // model's release edge, and the soak (tools/soak_host_delivery.py, 6000 solves x 4 configurations) plus the fault-injection
                // test cover what it rests on.
                if (mywr != keep) {
                    double *w = g.ws + (int64_t)(2 * bid) * SLOT + ftid;
    #pragma unroll
#if !(defined(PMT_SK_PAIR_FORMAL) && PMT_SK_PAIR_FORMAL)
                    for (int r = 0; r < C::NACC; ++r) __hip_atomic_store(&w[r * C::NT], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                    for (int r = 0; r < C::NACC; ++r) w[r * C::NT] = acc[r];
#endif
                }
#if !(defined(PMT_SK_PAIR_FORMAL) && PMT_SK_PAIR_FORMAL)
                __builtin_amdgcn_s_waitcnt(0);
#endif
                __syncthreads();
                if (tid == 0) {
#if !(defined(PMT_SK_PAIR_FORMAL) && PMT_SK_PAIR_FORMAL)
                    __hip_atomic_store(&g.pair_flags[(first ? 0 : 512) + rtile], g.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                    __hip_atomic_store(&g.pair_flags[(first ? 0 : 512) + rtile], g.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
                    const unsigned *other = &g.pair_flags[(first ? 512 : 0) + rtile];
                    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                    double late = 0.0;
                    while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > g.pair_timeout) { late = 1.0; break; }   // 2 s (100 MHz ticks)
                    }
#if !(defined(PMT_SK_PAIR_NOACQ) && PMT_SK_PAIR_NOACQ)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                    if (late != 0.0 && g.error) __hip_atomic_store(g.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    lds[0][0][0] = late;                         // (the panels are idle between the stage loop and the epilogue)
                }
                __syncthreads();
                const bool late = lds[0][0][0] != 0.0;
                if (mywr == keep) {
                    const double *w = g.ws + (int64_t)(2 * (bid ^ 1)) * SLOT + ftid;
    #pragma unroll
                    for (int r0 = 0; r0 < C::NACC; r0 += 4) {      // four loads in flight at a time (register budget)
                        double other[4];
    #pragma unroll
#if defined(PMT_SK_PAIR_NOACQ) && PMT_SK_PAIR_NOACQ
                        for (int r = 0; r < 4; ++r) other[r] = __hip_atomic_load(&w[(r0 + r) * C::NT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                        for (int r = 0; r < 4; ++r) other[r] = w[(r0 + r) * C::NT];
#endif
    #pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r0 + r] = late ? __builtin_nan("") : other[r] + acc[r0 + r];
                        asm volatile("" ::: "memory");
                    }
                }
#endif
                whole = true;
                hsel = keep;
            }
            if (whole) {
                static_assert(2 * 2 * ST * GP >= EPI_DOUBLES, "panel LDS must hold the epilogue staging tile");
                sk_epilogue<TN>(g, jb, kb, acc, &lds[0][0][0], tid, RANGED ? hsel : -1);
            } else {
                // partial tile -> workspace slot, stored [accumulator index][thread] (coalesced); the fix-up kernel knows the map
                const int slot = 2 * bid + (u == u0 ? 0 : 1);
                double *w = g.ws + (int64_t)slot * SLOT + (RANGED ? sk_fresh_tid() : tid);
    #pragma unroll
                for (int r = 0; r < C::NACC; ++r) w[r * C::NT] = acc[r];
            }
            u += (c1 - c0);
        }
    };
    if (b_first) phase_b();

    // phase A: tfull whole tiles per workgroup (contiguous, so consecutive tiles share their row panel in L2), written directly
    for (int t = 0; t < g.tfull; ++t) {
        int jb, kb;
        if (RANGED && g.strict) sk_tile_unrank_strict(g, g.seq_begin + g.seq_step * sk_phase_a_index(g, bid, t), jb, kb);
        else sk_tile_unrank(g, RANGED ? g.seq_begin + g.seq_step * sk_phase_a_index(g, bid, t) : sk_phase_a_index(g, bid, t), jb, kb);
        double acc[C::NACC];
        sk_accumulate<TN, BK, ABL>(g, (int64_t)jb * ST, (int64_t)kb * ST, jb == kb, 0, g.rows, acc, lds, tid);
        sk_epilogue<TN>(g, jb, kb, acc, &lds[0][0][0], tid);
    }

    if (!b_first) phase_b();
}

#ifndef PMT_SK_APB1_BELOW
#define PMT_SK_APB1_BELOW 64       // fix-up: one accumulator per workgroup (NACC workgroups per tile) while tiles x NACC / 4 stays below this
#endif
// one workgroup per tile: if the tile was split, add its partials in ascending workgroup order and write the terms
// APB = accumulators per thread handled by one workgroup: 4 normally; 1 when only a few tiles are split (tall matrices: one tile summed
// over up to 256 partials) so that the sum is spread over NACC instead of NACC/4 workgroups per tile
template <int TN, int APB>
__global__ __launch_bounds__(Cfg<TN>::NT) void gram_sk_fixup_kernel(SKArgs g) {
    using C = Cfg<TN>;
    const int rtile = blockIdx.x;                                              // index among the remainder (split) tiles
    const int tile = g.tfull * g.G + rtile;
    const int tid = threadIdx.x;
    const int64_t ub = (int64_t)rtile * g.nchunk, ue = ub + g.nchunk - 1;      // first / last remainder unit of this tile
    auto owner = [&](int64_t u) {
        int b = (int)((u * g.G) / g.U);
        if (b >= g.G) b = g.G - 1;
        while (b + 1 < g.G && sk_unit_begin(g, b + 1) <= u) ++b;
        while (b > 0 && sk_unit_begin(g, b) > u) --b;
        return b;
    };
    const int blo = owner(ub), bhi = owner(ue);
    if (blo == bhi) return;                                                   // one workgroup did the whole tile (and counted it)
    // blockIdx.y selects APB of the NACC accumulators of every thread, so a tile split many ways is summed by NACC/APB workgroups
    const int r0 = (int)blockIdx.y * APB;
    double acc[APB];
#pragma unroll
    for (int r = 0; r < APB; ++r) acc[r] = 0.0;
    auto slot_ptr = [&](int b) {
        const int64_t bu0 = sk_unit_begin(g, b);
        const int first_rtile = (int)(bu0 / g.nchunk);
        const int slot = 2 * b + (first_rtile == rtile ? 0 : 1);
        return g.ws + (int64_t)slot * SLOT + (int64_t)r0 * C::NT + tid;
    };
    // partials are ADDED in ascending workgroup order (deterministic) but LOADED LB workgroups at a time, so the kernel is not a
    // chain of dependent L2 round trips
    constexpr int LB = APB == 1 ? 16 : 4;
    int b = blo;
    for (; b + LB - 1 <= bhi; b += LB) {
        double v[LB][APB];
#pragma unroll
        for (int q = 0; q < LB; ++q) {
            const double *w = slot_ptr(b + q);
#pragma unroll
            for (int r = 0; r < APB; ++r) v[q][r] = w[r * C::NT];
        }
#pragma unroll
        for (int q = 0; q < LB; ++q)
#pragma unroll
            for (int r = 0; r < APB; ++r) acc[r] = acc[r] + v[q][r];
    }
    for (; b <= bhi; ++b) {
        const double *w = slot_ptr(b);
#pragma unroll
        for (int r = 0; r < APB; ++r) acc[r] = acc[r] + w[r * C::NT];
    }
    int jb, kb;
    if (g.strict) sk_tile_unrank_strict(g, g.seq_begin + g.seq_step * tile, jb, kb);
    else sk_tile_unrank(g, g.seq_begin + g.seq_step * tile, jb, kb);
#pragma unroll
    for (int r = 0; r < APB; ++r) {
        int row, col;
        sk_acc_pos<TN>(tid, r0 + r, row, col);
        sk_store_term(g, jb, kb, row, col, acc[r]);
    }
}

// The same sum for tiles split MANY ways (one off-diagonal tile of a 262144 x 256 matrix: 256 partials of 128 KB): the kernel above adds
// them as a chain of 16 rounds of 16 loads per thread with 32 workgroups per tile — 64 us for 33 MB.  Here a workgroup takes 64 elements of
// one accumulator row and EIGHT interleaved slices of the partials (slice t: workgroups blo + t, blo + t + 8, .. in order; 512-byte runs),
// the slices are added in order through LDS: 256 workgroups per tile, two rounds of loads each.  A fixed order, like the other one's.
template <int TN>
__global__ __launch_bounds__(512) void gram_sk_fixup_sliced_kernel(SKArgs g) {
    using C = Cfg<TN>;
    constexpr int NSL = 8;
    __shared__ double part[NSL][64];
    const int rtile = blockIdx.x;
    const int tile = g.tfull * g.G + rtile;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = (int)blockIdx.z * 64 + el;                                  // the thread of the contraction whose accumulator this is
    const int r0 = (int)blockIdx.y;
    const int64_t ub = (int64_t)rtile * g.nchunk, ue = ub + g.nchunk - 1;
    auto owner = [&](int64_t u) {
        int b = (int)((u * g.G) / g.U);
        if (b >= g.G) b = g.G - 1;
        while (b + 1 < g.G && sk_unit_begin(g, b + 1) <= u) ++b;
        while (b > 0 && sk_unit_begin(g, b) > u) --b;
        return b;
    };
    const int blo = owner(ub), bhi = owner(ue);
    if (blo == bhi) return;
    auto slot_ptr = [&](int b) {
        const int64_t bu0 = sk_unit_begin(g, b);
        const int first_rtile = (int)(bu0 / g.nchunk);
        const int slot = 2 * b + (first_rtile == rtile ? 0 : 1);
        return g.ws + (int64_t)slot * SLOT + (int64_t)r0 * C::NT + e;
    };
    double acc = 0.0;
    int b = blo + sl;
    for (; b + 3 * NSL <= bhi; b += 4 * NSL) {
        const double v0 = *slot_ptr(b), v1 = *slot_ptr(b + NSL), v2 = *slot_ptr(b + 2 * NSL), v3 = *slot_ptr(b + 3 * NSL);
        acc = acc + v0; acc = acc + v1; acc = acc + v2; acc = acc + v3;
    }
    for (; b <= bhi; b += NSL) acc = acc + *slot_ptr(b);
    part[sl][el] = acc;
    __syncthreads();
    if (sl != 0) return;
    double v = part[0][el];
#pragma unroll
    for (int t = 1; t < NSL; ++t) v = v + part[t][el];
    int jb, kb;
    if (g.strict) sk_tile_unrank_strict(g, g.seq_begin + g.seq_step * tile, jb, kb);
    else sk_tile_unrank(g, g.seq_begin + g.seq_step * tile, jb, kb);
    int row, col;
    sk_acc_pos<TN>(e, r0, row, col);
    sk_store_term(g, jb, kb, row, col, v);
}

size_t gram_sk_workspace_bytes(int64_t rows, int64_t cols) {
    (void)rows; (void)cols;
    return (size_t)MAXG * 2 * SLOT * sizeof(double);
}

// Tuning builds only (-DPMT_TUNING, tools/): kernel variant / ablation / grid size from the environment.  The shipped library reads
// no environment variable: its results cannot be changed from outside.
#ifdef PMT_TUNING
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

// seq_count < 0: all tiles.  Otherwise the launch covers the tiles [seq_begin, seq_begin + seq_count) of the tile sequence (gram_common.h:
// sk_tile_unrank) — a host delivery runs the contraction band range by band range (gram.hip), every range as its own stream-K launch over
// the whole chip, so that the copy engine can ship a range while the next one is computed.  pair_flags (ranged launches; R words, holding
// anything but `epoch`): tiles split exactly in two are summed inside the launch (see the kernel) instead of by the fix-up pass.
int launch_gram_sk(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                   pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, int order_w, int64_t seq_begin, int64_t seq_count,
                   unsigned *pair_flags, unsigned epoch, int *error_word, hipStream_t s, int strict) {
    SKArgs g;
    g.strict = (strict && seq_count >= 0) ? 1 : 0;
    g.A = A; g.lda = lda; g.rows = rows; g.cols = cols; g.xvar = xvar; g.varmap = varmap; g.moi = moi; g.out_quad = out_quad;
    g.out_csc = out_csc; g.alpha = alpha;
    g.ntiles = (int)cdiv(cols, ST);
    const int64_t T = seq_count >= 0 ? seq_count : (int64_t)g.ntiles * (g.ntiles + 1) / 2;      // tiles of this launch
    // Work units of SKC rows leave a small problem on a few CUs (512 x 512: 20 units, 42 us at one CU's rate each): where the units would
    // not fill the chip they shrink, down to 64 rows (four stages).  Host deliveries plan their stages on SKC-row units (gram.hip) and
    // keep them.
    g.skc = SKC;
    if (seq_count < 0 || g.strict)
        while (g.skc > 64 && T * cdiv(rows, g.skc) < 256) g.skc >>= 1;
    g.nchunk = (int)std::max<int64_t>(1, cdiv(rows, g.skc));
    g.seq_begin = (int)seq_begin; g.seq_step = 1;
    if (order_w < 0) {                                // walked from the end: position p of the walk is tile T_all - 1 - p of the sequence
        order_w = -order_w;
        g.seq_begin = (int)((int64_t)g.ntiles * (g.ntiles + 1) / 2 - 1 - seq_begin);
        g.seq_step = -1;
    }
    // variant: 0 = wg256 (two 4-wave workgroups per CU, 64x64 wave tiles), 1 = wg512 (one 8-wave workgroup per CU, 64x32), BK 16
#ifdef PMT_TUNING
#ifdef PMT_TUNING_ABLATE
    static const int variant = env_int("PMT_GRAM_SK_VARIANT", 1);   // measured equal within noise (profiles/r01b_gram_variants.txt)
    static const int abl = env_int("PMT_GRAM_SK_ABLATE", 0);
#else
    constexpr int variant = 1;
#endif
    static const int gdef = env_int("PMT_GRAM_SK_BLOCKS", 0);
    static const int order_env = env_int("PMT_GRAM_SK_ORDER_W", -1);
    if (order_env >= 0) order_w = order_env;
#else
    constexpr int variant = 1, gdef = 0;
#endif
    const int gwant = gdef > 0 ? gdef : (variant == 0 ? 512 : 256);
    g.G = (int)std::min<int64_t>(T * g.nchunk, std::min(gwant, MAXG));
    // strict launches of tall matrices (few tiles, each split over many workgroups): a grid that is a MULTIPLE of the tile count gives every
    // tile the same row ranges, so that the workgroups of different tiles that share a column panel read the same rows of it at the same
    // time (Infinity Cache) — 65536 x 1024: 28 tiles on 252 instead of 256 workgroups, 1.178 -> 1.163 ms
    if (g.strict && gdef == 0 && T >= 2 && T <= 32 && T * g.nchunk >= 4 * (int64_t)gwant) g.G = (int)(T * (gwant / T));
    g.tfull = (int)(T / g.G);                         // at n = r = 4096: 528 tiles = 2 per workgroup + 16 split 16 ways
    const int64_t R = T - (int64_t)g.tfull * g.G;     // remainder tiles, < G
    g.U = R * g.nchunk;
    g.vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
    g.ws = reinterpret_cast<double *>(workspace);
    g.order_w = order_w;
    if (T <= 0) return PMT_OK;
    // pair fold: every tile of this ranged launch is split in exactly two halves held by neighbouring workgroups
    g.pair_flags = nullptr; g.epoch = 0; g.flag_value = 0; g.error = error_word; g.pair_timeout = 200000000LL;
    // (the flags live in the tail of the workspace, behind the partial-tile slots of a grid of at most MAXG / 2 workgroups: gram.hip)
    const bool fold = seq_count >= 0 && pair_flags && g.tfull == 0 && (g.nchunk & 1) == 0 && 2 * R == g.G && g.U == (int64_t)g.G * (g.nchunk / 2) && g.G <= MAXG / 2;
    if (fold) {
        g.pair_flags = pair_flags; g.epoch = g.flag_value = epoch;
        // test hook (pmt_set_fault_injection(1)): the first halves announce a value nobody waits for and the wait is cut to 20 ms — the
        // second halves run into their bound, write NaN tiles and raise the error word
        if (dma::fault_injection() & 1) { g.flag_value = epoch + 0x40000000u; g.pair_timeout = 2000000LL; }
    }
    if (g.nchunk > 1 && !workspace) return fail(PMT_INVALID_ARGUMENT, "quad_gram: workspace required");
    const dim3 grid((unsigned)g.G);
#ifdef PMT_TUNING_ABLATE
    // (ablation instantiations: profiling only, results are wrong for abl != 0; they are a separate switch because their mere presence in
    // the translation unit moves the register allocation of the shipped kernel)
#define SK_LAUNCH(TN, BK, WPS)                                                                                              \
    do {                                                                                                                    \
        if (abl == 1) PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<TN, BK, WPS, 1, false>), grid, dim3(Cfg<TN>::NT), 0, s, g);  \
        else if (abl == 2) PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<TN, BK, WPS, 2, false>), grid, dim3(Cfg<TN>::NT), 0, s, g); \
        else if (abl == 3) PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<TN, BK, WPS, 3, false>), grid, dim3(Cfg<TN>::NT), 0, s, g); \
        else PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<TN, BK, WPS, 0, false>), grid, dim3(Cfg<TN>::NT), 0, s, g);       \
    } while (0)
    if (variant == 0) SK_LAUNCH(4, 16, 2);
    else SK_LAUNCH(2, 16, PMT_SK_WPS);
#undef SK_LAUNCH
#else
    // Two instantiations, plain and ranged.  (The 32-row-stage instantiation used for tall matrices in rounds 1-2 spilled 49 VGPRs — +1 % at r = 16384 when it
    // was introduced — and is gone: all shapes take 16-row stages.  So is the delivery instantiation that counted finished tiles per band
    // group inside the kernel: a delivery is now a sequence of plain launches, gram.hip.)
    if (seq_count >= 0) PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<2, PMT_SK_BK, PMT_SK_WPS, 0, true>), grid, dim3(Cfg<2>::NT), 0, s, g);
    else PMT_LAUNCH_NAMED("gram_sk_kernel", (gram_sk_kernel<2, PMT_SK_BK, PMT_SK_WPS, 0, false>), grid, dim3(Cfg<2>::NT), 0, s, g);
#endif
    int rc = check_launch("gram_sk_kernel");
    if (rc) return rc;
    if (g.nchunk > 1 && R > 0 && !fold) {
        // The split tiles are summed by a second launch.  (Round 3 measured the alternative — the workgroup that arrives last at a split tile
        // adds its partials inside the contraction: +55 us at n = r = 4096, one CU pulling 2 MB of partials, against 15 us of fix-up kernel
        // plus ~25 us of in-stream gaps; profiles/r03_gram_fold_experiment.txt.)
#ifdef PMT_TUNING
        static const int apb = env_int("PMT_GRAM_SK_FIXUP_APB", 0);
#else
        constexpr int apb = 0;
#endif
        // (tiles split more than 32 ways each — few tiles, many rows: the sliced form)
        if (apb == 0 && (int64_t)g.G >= 32 * R) PMT_LAUNCH_NAMED("gram_sk_fixup_sliced_kernel", (gram_sk_fixup_sliced_kernel<2>), dim3((unsigned)R, Cfg<2>::NACC, Cfg<2>::NT / 64), dim3(512), 0, s, g);
        else if (apb == 1 || (apb == 0 && R * (Cfg<2>::NACC / 4) < PMT_SK_APB1_BELOW)) PMT_LAUNCH_NAMED("gram_sk_fixup_kernel", (gram_sk_fixup_kernel<2, 1>), dim3((unsigned)R, Cfg<2>::NACC), dim3(Cfg<2>::NT), 0, s, g);
        else PMT_LAUNCH_NAMED("gram_sk_fixup_kernel", (gram_sk_fixup_kernel<2, 4>), dim3((unsigned)R, Cfg<2>::NACC / 4), dim3(Cfg<2>::NT), 0, s, g);
        rc = check_launch("gram_sk_fixup_kernel");
    }
    return rc;
}

}  // namespace pmt
