// Canonical least-squares objective for TALL matrices (rows >> columns, at most 128 columns): the usual least-squares shape
// (README.md:34-38 with many more residual rows than variables).  ONE pass over A produces everything the node needs:
//   upper triangle of A'A (f64 MFMA), q = A'c and c'c (c = 0.0 (+|-) b), as per-workgroup partials over interleaved 32-row stages,
// then one fix-up launch adds the partials in a fixed order and writes the MOI / native terms.
//
// Reference semantics replaced: _vecdot!/muladd! literal expansion (src/functions.jl:702-709,548-576) + canonicalize!
// (src/functions.jl:381-386, src/util.jl:9-26) + update!(::MOI.ScalarQuadraticFunction) (src/moi_interop.jl:45-62): SURVEY Appendix A.3.
//
// Why a second form beside the stream-K kernel (gram_sk.hip): with one 128-column tile that kernel computes the full 128 x 128 square
// (2 r n^2 FLOP for r n (n + 1) needed), its q and c'c are two more kernels that read A and b again, and the constant up to 8192 rows is a
// one-wave chain.  2^20 x 128: 1.06 ms for 1.07 GB of A.  Here:
//   * the triangle only: the 128 x 128 tile is 8 x 8 blocks of 16 x 16; the 36 blocks on and above the diagonal are dealt out NINE per
//     wave to a 4-wave workgroup (one wave per SIMD; 36 accumulator registers per lane), each wave's set chosen so that it needs few
//     distinct operand rows / columns from LDS.  Executed / needed flops = 36 * 256 / 8256 = 1.12 (the square: 1.98).
//   * small workgroups (4 waves, 70 KB LDS): two per CU, their barriers do not line up, the matrix pipes stay fed.
//   * q and c'c ride along on the VALU: every thread adds c_i * A[i, j] for the 16-byte pieces it has just loaded, on their way to LDS
//     (free: the same kernel without that arithmetic takes the same time).
//   * A is read once.  At n = 128 the node needs 16 FLOP per byte of A, the chip delivers ~10: both pipes matter, and with both busy the
//     chip runs at its power limit (2.08 GHz, matrix pipe 72 % busy: profiles/r05_gram_tall_pmc_sq.txt).
// 2^20 x 128: 0.39 ms kernel + 0.01 ms fix-up (was 1.06 ms); the measurement record, step by step, is profiles/r05_gram_tall.txt.
// Summation order (fixed; pmt_quad_gram_constant_order reports it, tests/gpu_util.py restates the constant's bit for bit): workgroup g
// takes the stages g, g + G, .. in order (MFMA k order for the triangle; per-thread row pairs, then an 8-lane tree, for q and c'c), the
// workgroups are added in 16 interleaved slices, then the slices in order.
#include "gram_common.h"

// Knobs of the measured sweep (profiles/r05_gram_tall.txt); the defaults are what ships.  UNROLL: k-steps of a stage unrolled together
// (1: 197 VGPRs at 32-row stages); MR: rows per stage; WPS: waves per SIMD the register budget aims at; MAXG: workgroups at most;
// K2: one 16-byte operand read for two k-steps; ABL: ablations (wrong results).
#ifndef PMT_TALL_UNROLL
#define PMT_TALL_UNROLL 1
#endif
#ifndef PMT_TALL_WPS
#define PMT_TALL_WPS 2
#endif
#ifndef PMT_TALL_MR
#define PMT_TALL_MR 32
#endif
#ifndef PMT_TALL_ABL
#define PMT_TALL_ABL 0
#endif
#ifndef PMT_TALL_MAXG
#define PMT_TALL_MAXG 512
#endif
#ifndef PMT_TALL_K2
#define PMT_TALL_K2 1
#endif
#ifndef PMT_TALL_DIAG3
#define PMT_TALL_DIAG3 1         // a diagonal block's third rotation is not computed (tall_diag_rule)
#endif

namespace pmt {

constexpr int TBK = PMT_TALL_MR;            // rows per stage (one barrier per stage): every column piece a workgroup asks for is 8 * TBK contiguous bytes
constexpr int TGP = TBK + 2;                // LDS pitch of a column: 2 mod 32 — the 16 columns x 2 k of a half-wave operand read fall on 32 distinct
                                            // bank pairs, and even, so that the row pairs go to LDS as 16-byte stores
constexpr int TSUB = TBK / 16;              // 16-row pieces of a column per stage
constexpr int TLK = PMT_TALL_K2 ? 2 : 1;    // rows between the contraction slots of a k-step (see tall_stage)
constexpr int TCOLS = 128;                  // columns of the one tile
constexpr int TBLK = 9;                     // 16 x 16 blocks per wave at most (NBC = 8)
constexpr int TACC = TBLK * 4;              // accumulators per lane at most (4 rotations per block)
constexpr int TPART = 4 * TACC * 64;        // doubles of triangle partial per workgroup at most (= 36 blocks x 256)
constexpr int TSTRIDE = TPART + TCOLS + 8;  // + q partial + c'c partial (padded to 64 bytes): what the workspace is sized for
constexpr int TSLICES = 16;                 // interleaved slices of the fix-up sum
constexpr int TALL_MAX_G = PMT_TALL_MAXG;             // workgroups (row chunks) at most
constexpr int TALL_MIN_CHUNK = 64;          // rows per chunk at least (one stage for one-tile shapes of up to 2048 rows: tall_chunk)

// NBC = block columns of the one tile the kernel is built for: 8 (113 .. 128 columns, and every diagonal tile of a wide matrix), 7 (97 .. 112),
// 6 (81 .. 96), 5 (65 .. 80) — round 6: a 100-column matrix used to pay for all 36 blocks of the 128-column triangle (2^20 x 80: 342 us,
// the same as 2^20 x 128).  The upper triangle of the NBC x NBC block grid is dealt to the four waves so that their MFMA counts are level
// (3 per diagonal block, 4 per other: 33-36 / 26-27 / 19-20 / 11-15 per pair of k-steps) and each wave needs few distinct operand rows /
// columns from LDS.  NBC = 8 is round 5's dealing by rectangles:
//   wave 0: rows {0,1,2} x cols {5,6,7}        wave 1: rows {0,1,2} x cols {3,4}, and the triangle on {3,4}
//   wave 2: the triangle on {0,1,2}, and row 3 x cols {5,6,7}        wave 3: rows 4..7 x cols {5,6,7} on and above the diagonal
// 5, 6, 7 come from a small annealing search (tools/gen_tall_deal.py prints these tables).  (Local constexpr tables inside constexpr
// functions: usable from device code without a device-side definition; every use below has compile-time arguments after unrolling.)
__host__ __device__ constexpr int tw_nblk(int nbc, int w) { constexpr int t[4][4] = {{3, 4, 4, 4}, {5, 5, 5, 6}, {7, 7, 7, 7}, {9, 9, 9, 9}}; return t[nbc - 5][w]; }
__host__ __device__ constexpr int tw_nr(int nbc, int w) { constexpr int t[4][4] = {{3, 3, 4, 2}, {5, 5, 4, 4}, {3, 4, 4, 5}, {3, 5, 4, 4}}; return t[nbc - 5][w]; }
__host__ __device__ constexpr int tw_nc(int nbc, int w) { constexpr int t[4][4] = {{1, 2, 1, 2}, {1, 1, 2, 4}, {3, 2, 2, 2}, {3, 2, 6, 3}}; return t[nbc - 5][w]; }
__host__ __device__ constexpr int tw_row(int nbc, int w, int i) {          // i-th distinct block row of wave w
    constexpr int t[4][4][6] = {{{2, 3, 4, 0, 0, 0}, {0, 1, 2, 0, 0, 0}, {0, 1, 2, 3, 0, 0}, {0, 1, 0, 0, 0, 0}}, {{0, 1, 2, 3, 4, 0}, {0, 1, 2, 3, 4, 0}, {0, 1, 2, 3, 0, 0}, {0, 1, 2, 5, 0, 0}}, {{0, 1, 2, 0, 0, 0}, {3, 4, 5, 6, 0, 0}, {0, 1, 2, 3, 0, 0}, {0, 1, 2, 3, 4, 0}}, {{0, 1, 2, 0, 0, 0}, {0, 1, 2, 3, 4, 0}, {0, 1, 2, 3, 0, 0}, {4, 5, 6, 7, 0, 0}}};
    return t[nbc - 5][w][i];
}
__host__ __device__ constexpr int tw_col(int nbc, int w, int i) {          // i-th distinct block column of wave w
    constexpr int t[4][4][6] = {{{4, 0, 0, 0, 0, 0}, {0, 2, 0, 0, 0, 0}, {3, 0, 0, 0, 0, 0}, {1, 4, 0, 0, 0, 0}}, {{5, 0, 0, 0, 0, 0}, {4, 0, 0, 0, 0, 0}, {1, 3, 0, 0, 0, 0}, {0, 1, 2, 5, 0, 0}}, {{0, 2, 6, 0, 0, 0}, {5, 6, 0, 0, 0, 0}, {3, 5, 0, 0, 0, 0}, {1, 4, 0, 0, 0, 0}}, {{5, 6, 7, 0, 0, 0}, {3, 4, 0, 0, 0, 0}, {0, 1, 2, 5, 6, 7}, {5, 6, 7, 0, 0, 0}}};
    return t[nbc - 5][w][i];
}
__host__ __device__ constexpr int tw_blk(int nbc, int w, int k, int which) {   // block k of wave w: (index into its rows, index into its columns; 6: a padding slot)
    constexpr int t[4][4][TBLK][2] = {{{{0, 0}, {1, 0}, {2, 0}, {0, 6}, {0, 6}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {0, 1}, {1, 1}, {2, 1}, {0, 6}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {0, 6}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {0, 6}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}}, {{{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 1}, {0, 6}, {0, 6}, {0, 6}, {0, 6}}, {{0, 0}, {1, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 3}, {0, 6}, {0, 6}, {0, 6}}}, {{{0, 0}, {0, 1}, {1, 1}, {2, 1}, {0, 2}, {1, 2}, {2, 2}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {2, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 1}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {0, 1}, {1, 1}, {2, 1}, {0, 6}, {0, 6}}, {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 1}, {4, 1}, {0, 6}, {0, 6}}}, {{{0, 0}, {1, 0}, {2, 0}, {0, 1}, {1, 1}, {2, 1}, {0, 2}, {1, 2}, {2, 2}}, {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 1}, {4, 1}}, {{0, 0}, {0, 1}, {1, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 3}, {3, 4}, {3, 5}}, {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}}}};
    return t[nbc - 5][w][k][which];
}
__host__ __device__ constexpr int tall_tblk(int nbc) { return nbc == 8 ? 9 : nbc == 7 ? 7 : nbc == 6 ? 6 : 4; }      // accumulator slots (blocks) per wave
__host__ __device__ constexpr int tall_tacc(int nbc) { return 4 * tall_tblk(nbc); }
__host__ __device__ constexpr int tall_part(int nbc) { return 4 * tall_tacc(nbc) * 64; }
__host__ __device__ constexpr int tall_stride(int nbc) { return tall_part(nbc) + TCOLS + 8; }

// (tm, tn) of block k of wave w, for the fix-up kernel (the same tables, as data; -1: a padding slot)
__device__ __constant__ signed char TALL_BLOCKS[4][4][TBLK][2] = {{{{2, 4}, {3, 4}, {4, 4}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 0}, {0, 2}, {1, 2}, {2, 2}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 3}, {1, 3}, {2, 3}, {3, 3}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 1}, {1, 1}, {0, 4}, {1, 4}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}}, {{{0, 5}, {1, 5}, {2, 5}, {3, 5}, {4, 5}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 4}, {1, 4}, {2, 4}, {3, 4}, {4, 4}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 1}, {0, 3}, {1, 3}, {2, 3}, {3, 3}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}, {{0, 0}, {1, 1}, {0, 2}, {1, 2}, {2, 2}, {5, 5}, {-1, -1}, {-1, -1}, {-1, -1}}}, {{{0, 0}, {0, 2}, {1, 2}, {2, 2}, {0, 6}, {1, 6}, {2, 6}, {-1, -1}, {-1, -1}}, {{3, 5}, {4, 5}, {5, 5}, {3, 6}, {4, 6}, {5, 6}, {6, 6}, {-1, -1}, {-1, -1}}, {{0, 3}, {1, 3}, {2, 3}, {3, 3}, {0, 5}, {1, 5}, {2, 5}, {-1, -1}, {-1, -1}}, {{0, 1}, {1, 1}, {0, 4}, {1, 4}, {2, 4}, {3, 4}, {4, 4}, {-1, -1}, {-1, -1}}}, {{{0, 5}, {1, 5}, {2, 5}, {0, 6}, {1, 6}, {2, 6}, {0, 7}, {1, 7}, {2, 7}}, {{0, 3}, {1, 3}, {2, 3}, {3, 3}, {0, 4}, {1, 4}, {2, 4}, {3, 4}, {4, 4}}, {{0, 0}, {0, 1}, {1, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 5}, {3, 6}, {3, 7}}, {{4, 5}, {5, 5}, {4, 6}, {5, 6}, {6, 6}, {4, 7}, {5, 7}, {6, 7}, {7, 7}}}};

struct TallArgs {
    const double *A; int64_t lda, rows, cols;
    const double *b; int sign;            // c_i = 0.0 (+|-) b[i]; b == null or sign == 0: c = 0
    int64_t chunk;                        // stages per workgroup when the stages are dealt out in contiguous chunks (interleave == 0)
    int64_t nstages;                      // stages of TBK rows in all
    int interleave;                       // 1: workgroup g takes the stages g, g + G, g + 2G, ..  At any moment the chip then reads ONE band of
                                          // G * TBK consecutive rows of every column: neighbouring workgroups ask for neighbouring pieces of the
                                          // same DRAM pages at about the same time.  With contiguous chunks every workgroup streams its own
                                          // 128 x (8 * TBK bytes) pieces from 128 x G different pages: 2 TB/s (profiles/r05_gram_tall.txt)
    double *ws;                           // gridDim.x x TSTRIDE doubles
    int vec_in;                           // A 16-byte aligned and lda even
};

// PMT_TALL_DPP (off; an experiment kept as a knob): the B operand of the r-th rotation of a 16 x 16 block is the value that the lane 4 r
// further up its 16-lane row holds for r = 0 (column group (b + r) & 3 instead of b), and the r = 0 value of block column t IS the A
// operand of block row t — so ONE LDS read per block column and k-step would do, the rotations being DPP row rotations (row_ror) of it:
// NB reads instead of 5 NB.  Correct, and SLOWER (2^20 rows; 16 / 32 / 64 columns: 34 -> 36, 65 -> 74, 157 -> 171 us): two v_mov_dpp per
// rotated double in the issue stream of the wave that also issues the MFMAs cost more than the LDS reads they replace, which the LDS
// pipe serves beside the matrix pipe.  profiles/r05_gram_shapes.txt.
#ifndef PMT_TALL_DPP
#define PMT_TALL_DPP 0
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffLL), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// value of the lane (lane + 4 r) mod 16 of the same row: row_ror by 16 - 4 r
template <int R>
__device__ __forceinline__ double rot_blocks(double v) {
    if (R == 0) return v;
    if (R == 1) return dpp_row<0x12C>(v);
    if (R == 2) return dpp_row<0x128>(v);
    return dpp_row<0x124>(v);
}

// one stage of one wave from the panel in LDS (`panel` already points at this lane's k offset).
// PMT_TALL_K2: the MFMA's contraction slot k = lane >> 4 of k-steps 2j and 2j + 1 is given the ADJACENT rows 8j + 2k and 8j + 2k + 1 (any
// assignment of rows to slots is a valid contraction as long as both operands use it), so ONE 16-byte LDS read per operand serves two
// k-steps: half the LDS instructions and half the operand waits per MFMA.
template <int NBC, int W>
__device__ __forceinline__ void tall_stage(const double *__restrict__ panel, int lm, double (&acc)[tall_tacc(NBC)]) {
    constexpr int NR = tw_nr(NBC, W), NC = tw_nc(NBC, W), NBLK = tw_nblk(NBC, W);
#if PMT_TALL_K2
#pragma unroll PMT_TALL_UNROLL
    for (int kk = 0; kk < TBK / 8; ++kk) {
        f64x2 a[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) a[t] = *reinterpret_cast<const f64x2 *>(panel + (tw_row(NBC, W, t) * 16 + lm) * TGP + kk * 8);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f64x2 bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rc = ((((lm >> 2) + r) & 3) << 2) | (lm & 3);      // column group rotated by r blocks (gram_sk.hip, lane maps)
                bv[r] = *reinterpret_cast<const f64x2 *>(panel + (tw_col(NBC, W, c) * 16 + rc) * TGP + kk * 8);
            }
#pragma unroll
            for (int k = 0; k < NBLK; ++k) {
                if (tw_blk(NBC, W, k, 1) != c) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (PMT_TALL_DIAG3 && r == 3 && tw_row(NBC, W, tw_blk(NBC, W, k, 0)) == tw_col(NBC, W, c)) continue;      // (diagonal block: see below)
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tw_blk(NBC, W, k, 0)].x, bv[r].x, acc[k * 4 + r], 0, 0, 0);
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tw_blk(NBC, W, k, 0)].y, bv[r].y, acc[k * 4 + r], 0, 0, 0);
                }
            }
        }
    }
#else
#pragma unroll PMT_TALL_UNROLL
    for (int ks = 0; ks < TBK / 4; ++ks) {
        double a[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) a[t] = panel[(tw_row(NBC, W, t) * 16 + lm) * TGP + ks * 4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rc = ((((lm >> 2) + r) & 3) << 2) | (lm & 3);      // column group rotated by r blocks (gram_sk.hip, lane maps)
                bv[r] = panel[(tw_col(NBC, W, c) * 16 + rc) * TGP + ks * 4];
            }
#pragma unroll
            for (int k = 0; k < NBLK; ++k) {
                if (tw_blk(NBC, W, k, 1) != c) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // a DIAGONAL block's third rotation — 4 x 4 sub-blocks (b, b + 3) — holds the transposes of the first rotation's
                    // (b + 1, b): never computed, the fix-up reads the (3, 0) sub-block of rotation 1 as (0, 3) (tall_diag_rule)
                    if (PMT_TALL_DIAG3 && r == 3 && tw_row(NBC, W, tw_blk(NBC, W, k, 0)) == tw_col(NBC, W, c)) continue;
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tw_blk(NBC, W, k, 0)], bv[r], acc[k * 4 + r], 0, 0, 0);
                }
            }
        }
    }
#endif
}

// thread (kp = tid & 7, cc = tid >> 3) owns the row pairs 16 j + 2 kp (j < TSUB) of the columns cc + 32 p (p < 4): the TSUB loads of a
// column are issued back to back, so the memory system sees 8 * TBK contiguous bytes per column and workgroup at a time
// FAST (whole stages, A 16-byte aligned with an even pitch, b aligned): unmasked 16-byte loads; a column beyond the matrix reads the LAST
// column instead (its products land in blocks nobody reads, its q in entries the fix-up drops) and the 32-column groups beyond the NBC
// block columns the kernel is built for are not loaded at all.
template <int NBC, bool FAST>
__device__ __forceinline__ void tall_load(const TallArgs &g, int64_t row0, int64_t rend, int kp, int cc, f64x2 (&reg)[4][TSUB], f64x2 (&cv)[TSUB]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (32 * p >= 16 * NBC) {                       // (no block of this kernel reads these columns of the panel)
#pragma unroll
            for (int j = 0; j < TSUB; ++j) { reg[p][j].x = 0.0; reg[p][j].y = 0.0; }
            continue;
        }
        const int64_t col = cc + 32 * p;
        const int64_t ccol = FAST ? min(col, g.cols - 1) : col;
#pragma unroll
        for (int j = 0; j < TSUB; ++j) {
            const int64_t row = row0 + 16 * j + 2 * kp;
            const double *src = g.A + ccol * g.lda + row;
            f64x2 v;
            if (FAST) {
                v = *reinterpret_cast<const f64x2 *>(src);
            } else {
                v.x = 0.0; v.y = 0.0;
                if (col < g.cols) {
                    if (g.vec_in && row + 1 < rend) v = *reinterpret_cast<const f64x2 *>(src);
                    else {
                        if (row < rend) v.x = src[0];
                        if (row + 1 < rend) v.y = src[1];
                    }
                }
            }
            reg[p][j] = v;
        }
    }
    // b's row pairs come in RAW: no arithmetic on a loaded value in front of the stage's MFMAs — an `s_waitcnt vmcnt(0)` for it would also
    // wait for the panel loads above, i.e. serialise the memory phase with the matrix phase (measured: 0.51 ms = 0.26 memory + 0.33 MFMA)
#pragma unroll
    for (int j = 0; j < TSUB; ++j) {
        const int64_t row = row0 + 16 * j + 2 * kp;
        f64x2 v;
        v.x = 0.0; v.y = 0.0;
        if (g.b) {
            if (FAST) {
                v = *reinterpret_cast<const f64x2 *>(g.b + row);
            } else {
                if (row < rend) v.x = g.b[row];
                if (row + 1 < rend) v.y = g.b[row + 1];
            }
        }
        cv[j] = v;
    }
}

// registers -> LDS (16-byte stores), and the affine part / constant of the rows just loaded
__device__ __forceinline__ void tall_store(double *__restrict__ panel, const f64x2 (&reg)[4][TSUB], const f64x2 (&cv)[TSUB], int sign, int kp, int cc,
                                           double (&qacc)[4], double &cacc) {
#pragma unroll
    for (int j = 0; j < TSUB; ++j) {
        // c = 0.0 (+|-) b (rows beyond the matrix were loaded as 0: they add 0 * 0)
        const double c0 = signed_const(cv[j].x, sign), c1 = signed_const(cv[j].y, sign);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f64x2 *>(panel + (cc + 32 * p) * TGP + 16 * j + 2 * kp) = reg[p][j];
            qacc[p] = qacc[p] + c0 * reg[p][j].x;
            qacc[p] = qacc[p] + c1 * reg[p][j].y;
        }
        cacc = cacc + c0 * c0;
        cacc = cacc + c1 * c1;
    }
}

// the whole row chunk of one workgroup as seen by wave W.  The wave's block set is a template parameter of the WHOLE body, not of the
// stage alone: with four stage bodies behind a branch inside one loop the accumulators are a 36-double phi at every merge and the
// allocator needs ~230 registers; specialised from the top each wave has its own 36 accumulators in fixed registers (~145).  All four
// bodies execute the same sequence of barriers (s_barrier counts waves, the branch is wave-uniform).
template <int NBC, int W, bool FAST>
__device__ __forceinline__ void tall_body(const TallArgs &g, double (&lds)[2][TCOLS * TGP], int tid) {
    constexpr int NACC = tall_tacc(NBC);
    const int lane = tid & 63;
    const int lm = lane & 15, lk = lane >> 4;
    const int kp = tid & 7, cc = tid >> 3;
    const int64_t G = gridDim.x, bid = blockIdx.x;
    const int64_t sbeg = g.interleave ? bid : bid * g.chunk;                       // first stage and stage step of this workgroup
    const int64_t sstep = g.interleave ? G : 1;
    const int nstage = (int)(g.interleave ? (g.nstages > bid ? (g.nstages - bid + G - 1) / G : 0) : max((int64_t)0, min(g.chunk, g.nstages - sbeg)));
    const int64_t rend = g.rows;

    double acc[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.0;
    double qacc[4] = {0.0, 0.0, 0.0, 0.0}, cacc = 0.0;
    auto stage_row = [&](int s) { return (sbeg + (int64_t)s * sstep) * TBK; };
    f64x2 reg[4][TSUB], cv[TSUB];

    if (nstage > 0) {
        tall_load<NBC, FAST>(g, stage_row(0), rend, kp, cc, reg, cv);
        tall_store(lds[0], reg, cv, g.sign, kp, cc, qacc, cacc);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nstage;
#if PMT_TALL_ABL == 1      // ablation (wrong results): no global loads / LDS stores after the first stage — the MFMA side alone
        tall_stage<NBC, W>(lds[cur] + TLK * lk, lm, acc);
#elif PMT_TALL_ABL == 2    // ablation (wrong results): no MFMAs — the memory side alone
        if (more) tall_load<NBC, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        if (more) tall_store(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#elif PMT_TALL_ABL == 3    // ablation (wrong results): global loads + MFMAs, no register -> LDS phase (one add keeps the loads alive)
        if (more) tall_load<NBC, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        tall_stage<NBC, W>(lds[cur] + TLK * lk, lm, acc);
        if (more) { cacc = cacc + reg[0][0].x; cacc = cacc + reg[3][TSUB - 1].y; cacc = cacc + cv[0].x; }
#elif PMT_TALL_ABL == 4    // ablation (wrong results): MFMAs + the register -> LDS phase of stale registers, no global loads after the first stage
        tall_stage<NBC, W>(lds[cur] + TLK * lk, lm, acc);
        if (more) tall_store(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#elif PMT_TALL_ABL == 5    // ablation (wrong results): everything but the q / c'c arithmetic (LDS stores only)
        if (more) tall_load<NBC, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        tall_stage<NBC, W>(lds[cur] + TLK * lk, lm, acc);
        if (more) {
#pragma unroll
            for (int j = 0; j < TSUB; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f64x2 *>(lds[cur ^ 1] + (cc + 32 * p) * TGP + 16 * j + 2 * kp) = reg[p][j];
            cacc = cacc + cv[0].x;
        }
#else
        if (more) tall_load<NBC, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        tall_stage<NBC, W>(lds[cur] + TLK * lk, lm, acc);
        if (more) tall_store(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#endif
        __syncthreads();
    }

    // partials -> workspace: [wave][accumulator][lane] (512-byte runs), then q and c'c
    double *w = g.ws + (int64_t)blockIdx.x * tall_stride(NBC);
#pragma unroll
    for (int r = 0; r < NACC; ++r) w[(W * NACC + r) * 64 + lane] = acc[r];
    // the 8 row-pair threads of a column are 8 consecutive lanes: tree in fixed order
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double v = qacc[p];
        v = v + __shfl_down(v, 4, 8);
        v = v + __shfl_down(v, 2, 8);
        v = v + __shfl_down(v, 1, 8);
        if (kp == 0) w[tall_part(NBC) + cc + 32 * p] = v;
    }
    if (W == 0) {
        double v = cacc;
        v = v + __shfl_down(v, 4, 8);
        v = v + __shfl_down(v, 2, 8);
        v = v + __shfl_down(v, 1, 8);
        if (tid == 0) w[tall_part(NBC) + TCOLS] = v;
    }
}

template <int NBC, bool FAST>
__global__ __launch_bounds__(256, PMT_TALL_WPS) void gram_tall_kernel(TallArgs g) {
    __shared__ double lds[2][TCOLS * TGP];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.y: the DIAGONAL tile this workgroup works on (wide tall matrices: every 128-column diagonal tile of A'A is this kernel's
    // triangle problem on the tile's own columns; the off-diagonal tiles are the stream-K kernel's, gram.hip)
    const int64_t tile = blockIdx.y;
    g.A += tile * TCOLS * g.lda;
    g.cols = min((int64_t)TCOLS, g.cols - tile * TCOLS);
    g.ws += tile * (int64_t)gridDim.x * tall_stride(NBC);
    if (wave == 0) tall_body<NBC, 0, FAST>(g, lds, tid);
    else if (wave == 1) tall_body<NBC, 1, FAST>(g, lds, tid);
    else if (wave == 2) tall_body<NBC, 2, FAST>(g, lds, tid);
    else tall_body<NBC, 3, FAST>(g, lds, tid);
}

struct TallFixArgs {
    const double *ws; int G;
    int nb;                               // 0: the dealt layout of gram_tall_kernel<nbc>; NB > 0: gram_narrow_kernel<NB>'s (blocks in packed upper order)
    int nbc;                              // nb == 0: block columns of the tall kernel that wrote the partials (5 .. 8)
    int part, pcols, stride;              // doubles of triangle partial, columns of the panel, doubles per workgroup
    int64_t cols; const int64_t *xvar; const int64_t *varmap; int moi;
    QT *out_quad; double *out_csc; double alpha; LT *out_lin; double *out_const;
};

// element e of the partial layout summed over the G workgroups: 16 interleaved slices (slice t adds the chunks t, t + 16, .. in order),
// then the slices in order; 64 consecutive elements per workgroup (512-byte reads).  Then the element goes where it belongs:
// QuadraticTerm (x 2: (j,k)+(k,j) combined off the diagonal, the MOI doubling on it), LinearTerm 2 * q_j, or the constant.
__global__ __launch_bounds__(1024) void gram_tall_fixup_kernel(TallFixArgs f) {
    __shared__ double part[TSLICES][64];
    const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int64_t tile = blockIdx.y, j0 = tile * TCOLS;          // (diagonal tile of a wide matrix: global indices j0 + ..)
    double sum = 0.0;
    const int nel = f.part + f.pcols + 1;
    const int64_t stride = f.stride;
    if (e < nel) {
        const double *p = f.ws + tile * (int64_t)f.G * stride + e;
        int gidx = slice;
        for (; gidx + 7 * TSLICES < f.G; gidx += 8 * TSLICES) {          // eight loads in flight per thread (the additions keep their order)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(gidx + u * TSLICES) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum = sum + v[u];
        }
        for (; gidx + 3 * TSLICES < f.G; gidx += 4 * TSLICES) {
            const double v0 = p[(int64_t)gidx * stride], v1 = p[(int64_t)(gidx + TSLICES) * stride];
            const double v2 = p[(int64_t)(gidx + 2 * TSLICES) * stride], v3 = p[(int64_t)(gidx + 3 * TSLICES) * stride];
            sum = sum + v0; sum = sum + v1; sum = sum + v2; sum = sum + v3;
        }
        for (; gidx < f.G; gidx += TSLICES) sum = sum + p[(int64_t)gidx * stride];
    }
    part[slice][el] = sum;
    __syncthreads();
    if (slice != 0 || e >= nel) return;
    double v = part[0][el];
#pragma unroll
    for (int t = 1; t < TSLICES; ++t) v = v + part[t][el];
    const int64_t n = f.cols;
    if (e < f.part) {
        const int lane = e & 63, a = e >> 6;              // a = wave * TACC + block * 4 + rotation  (narrow: block * 4 + rotation)
        const int r = a & 3;
        int tm, tn;
        if (f.nb) {                                       // block k of the packed upper order: (tm, tn) = (k - tn (tn + 1) / 2, tn)
            const int k = a >> 2;
            tn = k >= 6 ? 3 : k >= 3 ? 2 : k >= 1 ? 1 : 0;
            tm = k - tn * (tn + 1) / 2;
        } else {
            const int tacc = tall_tacc(f.nbc);
            const int w = a / tacc, k = (a % tacc) >> 2;
            tm = TALL_BLOCKS[f.nbc - 5][w][k][0]; tn = TALL_BLOCKS[f.nbc - 5][w][k][1];
            if (tm < 0) return;                           // (a padding slot of a wave with fewer blocks)
        }
        const int i = lane >> 4, b = (lane >> 2) & 3, jj = lane & 3;
        int64_t j = j0 + 16 * tm + 4 * b + i, kk = j0 + 16 * tn + 4 * ((b + r) & 3) + jj;
        if (tm == tn) {
            // tall_diag_rule — a diagonal block's 4 x 4 sub-blocks (b, (b + r) & 3): rotation 0 is the diagonal sub-blocks (their upper halves
            // are taken), rotation 1 gives (0,1), (1,2), (2,3) and, as its transpose (the same products in the same order: the same bits),
            // (0,3) out of (3,0); rotation 2 gives (0,2), (1,3) (its other two are their transposes); rotation 3 is never computed
            if (r == 3 || (r == 2 && j > kk) || (r == 0 && j > kk)) return;
            if (r == 1 && j > kk) { const int64_t t = j; j = kk; kk = t; }
        }
        if (kk >= n || j > kk) return;
        double c = v;
        if (f.moi || j != kk) c = 2 * c;
        if (f.out_csc) f.out_csc[kk * (kk + 1) / 2 + j] = f.alpha * c;
        if (f.out_quad) {
            const int64_t jv = f.xvar[j], kv = f.xvar[kk];
            u64 *o = reinterpret_cast<u64 *>(f.out_quad) + (j * n - (j * (j - 1)) / 2 + (kk - j)) * 3;
            o[0] = (u64)__double_as_longlong(c);
            o[1] = (u64)(f.moi ? map_var(f.varmap, jv) : jv);
            o[2] = (u64)(f.moi ? map_var(f.varmap, kv) : kv);
        }
    } else if (e < f.part + f.pcols) {
        const int64_t j = j0 + (e - f.part);
        if (j >= n) return;
        LT t;
        t.coeff = 2 * v;
        const int64_t xv = f.xvar[j];
        t.var = f.moi ? map_var(f.varmap, xv) : xv;
        f.out_lin[j] = t;
    } else if (tile == 0) {
        *f.out_const = v;                 // (every tile's workgroups sum the same c'c: the first tile's is the node's)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// NARROW tall matrices (at most 64 columns): 16 FLOP per byte of A at 128 columns become (n + 1) / 8 — 8 at 64 columns, 2 at 16 — and the
// node is bound by the HBM stream alone; the 128-column kernel above spends the same 36 blocks of MFMAs on it whatever n is
// (2^20 x 16: 0.34 ms for 134 MB).  Here the panel is 16 NB columns (NB = 1, 2, 4) and the stage 4096 / (16 NB) rows — the same 32 KB of A
// per stage and workgroup, in pieces of 8 * R contiguous bytes per column —, there are only NB (NB + 1) / 2 blocks (1, 3, 10), and instead of
// dealing blocks out to the four waves every wave takes ALL blocks over its own quarter of the stage's rows (a split of the contraction
// index: the same operand reads per MFMA, a quarter of the accumulator traffic at the end, and the waves' sums are added in wave order
// through LDS before the workgroup's partial is written).  Loads, LDS layout (pitch R + 2), q and c'c on the VALU, interleaved stages and
// the fix-up are the tall kernel's.  NB = 1: 16 lanes (not 8) walk down a column, so that a column run is still >= 256 contiguous bytes.
#ifndef PMT_NARROW_STAGE
#define PMT_NARROW_STAGE 4096      // doubles per stage and workgroup (NB = 4 always 4096: the waves' sums are folded through the panels)
#endif
#ifndef PMT_NARROW_NT
#define PMT_NARROW_NT 0            // A loaded with the nontemporal policy
#endif
#ifndef PMT_NARROW_WPS
#define PMT_NARROW_WPS 2
#endif
#ifndef PMT_NARROW_MAXG
#define PMT_NARROW_MAXG 512
#endif
#ifndef PMT_NARROW_ABL
#define PMT_NARROW_ABL 0           // ablations (wrong results): 1 no loads after the first stage, 2 no MFMAs
#endif
template <int NB> struct Narrow {
    static constexpr int C = 16 * NB;                    // columns of the panel
    static constexpr int LPC = NB == 1 ? 16 : 8;         // lanes per column run (row pairs 2 kp of a piece)
    static constexpr int NCC = 256 / LPC;                // column runs per slot over the workgroup
    static constexpr int STAGE = NB == 4 ? 4096 : PMT_NARROW_STAGE;
    static constexpr int SLOTS = STAGE / 512;            // 16-byte loads per thread and stage
    static constexpr int R = STAGE / C;                  // rows per stage
    static constexpr int PITCH = R + 2;                  // (2 mod 32, as TGP)
    static constexpr int PIECE = 2 * LPC;                // rows per piece
    static constexpr int NQ = C > NCC ? C / NCC : 1;     // distinct columns per thread
    static constexpr int NJ = SLOTS / NQ;                // pieces of a column per thread and stage (= R / PIECE)
    static constexpr int NBLK = NB * (NB + 1) / 2;
    static constexpr int NACC = NBLK * 4;
    static constexpr int PART = NACC * 64;               // doubles of triangle partial per workgroup
    static constexpr int STRIDE = PART + C + 8;
    static constexpr int KSTEPS = R / 16;                // k-steps per wave and stage
    static_assert(NJ * PIECE == R && NQ * NCC >= C, "narrow panel geometry");
};
// 48 columns: the STREAM form only (gram_stream_kernel<3>: six blocks instead of the ten a 33 .. 48-column matrix paid for as a 64-column panel);
// the panel kernel keeps its power-of-two panels
template <> struct Narrow<3> {
    static constexpr int C = 48, NBLK = 6, NACC = NBLK * 4, PART = NACC * 64, STRIDE = PART + C + 8;
};

template <int NB, bool FAST>
__device__ __forceinline__ void narrow_load(const TallArgs &g, int64_t row0, int64_t rend, int kp, int cc,
                                            f64x2 (&reg)[Narrow<NB>::NQ][Narrow<NB>::NJ], f64x2 (&cv)[Narrow<NB>::NJ]) {
    using N = Narrow<NB>;
#pragma unroll
    for (int q = 0; q < N::NQ; ++q) {
        const int64_t col = cc + N::NCC * q;
#pragma unroll
        for (int j = 0; j < N::NJ; ++j) {
            const int64_t row = row0 + N::PIECE * j + 2 * kp;
            const double *src = g.A + col * g.lda + row;
            f64x2 v;
            if (FAST) {
                v = PMT_NARROW_NT ? __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(src)) : *reinterpret_cast<const f64x2 *>(src);
            } else {
                v.x = 0.0; v.y = 0.0;
                if (col < g.cols) {
                    if (g.vec_in && row + 1 < rend) v = *reinterpret_cast<const f64x2 *>(src);
                    else {
                        if (row < rend) v.x = src[0];
                        if (row + 1 < rend) v.y = src[1];
                    }
                }
            }
            reg[q][j] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < N::NJ; ++j) {                    // b comes in raw (see tall_load)
        const int64_t row = row0 + N::PIECE * j + 2 * kp;
        f64x2 v;
        v.x = 0.0; v.y = 0.0;
        if (g.b) {
            if (FAST) {
                v = *reinterpret_cast<const f64x2 *>(g.b + row);
            } else {
                if (row < rend) v.x = g.b[row];
                if (row + 1 < rend) v.y = g.b[row + 1];
            }
        }
        cv[j] = v;
    }
}

template <int NB>
__device__ __forceinline__ void narrow_store(double *__restrict__ panel, const f64x2 (&reg)[Narrow<NB>::NQ][Narrow<NB>::NJ],
                                             const f64x2 (&cv)[Narrow<NB>::NJ], int sign, int kp, int cc, double (&qacc)[Narrow<NB>::NQ], double &cacc) {
    using N = Narrow<NB>;
#pragma unroll
    for (int j = 0; j < N::NJ; ++j) {
        const double c0 = signed_const(cv[j].x, sign), c1 = signed_const(cv[j].y, sign);
#pragma unroll
        for (int q = 0; q < N::NQ; ++q) {
            *reinterpret_cast<f64x2 *>(panel + (cc + N::NCC * q) * N::PITCH + N::PIECE * j + 2 * kp) = reg[q][j];
            qacc[q] = qacc[q] + c0 * reg[q][j].x;
            qacc[q] = qacc[q] + c1 * reg[q][j].y;
        }
        cacc = cacc + c0 * c0;
        cacc = cacc + c1 * c1;
    }
}

// this wave's quarter of a stage: every block, KSTEPS k-steps (`panel` points at the wave's first row + this lane's k offset).
// PMT_NARROW_K2 (as PMT_TALL_K2): the contraction slot k = lane >> 4 of k-steps 2j and 2j + 1 takes the ADJACENT rows 8j + 2k, 8j + 2k + 1, so
// ONE 16-byte LDS read per operand serves two k-steps.
#ifndef PMT_NARROW_K2
#define PMT_NARROW_K2 1
#endif
template <int NB>
__device__ __forceinline__ void narrow_stage(const double *__restrict__ panel, int lm, double (&acc)[Narrow<NB>::NACC]) {
    using N = Narrow<NB>;
#if PMT_NARROW_K2
    static_assert(N::KSTEPS % 2 == 0, "k-steps in pairs");
#pragma unroll
    for (int kk = 0; kk < N::KSTEPS / 2; ++kk) {
        f64x2 a[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) a[t] = *reinterpret_cast<const f64x2 *>(panel + (t * 16 + lm) * N::PITCH + kk * 8);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            f64x2 bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (PMT_TALL_DIAG3 && r == 3 && c == 0) { bv[r].x = 0.0; bv[r].y = 0.0; continue; }
                const int rc = ((((lm >> 2) + r) & 3) << 2) | (lm & 3);
                bv[r] = *reinterpret_cast<const f64x2 *>(panel + (c * 16 + rc) * N::PITCH + kk * 8);
            }
#pragma unroll
            for (int tm = 0; tm <= c; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (PMT_TALL_DIAG3 && r == 3 && tm == c) continue;    // (diagonal block: tall_stage)
                    const int k = c * (c + 1) / 2 + tm;
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm].x, bv[r].x, acc[k * 4 + r], 0, 0, 0);
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm].y, bv[r].y, acc[k * 4 + r], 0, 0, 0);
                }
        }
    }
#else
#pragma unroll
    for (int ks = 0; ks < N::KSTEPS; ++ks) {
        double a[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) a[t] = panel[(t * 16 + lm) * N::PITCH + ks * 4];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            double bv[4];
#if PMT_TALL_DPP
            bv[0] = a[c]; bv[1] = rot_blocks<1>(a[c]); bv[2] = rot_blocks<2>(a[c]); bv[3] = rot_blocks<3>(a[c]);
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r == 3 && c == 0) { bv[r] = 0.0; continue; }         // (column 0 has the diagonal block only)
                const int rc = ((((lm >> 2) + r) & 3) << 2) | (lm & 3);
                bv[r] = panel[(c * 16 + rc) * N::PITCH + ks * 4];
            }
#endif
#pragma unroll
            for (int tm = 0; tm <= c; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (PMT_TALL_DIAG3 && r == 3 && tm == c) continue;    // (diagonal block: tall_stage)
                    const int k = c * (c + 1) / 2 + tm;
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tm], bv[r], acc[k * 4 + r], 0, 0, 0);
                }
        }
    }
#endif
}

template <int NB, bool FAST>
__global__ __launch_bounds__(256, PMT_NARROW_WPS) void gram_narrow_kernel(TallArgs g) {
    using N = Narrow<NB>;
    __shared__ double lds[2][N::C * N::PITCH];
    static_assert(3 * N::NACC * 64 <= 2 * N::C * N::PITCH, "the waves' sums are folded through the panels");
    static_assert(N::KSTEPS >= 1, "a stage holds at least 16 rows per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, lk = lane >> 4;
    const int kp = tid % N::LPC, cc = tid / N::LPC;
    const int64_t G = gridDim.x, bid = blockIdx.x;
    const int nstage = (int)(g.nstages > bid ? (g.nstages - bid + G - 1) / G : 0);      // stages bid, bid + G, ..
    const int64_t rend = g.rows;

    double acc[N::NACC];
#pragma unroll
    for (int r = 0; r < N::NACC; ++r) acc[r] = 0.0;
    double qacc[N::NQ], cacc = 0.0;
#pragma unroll
    for (int q = 0; q < N::NQ; ++q) qacc[q] = 0.0;
    f64x2 reg[N::NQ][N::NJ], cv[N::NJ];
    auto stage_row = [&](int s) { return (bid + (int64_t)s * G) * N::R; };

    if (nstage > 0) {
        narrow_load<NB, FAST>(g, stage_row(0), rend, kp, cc, reg, cv);
        narrow_store<NB>(lds[0], reg, cv, g.sign, kp, cc, qacc, cacc);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nstage;
#if PMT_NARROW_ABL == 1
        narrow_stage<NB>(lds[cur] + wave * (N::R / 4) + (PMT_NARROW_K2 ? 2 : 1) * lk, lm, acc);
        if (more) narrow_store<NB>(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#elif PMT_NARROW_ABL == 2
        if (more) narrow_load<NB, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        if (more) narrow_store<NB>(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#else
        if (more) narrow_load<NB, FAST>(g, stage_row(s + 1), rend, kp, cc, reg, cv);
        narrow_stage<NB>(lds[cur] + wave * (N::R / 4) + (PMT_NARROW_K2 ? 2 : 1) * lk, lm, acc);
        if (more) narrow_store<NB>(lds[cur ^ 1], reg, cv, g.sign, kp, cc, qacc, cacc);
#endif
        __syncthreads();
    }

    // the four waves' sums over their quarters, added in wave order; then q (one LPC-lane tree per column) and c'c (the first column run's)
    double *red = &lds[0][0];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < N::NACC; ++r) red[((wave - 1) * N::NACC + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    double *w = g.ws + (int64_t)blockIdx.x * N::STRIDE;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < N::NACC; ++r) {
            double v = acc[r];
            v = v + red[(0 * N::NACC + r) * 64 + lane];
            v = v + red[(1 * N::NACC + r) * 64 + lane];
            v = v + red[(2 * N::NACC + r) * 64 + lane];
            w[r * 64 + lane] = v;
        }
    }
#pragma unroll
    for (int q = 0; q < N::NQ; ++q) {
        double v = qacc[q];
#pragma unroll
        for (int h = N::LPC / 2; h >= 1; h >>= 1) v = v + __shfl_down(v, h, N::LPC);
        if (kp == 0) w[N::PART + cc + N::NCC * q] = v;
    }
    if (wave == 0) {
        double v = cacc;
#pragma unroll
        for (int h = N::LPC / 2; h >= 1; h >>= 1) v = v + __shfl_down(v, h, N::LPC);
        if (tid == 0) w[N::PART + N::C] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// STREAM form of the narrow panels (round 6).  The panel kernel above moves every stage global -> registers -> LDS ->
// barrier -> LDS operand reads -> MFMA: its memory side alone streams at 5.8 TB/s and its matrix side alone needs 18 us at 16 columns, but
// the two phases of a workgroup overlap poorly (29 us; profiles/r05_gram_shapes.txt) and the fix-up is a second launch.  Here there is no
// LDS and no barrier on the way: every WAVE streams its own row pieces straight into the MFMA operand layout — lane (lm = lane & 15,
// lk = lane >> 4) loads the 16 bytes of column 16 t + lm at rows 8 i + 2 lk, 8 i + 2 lk + 1 of its iteration: exactly the two values the
// contraction slots k = lk of k-steps 2 i and 2 i + 1 take for BOTH operands (any assignment of rows to slots is a valid contraction as
// long as both operands use it: narrow_stage) — D iterations in flight per wave in registers, and the rotated B operands of a block column
// are DPP row rotations of the A operand (rot_blocks; in the panel kernel they cost more than the LDS reads they replaced, here there is
// no LDS pipe to lean on and the VALU is idle).  A load instruction touches 64 contiguous bytes of each of 16 columns; the wave's IT
// instructions of an iteration cover 64 IT contiguous bytes per column, neighbouring waves neighbouring pieces (iteration = global wave
// index + k * waves: the chip reads one band of consecutive rows at a time, as the panel kernels do).  q = A'c and c'c ride on the VALU
// with the loaded values.  The four waves of a workgroup add their sums in wave order through LDS once, at the end; the workgroups'
// partials have the panel kernel's layout (gram_tall_fixup_kernel adds them).
#ifndef PMT_STREAM
#define PMT_STREAM 1               // 0: the panel kernel (gram_narrow_kernel) for every narrow shape
#endif
#ifndef PMT_STREAM_D1
#define PMT_STREAM_D1 3            // iterations in flight per wave: 16-column panels,
#endif
#ifndef PMT_STREAM_D2
#define PMT_STREAM_D2 3            // 32,
#endif
#ifndef PMT_STREAM_D3
#define PMT_STREAM_D3 2            // 48,
#endif
#ifndef PMT_STREAM_D4
#define PMT_STREAM_D4 2            // 64
#endif
#ifndef PMT_STREAM_IT4
#define PMT_STREAM_IT4 2           // 16-byte loads per column group, lane and iteration at 64 columns
#endif
#ifndef PMT_STREAM_WPS
#define PMT_STREAM_WPS 2
#endif
#ifndef PMT_STREAM_WPS12
#define PMT_STREAM_WPS12 1         // 16 / 32 columns run one workgroup per CU (PMT_STREAM_MAXG): the whole register file for a round of D iterations as one block (256 registers: 20 spilled at 32 columns)
#endif
#ifndef PMT_STREAM_WPS4
#define PMT_STREAM_WPS4 2          // workgroups per CU the 64-column kernel's registers are budgeted for
#endif
#ifndef PMT_STREAM_MAXG
#define PMT_STREAM_MAXG 256        // workgroups at most, 16- and 32-column panels: ONE wave per SIMD streams best (HBM-bound: 2^20 x 16 25 us against 27 with two)
#endif
#ifndef PMT_STREAM_MAXG3
#define PMT_STREAM_MAXG3 512       // 48 columns
#endif
#ifndef PMT_STREAM_MAXG4
#define PMT_STREAM_MAXG4 512       // 64 columns: the matrix pipe matters as much as the stream — two waves per SIMD (256 workgroups: 142 us against 118)
#endif
#ifndef PMT_STREAM_ABL
#define PMT_STREAM_ABL 0           // ablations (wrong results): 1 no MFMAs / rotations, 2 no loads after the first D iterations
#endif
#ifndef PMT_STREAM_SADDR
#define PMT_STREAM_SADDR 1         // scalar-base loads, no branch around b's load, whole rounds of D iterations as one branch-free block
#endif
#ifndef PMT_STREAM_DPP
#define PMT_STREAM_DPP 1           // 0: the rotated operands are loaded again from global memory (L1 hits) instead of DPP rotations
#endif
template <int NB> struct Stream {
    static constexpr int IT = NB == 4 ? PMT_STREAM_IT4 : 4;           // 16-byte loads per column group, lane and iteration (= pairs of k-steps)
    static constexpr int RI = 8 * IT;                    // rows per iteration
    static constexpr int D = NB == 1 ? PMT_STREAM_D1 : NB == 2 ? PMT_STREAM_D2 : NB == 3 ? PMT_STREAM_D3 : PMT_STREAM_D4;
};

// one iteration's loads of one wave: buf[t][i] = rows (row0 + 8 i + 2 lk, + 1) of column 16 t + lm; cb = row pair (lane & (4 IT - 1)) of b, raw —
// ONE coalesced load per iteration; the contraction slots pick their pairs out of the wave's LDS piece (stream_compute).
// FAST: scalar base (A + row0: wave-uniform) + a 32-bit per-lane byte offset (voff[t], fixed for the kernel) + an immediate: no vector
// address arithmetic per load.  Otherwise no branch either: a load outside the matrix reads a clamped (valid) address and is replaced by
// 0.0 afterwards — a branch around a load would make the compiler drain the whole pipeline at every merge (s_waitcnt vmcnt counts loads in order).
// FAST (A 16-byte aligned with an even pitch, b 16-byte aligned, the panel within 4 GiB): a WHOLE iteration's loads — scalar base (A + row0:
// wave-uniform) + a 32-bit per-lane byte offset (voff[t], fixed for the kernel; a column beyond the matrix reads the LAST column instead:
// its products land in accumulators nobody reads) + an immediate: no vector address arithmetic, no mask, no branch.  Otherwise, and for the
// one ragged iteration at the end of the matrix, 8-byte loads from clamped addresses, what lies outside replaced by 0.0 (no branch around a
// load either: s_waitcnt vmcnt counts loads in order, a merge would drain the pipeline).
template <int NB, bool FAST>
__device__ __forceinline__ void stream_load(const TallArgs &g, int64_t row0, int lane, unsigned (&voff)[NB], f64x2 (&buf)[NB][Stream<NB>::IT], f64x2 &cb) {
    using S = Stream<NB>;
    const int lm = lane & 15, lk = lane >> 4, bp = lane & (4 * S::IT - 1);
    if (FAST) {
        const char *base = reinterpret_cast<const char *>(g.A + row0);
#pragma unroll
        for (int t = 0; t < NB; ++t) {
#if PMT_STREAM_SADDR
            // the lane offset passes through an empty asm: the compiler cannot hoist its zero extension out of the loop (it did, and
            // then paid a 64-bit vector add per load instead of the scalar-base form; gram_mid.hip: mid_step)
            asm volatile("" : "+v"(voff[t]));
#endif
#pragma unroll
            for (int i = 0; i < S::IT; ++i) buf[t][i] = *reinterpret_cast<const f64x2 *>(base + voff[t] + 64 * i);
        }
#if PMT_STREAM_SADDR
        // no b: sign is 0 and signed_const ignores what is loaded — any valid address keeps the loop free of branches
        unsigned bo = 16u * (unsigned)bp;
        asm volatile("" : "+v"(bo));
        cb = *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>((g.b ? g.b : g.A) + row0) + bo);
#else
        cb.x = 0.0; cb.y = 0.0;
        if (g.b) cb = *reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(g.b + row0) + 16u * (unsigned)bp);
#endif
        return;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int64_t col = 16 * t + lm;
        const bool cv = col < g.cols;
#pragma unroll
        for (int i = 0; i < S::IT; ++i) {
            const int64_t row = row0 + 8 * i + 2 * lk;
            const bool v0 = cv && row < g.rows, v1 = cv && row + 1 < g.rows;
            const double *p0 = g.A + (v0 ? col * g.lda + row : 0), *p1 = g.A + (v1 ? col * g.lda + row + 1 : 0);
            const double x = *p0, y = *p1;
            buf[t][i].x = v0 ? x : 0.0;
            buf[t][i].y = v1 ? y : 0.0;
        }
    }
    cb.x = 0.0; cb.y = 0.0;
    if (g.b) {
        const int64_t row = row0 + 2 * bp;
        const bool v0 = row < g.rows, v1 = row + 1 < g.rows;
        const double x = g.b[v0 ? row : 0], y = g.b[v1 ? row + 1 : 0];
        cb.x = v0 ? x : 0.0;
        cb.y = v1 ? y : 0.0;
    }
}

// One iteration of one wave.  The B operand of rotation r of a block column is the A operand of the lane 4 r further up its 16-lane row
// (column group (b + r) & 3 instead of b): the wave stores the values it has loaded into its PRIVATE piece of LDS once (16 bytes per lane,
// linear: no bank conflicts) and reads them back rotated — LDS-pipe instructions beside the MFMAs, no barrier (a wave's LDS operations
// execute in order), no VALU work: DPP rotations of the same values cost 20 v_mov_dpp per 20 MFMAs at 32 columns and doubled the
// kernel's matrix phase (profiles/r06_gram_stream.txt).  q = A'c rides on the matrix pipe too: one more MFMA per column group and
// k-step whose B operand is c itself in EVERY lane of the contraction slot — all 16 columns of the operand are c, so block b of the
// result holds q of the columns 4 b .. 4 b + 3 (four copies), no rotation needed.  Only c = 0.0 (+|-) b and c'c stay on the VALU
// (the constant's order is restated bit for bit by the tests: pmt_quad_gram_constant_order, order 4).
template <int NB>
__device__ __forceinline__ void stream_compute(double *__restrict__ rot, const f64x2 (&buf)[NB][Stream<NB>::IT], const f64x2 &cb, int sign,
                                               int lane, double (&acc)[Narrow<NB>::NACC], double (&qacc)[NB], double &cacc) {
    using S = Stream<NB>;
    const int lm = lane & 15, lrow = lane & 48, lk = lane >> 4;
    double *__restrict__ rotb = rot + NB * S::IT * 128;              // c's row pairs: lane l < 4 IT holds pair l
    // c = 0.0 (+|-) b ONCE per loaded value, before the pairs go to LDS (round 6c): done behind the read-back, every contraction slot
    // recomputed it for its pair — 12 vector ALU instructions per pair of k-steps in the wave's one issue stream (48 of the 64 beside
    // 32 MFMAs per iteration at 16 columns), and vector ALU work does not hide behind the wave's own f64 MFMAs.  Same values, same bits.
    f64x2 cs;
    cs.x = signed_const(cb.x, sign); cs.y = signed_const(cb.y, sign);
    *reinterpret_cast<f64x2 *>(rotb + lane * 2) = cs;
#if PMT_STREAM_ABL != 4
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int i = 0; i < S::IT; ++i) *reinterpret_cast<f64x2 *>(rot + ((t * S::IT + i) * 64 + lane) * 2) = buf[t][i];
#endif
#pragma unroll
    for (int i = 0; i < S::IT; ++i) {
#pragma unroll
        for (int c = 0; c < (PMT_STREAM_ABL == 1 ? 0 : NB); ++c) {
            f64x2 bv[4];
            bv[0] = buf[c][i];
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                if (r == 3 && c == 0) { bv[r].x = 0.0; bv[r].y = 0.0; continue; }      // (block column 0 holds the diagonal block only)
                if (PMT_STREAM_ABL == 4) { bv[r] = buf[c][i]; continue; }                  // (ablation: no rotations)
                bv[r] = *reinterpret_cast<const f64x2 *>(rot + ((c * S::IT + i) * 64 + lrow + ((lm + 4 * r) & 15)) * 2);
            }
#pragma unroll
            for (int tm = 0; tm <= c; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r == 3 && tm == c) continue;                                      // (diagonal block: tall_diag_rule)
                    const int k = c * (c + 1) / 2 + tm;
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm][i].x, bv[r].x, acc[k * 4 + r], 0, 0, 0);
                    acc[k * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm][i].y, bv[r].y, acc[k * 4 + r], 0, 0, 0);
                }
        }
        if (PMT_STREAM_ABL == 5) continue;                                                // (ablation: no q / c'c)
        const f64x2 bi = *reinterpret_cast<const f64x2 *>(rotb + (4 * i + lk) * 2);        // rows 8 i + 2 lk, + 1 of c
        const double c0 = bi.x, c1 = bi.y;
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t][i].x, c0, qacc[t], 0, 0, 0);
            qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t][i].y, c1, qacc[t], 0, 0, 0);
        }
        cacc = cacc + c0 * c0;
        cacc = cacc + c1 * c1;
    }
}

template <int NB, bool FAST>
__global__ __launch_bounds__(256, NB == 4 ? PMT_STREAM_WPS4 : NB <= 2 ? PMT_STREAM_WPS12 : PMT_STREAM_WPS) void gram_stream_kernel(TallArgs g) {
    using S = Stream<NB>;
    using N = Narrow<NB>;
    // per wave: the rotation buffer of one iteration (NB IT KB) during the loop; afterwards the same memory carries the waves' sums
    constexpr int ROT = NB * S::IT * 128 + 128, RED = N::NACC * 64 + 16 * NB + 8;
    constexpr int SH = 4 * ROT > RED ? 4 * ROT : RED;
    __shared__ double sh[SH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, lk = lane >> 4;
    const int64_t W = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wave;
    // iterations gw, gw + W, .. of the nfull WHOLE ones; the ragged one behind them (rows % RI rows) is the last iteration of the wave it
    // falls to — loaded with masks, outside the pipelined loop
    const int64_t nfull = g.rows / S::RI;
    const int my = (int)(nfull > gw ? (nfull - gw + W - 1) / W : 0);
    const bool tail = nfull * S::RI < g.rows && nfull % W == gw;

    double acc[N::NACC];
#pragma unroll
    for (int r = 0; r < N::NACC; ++r) acc[r] = 0.0;
    double qacc[NB], cacc = 0.0;
#pragma unroll
    for (int t = 0; t < NB; ++t) qacc[t] = 0.0;
    f64x2 buf[S::D][NB][S::IT], cb[S::D];
    double *rot = sh + wave * ROT;
    unsigned voff[NB];                                    // (FAST: the launch checks that the panel spans less than 4 GiB)
#pragma unroll
    for (int t = 0; t < NB; ++t) voff[t] = (unsigned)((min((int64_t)(16 * t + lm), g.cols - 1) * g.lda + 2 * lk) * 8);
    // (an iteration index beyond the wave's last is clamped to the last: a repeated, cached load that is never used)
    auto row_of = [&](int s) { return (gw + (int64_t)min(s, max(my - 1, 0)) * W) * S::RI; };
    if (my > 0) {
#pragma unroll
        for (int d = 0; d < S::D; ++d) stream_load<NB, FAST>(g, row_of(d), lane, voff, buf[d], cb[d]);
        if (PMT_STREAM_SADDR && FAST && PMT_STREAM_ABL == 0) {
            // whole rounds of D iterations: one block without a branch (the number of loads in flight does not depend on the path, the
            // compiler's vmcnt waits stay counted ones); the loads beyond the wave's last iteration repeat the last one
            int s0 = 0;
            for (; s0 + S::D <= my; s0 += S::D) {
#pragma unroll
                for (int d = 0; d < S::D; ++d) {
                    stream_compute<NB>(rot, buf[d], cb[d], g.sign, lane, acc, qacc, cacc);
                    stream_load<NB, FAST>(g, row_of(s0 + d + S::D), lane, voff, buf[d], cb[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < S::D - 1; ++d)
                if (s0 + d < my) stream_compute<NB>(rot, buf[d], cb[d], g.sign, lane, acc, qacc, cacc);
        } else {
            for (int s0 = 0; s0 < my; s0 += S::D) {
#pragma unroll
                for (int d = 0; d < S::D; ++d) {
                    if (s0 + d < my) stream_compute<NB>(rot, buf[d], cb[d], g.sign, lane, acc, qacc, cacc);
                    if (PMT_STREAM_ABL != 2 && PMT_STREAM_ABL != 4 && PMT_STREAM_ABL != 5) stream_load<NB, FAST>(g, row_of(s0 + d + S::D), lane, voff, buf[d], cb[d]);
                }
            }
        }
    }
    if (tail) {
        stream_load<NB, false>(g, nfull * S::RI, lane, voff, buf[0], cb[0]);
        stream_compute<NB>(rot, buf[0], cb[0], g.sign, lane, acc, qacc, cacc);
    }
    // c'c: the four contraction slots (lanes 0, 16, 32, 48 of column 0), tree in fixed order
    cacc = cacc + __shfl_down(cacc, 32, 64);
    cacc = cacc + __shfl_down(cacc, 16, 64);
    // q of column 16 t + 4 b + i sits in the lanes (i = lane >> 4, b = (lane >> 2) & 3, any lane & 3) of qacc[t]
    const bool qlane = (lane & 3) == 0;
    const int qcol = 4 * ((lane >> 2) & 3) + (lane >> 4);
    __syncthreads();                                      // every wave is done with its rotation buffer
    // the four waves' sums, added in wave order: waves 1, 2, 3 hand theirs to wave 0 one after the other through ONE piece of LDS (three
    // pieces side by side were 63 KB at 64 columns: two workgroups per CU at most)
    double *w = g.ws + (int64_t)blockIdx.x * N::STRIDE;
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int r = 0; r < N::NACC; ++r) sh[r * 64 + lane] = acc[r];
            if (qlane) {
#pragma unroll
                for (int t = 0; t < NB; ++t) sh[N::NACC * 64 + 16 * t + qcol] = qacc[t];
            }
            if (lane == 0) sh[N::NACC * 64 + 16 * NB] = cacc;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < N::NACC; ++r) acc[r] = acc[r] + sh[r * 64 + lane];
#pragma unroll
            for (int t = 0; t < NB; ++t) qacc[t] = qacc[t] + sh[N::NACC * 64 + 16 * t + (qlane ? qcol : 0)];
            cacc = cacc + sh[N::NACC * 64 + 16 * NB];
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < N::NACC; ++r) w[r * 64 + lane] = acc[r];
    if (qlane) {
#pragma unroll
        for (int t = 0; t < NB; ++t) w[N::PART + 16 * t + qcol] = qacc[t];
    }
    if (lane == 0) w[N::PART + N::C] = cacc;
}

// (shape alone decides — the constant's order must follow from (rows, cols): below 32768 rows the panel kernel's two launches are the
// shorter ones: 80 x 50 9.6 against 12 us, 2000 x 40 9.8 against 12.4, 16384 x 64 13.2 against 14.8; 77777 x 50 23.7 against 22.9)
#ifndef PMT_STREAM_MINROWS
#define PMT_STREAM_MINROWS 32768
#endif
static bool stream_form(int nb, int64_t rows) { return PMT_STREAM && nb >= 1 && nb <= 4 && rows >= PMT_STREAM_MINROWS; }
// column groups of the panel: 1, 2, 4 — and 3 (33 .. 48 columns) where the stream form takes the shape
static int narrow_nb(int64_t rows, int64_t cols) {
    if (cols > 32 && cols <= 48 && stream_form(3, rows)) return 3;
    return cols <= 16 ? 1 : cols <= 32 ? 2 : cols <= 64 ? 4 : 0;
}
static int stream_iteration_rows(int nb) { return nb == 1 ? Stream<1>::RI : nb == 2 ? Stream<2>::RI : nb == 3 ? Stream<3>::RI : Stream<4>::RI; }
static int stream_groups(int64_t rows, int nb) {
    const int64_t nit = cdiv(rows, stream_iteration_rows(nb));
    return (int)std::max<int64_t>(1, std::min<int64_t>(nb == 4 ? PMT_STREAM_MAXG4 : nb == 3 ? PMT_STREAM_MAXG3 : PMT_STREAM_MAXG, cdiv(nit, 4)));
}

static int narrow_nb(int64_t rows, int64_t cols);
static int narrow_stage_rows(int nb) { return nb == 1 ? Narrow<1>::R : nb == 2 ? Narrow<2>::R : Narrow<4>::R; }
static int narrow_stride(int nb) { return nb == 1 ? Narrow<1>::STRIDE : nb == 2 ? Narrow<2>::STRIDE : nb == 3 ? Narrow<3>::STRIDE : Narrow<4>::STRIDE; }
static int narrow_groups(int64_t rows, int nb) {
    if (stream_form(nb, rows)) return stream_groups(rows, nb);
    const int64_t nst = cdiv(rows, narrow_stage_rows(nb));
    return (int)cdiv(nst, cdiv(nst, (int64_t)PMT_NARROW_MAXG));
}

template <int NB>
static int launch_gram_narrow(TallArgs g, TallFixArgs f, bool b_aligned, hipStream_t s) {
    using N = Narrow<NB>;
    const int G = narrow_groups(g.rows, NB);
    if (stream_form(NB, g.rows)) {
        g.nstages = cdiv(g.rows, Stream<NB>::RI);                 // iterations of RI rows, dealt out to the 4 G waves
        // aligned pieces of A and b, 32-bit lane offsets: the whole iterations take unmasked 16-byte loads
        const bool fast = g.vec_in && b_aligned && (uint64_t)g.lda * N::C * 8 < (1ull << 32);
        if (fast) PMT_LAUNCH_NAMED("gram_stream_kernel", (gram_stream_kernel<NB, true>), dim3((unsigned)G), dim3(256), 0, s, g);
        else PMT_LAUNCH_NAMED("gram_stream_kernel", (gram_stream_kernel<NB, false>), dim3((unsigned)G), dim3(256), 0, s, g);
        if (int rc = check_launch("gram_stream_kernel")) return rc;
    } else {
        g.nstages = cdiv(g.rows, N::R);
        const bool fast = g.vec_in && g.cols == N::C && g.rows % N::R == 0 && b_aligned;
        if (fast) PMT_LAUNCH_NAMED("gram_narrow_kernel", (gram_narrow_kernel<NB, true>), dim3((unsigned)G), dim3(256), 0, s, g);
        else PMT_LAUNCH_NAMED("gram_narrow_kernel", (gram_narrow_kernel<NB, false>), dim3((unsigned)G), dim3(256), 0, s, g);
        if (int rc = check_launch("gram_narrow_kernel")) return rc;
    }
    f.G = G; f.nb = NB; f.nbc = 8; f.part = N::PART; f.pcols = N::C; f.stride = N::STRIDE;
    PMT_LAUNCH(gram_tall_fixup_kernel, dim3((unsigned)cdiv(N::PART + N::C + 1, 64)), dim3(1024), 0, s, f);
    return check_launch("gram_tall_fixup_kernel");
}

// 33 .. 48 columns from 32768 rows: the stream form on three column groups (no panel kernel of that width exists)
static int launch_gram_stream3(TallArgs g, TallFixArgs f, bool b_aligned, hipStream_t s) {
    using N = Narrow<3>;
    const int G = narrow_groups(g.rows, 3);
    g.nstages = cdiv(g.rows, Stream<3>::RI);
    const bool fast = g.vec_in && b_aligned && (uint64_t)g.lda * N::C * 8 < (1ull << 32);
    if (fast) PMT_LAUNCH_NAMED("gram_stream_kernel", (gram_stream_kernel<3, true>), dim3((unsigned)G), dim3(256), 0, s, g);
    else PMT_LAUNCH_NAMED("gram_stream_kernel", (gram_stream_kernel<3, false>), dim3((unsigned)G), dim3(256), 0, s, g);
    if (int rc = check_launch("gram_stream_kernel")) return rc;
    f.G = G; f.nb = 3; f.nbc = 8; f.part = N::PART; f.pcols = N::C; f.stride = N::STRIDE;
    PMT_LAUNCH(gram_tall_fixup_kernel, dim3((unsigned)cdiv(N::PART + N::C + 1, 64)), dim3(1024), 0, s, f);
    return check_launch("gram_tall_fixup_kernel");
}

// shapes the fused tall form takes: one 128-column tile, any number of rows — the stream-K form's one workgroup per 256 rows of a tile works
// at one CU's rate (>= 34 us for 100 x 100, measured), computes the full square, and needs two more kernels for q and c'c.  Tiny shapes
// are the small-plan interpreter's (row-order sums, gram.hip) and keep the stream-K node as their stand-alone form.
bool gram_tiny(int64_t rows, int64_t cols) {           // (what one interpreter node may cost: SMALL_NODE_WORK_MAX, common.h)
    // (the interpreter's node walks the rows in order, one thread per (j, k): its time grows with the ROW count — 300 x 8 took 80 us where
    // the stream form's two launches take 14; up to 64 rows it stays below them)
    const int64_t t = rows * cols * (cols + 1) / 2;
    return cols > 0 && rows <= 64 && t <= 16384 && t + 64 * rows <= SMALL_NODE_WORK_MAX;
}
bool gram_tall_applies(int64_t rows, int64_t cols) { return cols >= 1 && cols <= TCOLS && rows >= 1 && !gram_tiny(rows, cols); }
// WIDE shapes (129 .. 2048 columns, any number of rows): the diagonal tiles take this kernel — one launch over (row groups x tiles), which
// also yields q for every column and c'c — and the strictly upper tiles the stream-K kernel in ONE ranged launch (gram.hip).  The stream-K
// form alone computes every diagonal tile as a full square (17 % of the executed flops at 512 columns) and needs two more kernels and a
// side-stream fork for q and c'c.  Measured (profiles/r05_gram_tall.txt) against the stream-K node: better at every row count from 100 to
// 2^20 (300 x 300: 64 -> 37 us; 4096 x 512: 112 -> 69 us; 262144 x 512: 1.76 -> 1.44 ms; 65536 x 2048: 5.69 -> 5.37 ms), equal at
// 8192 x 1024 and 4096 x 2048.  Config 2 (4096 columns) keeps the plain stream-K node.
#ifndef PMT_WIDE_MAXCOLS
#define PMT_WIDE_MAXCOLS (16 * TCOLS)
#endif
bool gram_tall_diag_applies(int64_t rows, int64_t cols) {
#ifdef PMT_TUNING
    static const int64_t minrows = [] { const char *e = getenv("PMT_TALL_DIAG_MINROWS"); return e ? atoll(e) : 1LL; }();
    static const int64_t ratio = [] { const char *e = getenv("PMT_TALL_DIAG_RATIO"); return e ? atoll(e) : 0LL; }();
    static const int64_t maxcols = [] { const char *e = getenv("PMT_TALL_DIAG_MAXCOLS"); return e ? atoll(e) : 16LL * TCOLS; }();
    return cols > TCOLS && cols <= maxcols && rows >= minrows && rows >= ratio * cols;
#else
    return cols > TCOLS && cols <= PMT_WIDE_MAXCOLS && rows >= 1;
#endif
}

// stages per workgroup: at most TALL_MAX_G workgroups over all tiles, at least TALL_MIN_CHUNK rows each
static int64_t tall_chunk(int64_t rows, int64_t cols) {
    const int64_t nst = cdiv(rows, TBK), nt = std::max<int64_t>(1, cdiv(cols, TCOLS));
    const int64_t maxg = nt == 1 ? TALL_MAX_G : std::max<int64_t>(64, 2 * TALL_MAX_G / nt);
    // (up to 2048 rows of one tile a workgroup takes ONE stage: the launch is latency-bound, a second stage per workgroup costs 2 us —
    // 500 x 100: 15 -> 11.5 us; beyond, and for several tiles, more groups only make the fix-up longer: 4096 x 512 65 -> 70 us)
    const int64_t minst = (nt == 1 && rows <= 2048) ? 1 : TALL_MIN_CHUNK / TBK;
    return std::max<int64_t>(minst, cdiv(nst, maxg));
}
int gram_tall_stage_rows(int64_t rows, int64_t cols) {
    const int nb = narrow_nb(rows, cols);
    return nb ? (stream_form(nb, rows) ? stream_iteration_rows(nb) : narrow_stage_rows(nb)) : TBK;
}
// row-pair lanes per column run (8 or 16: the panel kernels); 4: the stream form (four contraction slots per column, iterations dealt to WAVES)
int gram_tall_run_lanes(int64_t rows, int64_t cols) { const int nb = narrow_nb(rows, cols); return nb && stream_form(nb, rows) ? 4 : nb == 1 ? Narrow<1>::LPC : 8; }
int gram_tall_groups(int64_t rows, int64_t cols) {
    if (const int nb = narrow_nb(rows, cols)) return narrow_groups(rows, nb);
    return (int)cdiv(cdiv(rows, TBK), tall_chunk(rows, cols));
}
size_t gram_tall_workspace_bytes(int64_t rows, int64_t cols) {
    if (!gram_tall_applies(rows, cols) && !gram_tall_diag_applies(rows, cols)) return 0;
    if (const int nb = narrow_nb(rows, cols)) return sizeof(double) * (size_t)narrow_groups(rows, nb) * (size_t)narrow_stride(nb);
    return sizeof(double) * (size_t)gram_tall_groups(rows, cols) * (size_t)cdiv(cols, TCOLS) * TSTRIDE;
}

// all diagonal tiles of A'A (one tile when cols <= 128), q = 2 A'c for every column and c'c
int launch_gram_tall(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign, int moi,
                     const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                     double *out_const, void *workspace, hipStream_t s) {
    if (!workspace) return fail(PMT_INVALID_ARGUMENT, "quad_gram: workspace required");
    TallArgs g;
    g.A = A; g.lda = lda; g.rows = rows; g.cols = cols; g.b = (b && sign) ? b : nullptr; g.sign = g.b ? sign : 0;
    g.chunk = tall_chunk(rows, cols);
    g.nstages = cdiv(rows, TBK);
#ifdef PMT_TUNING
    static const int inter = [] { const char *e = getenv("PMT_TALL_INTERLEAVE"); return e ? atoi(e) : 1; }();
    g.interleave = inter;
#else
    g.interleave = 1;
#endif
    g.ws = reinterpret_cast<double *>(workspace);
    g.vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
    TallFixArgs f;
    f.ws = g.ws; f.cols = cols; f.xvar = xvar; f.varmap = varmap; f.moi = moi; f.out_quad = out_quad; f.out_csc = out_csc; f.alpha = alpha;
    f.out_lin = out_lin; f.out_const = out_const;
    if (const int nb = narrow_nb(rows, cols)) {
        const bool b_aligned = (reinterpret_cast<uintptr_t>(g.b) & 15) == 0;
        return nb == 1 ? launch_gram_narrow<1>(g, f, b_aligned, s) : nb == 2 ? launch_gram_narrow<2>(g, f, b_aligned, s) :
               nb == 3 ? launch_gram_stream3(g, f, b_aligned, s) : launch_gram_narrow<4>(g, f, b_aligned, s);
    }
    const int G = gram_tall_groups(rows, cols);
    const unsigned nt = (unsigned)cdiv(cols, TCOLS);
    // whole stages, aligned pieces (of A and of b): no bounds checks (columns beyond the matrix are clamped to its last one: tall_load)
    const bool fast = g.vec_in && rows % TBK == 0 && (reinterpret_cast<uintptr_t>(g.b) & 15) == 0;
    // one tile of 65 .. 112 columns: the kernel built for 5 / 6 / 7 block columns (15 / 21 / 28 blocks instead of 36)
    const int nbc = nt == 1 ? (int)std::max<int64_t>(5, cdiv(cols, 16)) : 8;
#ifdef PMT_TALL_NBC8
    const int use = 8;
#else
    const int use = nbc;
#endif
#define TALL_LAUNCH(N)                                                                                                          \
    do {                                                                                                                        \
        if (fast) PMT_LAUNCH_NAMED("gram_tall_kernel", (gram_tall_kernel<N, true>), dim3((unsigned)G, nt), dim3(256), 0, s, g);   \
        else PMT_LAUNCH_NAMED("gram_tall_kernel", (gram_tall_kernel<N, false>), dim3((unsigned)G, nt), dim3(256), 0, s, g);      \
    } while (0)
    if (use == 5) TALL_LAUNCH(5); else if (use == 6) TALL_LAUNCH(6); else if (use == 7) TALL_LAUNCH(7); else TALL_LAUNCH(8);
#undef TALL_LAUNCH
    if (int rc = check_launch("gram_tall_kernel")) return rc;
    f.G = G; f.nb = 0; f.nbc = use; f.part = tall_part(use); f.pcols = TCOLS; f.stride = tall_stride(use);
    PMT_LAUNCH(gram_tall_fixup_kernel, dim3((unsigned)cdiv(tall_part(use) + TCOLS + 1, 64), nt), dim3(1024), 0, s, f);
    return check_launch("gram_tall_fixup_kernel");
}

}  // namespace pmt
