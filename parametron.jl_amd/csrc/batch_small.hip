// Batched small instances (BASELINE config 4: n = r = 128, m = 16): ONE persistent workgroup per CU; the whole coefficient slab of an
// instance — Q = 2 A'A (upper triangle), q = 2 A'c, c'c, the constraint block C (row-major) and 0 (+|-) d — from a single pass over
// its A, written out as ONE contiguous stream.  In the reference a batch is many independent Models (src/model.jl:1-22); what is
// replaced per instance is _vecdot!/muladd! (src/functions.jl:702-709,548-576) + canonicalize! (:381-386) + the MOI copies
// (src/moi_interop.jl:45-81), coefficients only (the index arrays are identical for every instance, batch.hip).
//
// Both bounds of the step are ~0.3 ms for 8192 instances (1.9 GB of HBM traffic; 37.7 M v_mfma_f64_4x4x4_4b at 16 cycles on 1024 SIMDs),
// so the matrix pipe, the loads and the stores must all run at the same time.  The 8 waves of the workgroup are SPECIALISED (w and w + 4
// share a SIMD):
//   * waves 0-3 ("matrix waves", one per SIMD) do nothing but LDS operand reads and MFMAs.  Only the 36 of the 64 16x16 sub-tiles that
//     touch the upper triangle are computed; wave w owns the column strips 7 - w (sub-tiles tm = 0..7-w) and w (tm = 0..w): 9 sub-tiles
//     per wave, and because both strips start at tm = 0 the A operands are shared — 8 - w + 8 LDS reads per 36 MFMAs (round 1: 13 per 20).
//   * waves 4, 5 ("loader waves") stream A in 32-row chunks through TWO LDS panels: in phase g the matrix waves multiply chunk g out of
//     panel g & 1 while the loaders issue the global loads of chunk g + 2 and write chunk g + 1 (loaded during phase g - 1) into the
//     other panel — one barrier per phase, a whole phase of flight time per load, chunks run on across instance boundaries.  They also
//     accumulate q from the LDS chunk with their vector ALU (round 1's summation order).
//   * waves 6, 7 ("storer waves") write results: the finished slab of an instance is assembled in LDS in its final memory order (Q packed
//     row-major upper triangle | q | c'c | C row-major | d-constants: 83.6 KB) and copied to HBM as 16-byte stores during the NEXT instance's
//     phases, so the write-out never stops the matrix pipe (round 1 wrote 8 bytes per lane, row by row, with the pipe idle).  They also
//     transpose the constraint block and compute c'c — a serial left-to-right chain per instance (src/functions.jl:574) — for 64 instances
//     at a time, one per lane, while the first chunk is in flight (round 1: a second kernel, 19 us).
// Why three kinds of waves (measured, profiles/r02_batch_small.txt): the vector-memory path of a CU accepts ~10 B/cycle when the whole chip
// streams, so issuing a chunk's loads takes ~3400 cycles and a 2 KB store ~400 — the issuing wave just sits there.  With every wave doing
// loads, stores, address arithmetic and MFMAs in turn the matrix pipe was busy 37 % (round 1) / 50 % (pipelined, unspecialised) of the
// kernel; loads and stores in the SAME wave also wait for each other (one in-order counter).  Separated, each stream has the whole phase.
// The kernel runs at the chip's power limit: 0.39 ms without the HBM reads at 2.39 GHz, 0.46 ms with them at 1.97 GHz, the same ~1.0 M
// cycles either way (profiles/r02_batch_small.txt).
// Q uses the MFMA lane mapping and k order of gram_sk.hip: bit-identical to pmt_quad_gram_f64 on the same instance.
#include <type_traits>

#include "common.h"

#ifndef PMT_BS_SKIP
#define PMT_BS_SKIP 0      // profiling builds only: 1 = no contraction, 2 = no q, 8 = no chunk loads, 32 = no operand reads
#endif
#ifndef PMT_BS_PITCH
#define PMT_BS_PITCH 34    // LDS pitch of a panel column in doubles: 2 (mod 32) is conflict-free for ds_read_b64 (an odd pitch for ds_read2_b64)
#endif
#ifndef PMT_BS_PRIO
#define PMT_BS_PRIO 0      // s_setprio of the matrix waves (helpers stay at 0): 0, 2, 3 measured equal
#endif
#ifndef PMT_BS_FENCE
#define PMT_BS_FENCE 1     // operand reads are hoisted at most one k-step ahead of their MFMAs (unfenced the scheduler hoists several k-steps and spills)
#endif

#ifndef PMT_BS_PIN
#define PMT_BS_PIN 0       // (measured round 6c: 0.450 against 0.416 ms per step — SLOWER here, profiles/r06_batch_small_pin.txt; not shipped) 1: the matrix waves' k-step as one pinned instruction stream: one operand read behind every second MFMA (0: the reads of k-step ks + 1 in one run in front of the MFMAs of ks)
#endif
#ifndef PMT_BS_GLDS
#define PMT_BS_GLDS 0      // 1: the FAST path's A chunks go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write)
#endif

#ifndef PMT_BS_DIAG3
#define PMT_BS_DIAG3 1    // the third rotation of diagonal sub-tiles is not computed (see matrix_wave)
#endif
#ifndef PMT_BS_NT
#define PMT_BS_NT 3        // bit 0 = the A stream is loaded with the nt policy, bit 1 = the slab copy-out stores are nontemporal
#endif
// (measured, profiles/r04_batch_small.txt: A is read once and the slab written once — with both marked nontemporal the step takes 0.430 instead
// of 0.443 ms; either alone is within noise)

#ifndef PMT_BS_TRACE
#define PMT_BS_TRACE 0     // tuning builds: s_memtime stamps of workgroup 0's matrix wave 0, read back with pmt_debug_bs_trace (the loader waves must not
                           // be instrumented: a stamp is a global store, and their hand-counted s_waitcnt assumes they issue loads only)
#endif

namespace pmt {

#if PMT_BS_TRACE
__device__ long long g_bs_trace[2][8192];
#define BS_STAMP(who, idx) do { if (blockIdx.x == 0 && (idx) < 8192) g_bs_trace[who][idx] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BS_STAMP(who, idx) do { } while (0)
#endif

namespace {

constexpr int SN = 128;                 // columns handled (smaller instances are zero padded)
constexpr int CK = 32;                  // rows per chunk
constexpr int SGP = PMT_BS_PITCH;
constexpr int NT = 512;                 // threads per workgroup: waves 0-3 matrix, waves 4-7 helpers
constexpr int NH = 256;                 // first helper thread
constexpr int NL = 128, NS = 128;       // loader / storer threads
constexpr int PANEL = SN * SGP;         // doubles per panel
constexpr int STAGE_CAP = 10464;        // slab staging capacity in doubles (config 4: 10449 + 1 alignment shift)
constexpr int NPL = 16;                 // 16-byte pieces per loader thread per chunk
constexpr int CREG = 16;                // constraint-block entries prefetched per storer thread (m*n <= 2048)

typedef double f64x2 __attribute__((ext_vector_type(2)));

// LDS-DMA panel layout (GL): a global_load_lds_dwordx4 writes 64 lanes x 16 bytes LINEARLY from a wave-uniform base, so the image cannot be
// padded per column.  One instruction fills a GROUP of four columns (4 x 256 bytes; lane L = column L >> 4, 16-byte slot L & 15); groups
// are 1088 bytes apart (64 bytes of padding: the panel is exactly as large as the padded one) and inside a group slot s of column cg holds
// the row pair s ^ cg (the lane reads that pair from HBM: the swizzle is on the SOURCE address, a quarter wave still reads one whole 256-byte
// column piece).  16 lanes reading the same row pair of 16 consecutive columns then touch 16 distinct 16-byte slots: (group & 3) * 64 +
// ((pair ^ cg) & 3) * 16 + ... covers every bank once — conflict-free like the 34-double pitch, for 8- and 16-byte reads alike.
constexpr int GLG = 136;                // doubles per column group (1088 bytes)
// doubles offset of (column c, row r) in a GL panel
__device__ __forceinline__ int gl_off(int c, int r) { return (c >> 2) * GLG + (c & 3) * 32 + ((((r >> 1) ^ (c & 3)) << 1) | (r & 1)); }

struct SmallArgs {
    const double *A; int64_t lda, rows, cols, strideA;
    const double *b; int64_t strideb; int sign;
    double *out; int64_t out_stride;    // slab of instance i at out + i*out_stride: [Q | q | const | C | d]
    int64_t B;
    const double *Cm; int64_t m; const double *d; int sign_d;   // optional constraint block (m x cols, column-major)
    int stage_all;                      // the whole slab fits the LDS staging buffer (else C and d are written directly)
};

struct Shared {
    double *panel, *stage, *cvec;       // panel[2][PANEL], stage[STAGE_CAP + 2], cvec[2][CK]
};

__device__ __forceinline__ void phase_barrier() { __syncthreads(); }

// ---- matrix wave W (0..3): column strips 7 - W (tm = 0 .. 7 - W) and W (tm = 0 .. W)
// ALLCOLS: n == 128 (no column predicate on the staging stores)
template <int W, bool ALLCOLS, bool GL>
__device__ __forceinline__ void matrix_wave(const SmallArgs &p, const Shared &sh, int lane, int n, int nchunk) {
    constexpr int KA = 8 - W, KB = W + 1, TNA = 7 - W, TNB = W;
    const int lm = lane & 15, lk = lane >> 4;
    double accA[KA][4], accB[KB][4];
#pragma unroll
    for (int i = 0; i < KA; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) accA[i][r] = 0.0;
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) accB[i][r] = 0.0;
    int rc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rc[r] = (((((lm >> 2) + r) & 3) << 2) | (lm & 3)) * SGP;     // column group rotated by r blocks
    // staging positions of this lane's accumulator elements: element (tile tm = i, r) is C[row_i][colX_r] with row_i = 16 i + 4 bq + i_,
    // colX_r = 16 TNX + 4 ((bq + r) & 3) + j_  ->  packed upper-triangular position row n - row (row - 1) / 2 + (col - row); the same for
    // every instance, so they are computed once (rb[i] + cX[r]); only the two diagonal sub-tiles (and columns >= n) need a predicate
    const int i_ = lane >> 4, bq = (lane >> 2) & 3, j_ = lane & 3;
    int rb[KA], cA[4], cB[4];
#pragma unroll
    for (int i = 0; i < KA; ++i) { const int row = i * 16 + 4 * bq + i_; rb[i] = row * n - (row * (row - 1)) / 2 - row; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { cA[r] = TNA * 16 + 4 * ((bq + r) & 3) + j_; cB[r] = TNB * 16 + 4 * ((bq + r) & 3) + j_; }
    const int rowdA = (KA - 1) * 16 + 4 * bq + i_, rowdB = (KB - 1) * 16 + 4 * bq + i_;      // rows of the diagonal sub-tiles (tm == tn)
    if (PMT_BS_PRIO) __builtin_amdgcn_s_setprio(PMT_BS_PRIO);                                                           // the matrix pipe's wave outranks the helper on its SIMD
    const int a0 = lm * SGP + lk;                                                            // sub-tile tm: + tm * 16 * SGP
    const int bA0 = (TNA * 16) * SGP + lk, bB0 = (TNB * 16) * SGP + lk;
    // GL panels: (column 16 t + c', row 4 ks + lk) sits at gl_off; the row pair 2 ks + (lk >> 1) is XORed with the column's cg = c' & 3, whose
    // bit 1 meets the parity of ks: one base for even and one for odd k-steps, everything else is an immediate (t * 544, (ks >> 1) * 8 doubles)
    const int cgl = lm & 3;
    const int glane = cgl * 32 + ((((lk >> 1) ^ (cgl & 1)) << 1) | (lk & 1));
    const int gaE = (lm >> 2) * GLG + glane + (cgl >> 1) * 4, gaO = (lm >> 2) * GLG + glane + (1 - (cgl >> 1)) * 4;
    int gbE[4], gbO[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int grp = ((lm >> 2) + r) & 3;
        gbE[r] = grp * GLG + glane + (cgl >> 1) * 4;
        gbO[r] = grp * GLG + glane + (1 - (cgl >> 1)) * 4;
    }
    const int64_t G = gridDim.x;
    int cur = 0;
    int tphase = 0; (void)tphase;
    phase_barrier();                                          // panel 0 holds the first chunk
    for (int64_t inst = blockIdx.x; inst < p.B; inst += G) {
        for (int ch = 0; ch < nchunk; ++ch) {
            const double *pan = sh.panel + cur * PANEL;
            if (PMT_BS_TRACE && W == 0 && lane == 0) BS_STAMP(0, 4 * tphase + 0);
            if (!(PMT_BS_SKIP & 1)) {
                // explicit software pipeline: the operands of k-step ks + 1 are read (into the other register set) BEFORE the MFMAs of
                // k-step ks are issued, so the LDS latency hides behind 36 MFMAs instead of idling this SIMD's matrix pipe (there is no second
                // matrix wave on the SIMD to cover it).  The fences keep the reads from being hoisted further (register pressure).
                double a[2][KA], bA[2][4], bB[2][4];
                auto read_operands = [&](int s_, int ks) {
                    if (PMT_BS_SKIP & 32) {
#pragma unroll
                        for (int i = 0; i < KA; ++i) a[s_][i] = (double)(lane + i + ks);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { bA[s_][r] = (double)(lane + r); bB[s_][r] = (double)(lane - r); }
                        return;
                    }
                    if (GL) {
                        const int hi = (ks >> 1) * 8;
#pragma unroll
                        for (int i = 0; i < KA; ++i) a[s_][i] = pan[((ks & 1) ? gaO : gaE) + i * 4 * GLG + hi];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int b0 = (ks & 1) ? gbO[r] : gbE[r];
                            bA[s_][r] = pan[b0 + TNA * 4 * GLG + hi];
                            bB[s_][r] = pan[b0 + TNB * 4 * GLG + hi];
                        }
                        return;
                    }
#pragma unroll
                    for (int i = 0; i < KA; ++i) a[s_][i] = pan[a0 + i * 16 * SGP + ks * 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bA[s_][r] = pan[bA0 + rc[r] + ks * 4]; bB[s_][r] = pan[bB0 + rc[r] + ks * 4]; }
                };
                read_operands(0, 0);
#if PMT_BS_PIN
                // Round 6c: the k-step as ONE pinned instruction stream (gram_mid.hip: mid_step).  The matrix wave is alone with its SIMD's
                // matrix pipe: the KA + 8 operand reads of k-step ks + 1 stood in one run in front of the 34-36 MFMAs of k-step ks, and a
                // run of LDS instructions is issue time the pipe waits for.  Here ONE read stands behind every second MFMA, in the order
                // the next k-step needs them (the A operands, then the B operands rotation by rotation); same MFMAs in the same order.
                auto read_one = [&](int s_, int ks, int idx) {
                    if (PMT_BS_SKIP & 32) { read_operands(s_, ks); return; }
                    const int hi = (ks >> 1) * 8;
                    if (idx < KA) {
                        a[s_][idx] = GL ? pan[((ks & 1) ? gaO : gaE) + idx * 4 * GLG + hi] : pan[a0 + idx * 16 * SGP + ks * 4];
                    } else {
                        const int r = (idx - KA) >> 1;
                        const bool isB = ((idx - KA) & 1) != 0;
                        if (GL) {
                            const int b0 = (ks & 1) ? gbO[r] : gbE[r];
                            if (isB) bB[s_][r] = pan[b0 + TNB * 4 * GLG + hi]; else bA[s_][r] = pan[b0 + TNA * 4 * GLG + hi];
                        } else {
                            if (isB) bB[s_][r] = pan[bB0 + rc[r] + ks * 4]; else bA[s_][r] = pan[bA0 + rc[r] + ks * 4];
                        }
                    }
                };
#pragma unroll
                for (int ks = 0; ks < CK / 4; ++ks) {
                    const int s_ = ks & 1;
                    const bool more = ks + 1 < CK / 4;
                    int m = 0, nread = 0;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int i = 0; i < KA + KB; ++i) {
                            const bool isA = i < KA;
                            const int ii = isA ? i : i - KA;
                            if (PMT_BS_DIAG3 && r == 3 && ii == (isA ? KA : KB) - 1) continue;
                            if (isA) accA[ii][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s_][ii], bA[s_][r], accA[ii][r], 0, 0, 0);
                            else accB[ii][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s_][ii], bB[s_][r], accB[ii][r], 0, 0, 0);
                            if (more && (m & 1) == 0 && nread < KA + 8 && !(PMT_BS_SKIP & 32)) {
                                __builtin_amdgcn_sched_barrier(0);
                                read_one(s_ ^ 1, ks + 1, nread);
                                __builtin_amdgcn_sched_barrier(0);
                                ++nread;
                            }
                            ++m;
                        }
                    }
                    if (more && (PMT_BS_SKIP & 32)) read_operands(s_ ^ 1, ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
#pragma unroll
                for (int ks = 0; ks < CK / 4; ++ks) {
                    const int s_ = ks & 1;
                    if (ks + 1 < CK / 4) read_operands(s_ ^ 1, ks + 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (the third rotation of a DIAGONAL sub-tile — i = KA - 1 / KB - 1 — holds the transposes of the first one's 4 x 4 blocks:
                        // not computed, the write-out below takes block (3, 0) of rotation 1 as (0, 3); gram_tall.hip tall_diag_rule)
#pragma unroll
                        for (int i = 0; i < KA; ++i)
                            if (!(PMT_BS_DIAG3 && r == 3 && i == KA - 1))
                                accA[i][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s_][i], bA[s_][r], accA[i][r], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < KB; ++i)
                            if (!(PMT_BS_DIAG3 && r == 3 && i == KB - 1))
                                accB[i][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s_][i], bB[s_][r], accB[i][r], 0, 0, 0);
                    }
                    if (PMT_BS_FENCE) asm volatile("" ::: "memory");
                }
#endif
            }
            if (PMT_BS_TRACE && W == 0 && lane == 0) BS_STAMP(0, 4 * tphase + 1);
            if (ch == nchunk - 1) {
                phase_barrier();                              // the storer waves have copied the previous slab out of the staging buffer
                // ---- the Q coefficients of this instance, x2, at their packed row-major upper-triangular positions in the staging buffer
                double *outp = p.out + inst * p.out_stride;
                double *st = sh.stage + (int)((reinterpret_cast<uintptr_t>(outp) >> 3) & 1);
#pragma unroll
                for (int i = 0; i < KA; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (PMT_BS_DIAG3 && i == KA - 1) {
                            // diagonal sub-tile: rotation 0 keeps row <= col, 2 keeps row < col, 1 keeps everything — its block (3, 0) at the
                            // transposed position —, 3 was not computed
                            const int row = rowdA, col = cA[r];
                            if (r == 1 && row > col) { if (ALLCOLS || row < n) st[(col * n - (col * (col - 1)) / 2 - col) + row] = 2 * accA[i][r]; }
                            else if (r != 3 && row <= col && (ALLCOLS || col < n)) st[rb[i] + col] = 2 * accA[i][r];
                        } else {
                            const bool ok = (i < KA - 1 || rowdA <= cA[r]) && (ALLCOLS || cA[r] < n);
                            if (ok) st[rb[i] + cA[r]] = 2 * accA[i][r];
                        }
                        accA[i][r] = 0.0;
                    }
                    if (i < KB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (PMT_BS_DIAG3 && i == KB - 1) {
                                const int row = rowdB, col = cB[r];
                                if (r == 1 && row > col) { if (ALLCOLS || row < n) st[(col * n - (col * (col - 1)) / 2 - col) + row] = 2 * accB[i][r]; }
                                else if (r != 3 && row <= col && (ALLCOLS || col < n)) st[rb[i] + col] = 2 * accB[i][r];
                            } else {
                                const bool ok = (i < KB - 1 || rowdB <= cB[r]) && (ALLCOLS || cB[r] < n);
                                if (ok) st[rb[i] + cB[r]] = 2 * accB[i][r];
                            }
                            accB[i][r] = 0.0;
                        }
                    }
                }
            }
            if (PMT_BS_TRACE && W == 0 && lane == 0) BS_STAMP(0, 4 * tphase + 2);
            phase_barrier();
            if (PMT_BS_TRACE && W == 0 && lane == 0) { BS_STAMP(0, 4 * tphase + 3); }
            ++tphase;
            cur ^= 1;
        }
    }
}

// ---- loader waves 4, 5 (lt = 0..127): chunk loads -> LDS panels, and q from the LDS chunk
// ---- loader waves 4, 5, LDS-DMA form (FAST shapes only): in phase g the 16 global_load_lds_dwordx4 of chunk g + 1 go straight into the panel
// chunk g - 1 has just left (no staging registers, no ds_write pass), q is accumulated from panel g while they fly, and the wave waits for
// them (vmcnt(0), by hand: the compiler does not count asm loads) just before the phase's barrier — the matrix waves read the panel one
// phase after that wait.
__device__ __forceinline__ void loader_waves_glds(const SmallArgs &p, const Shared &sh, int lt, int n, int nchunk) {
    const int lane = lt & 63, lw = __builtin_amdgcn_readfirstlane(lt >> 6);
    const int lm = lane & 15;
    const int nq = n * (n + 1) / 2;
    const int64_t G = gridDim.x;
    const bool has_c = p.b && p.sign;
    const double *csrc = has_c ? p.b : p.A;
    const int64_t cstride = has_c ? p.strideb : p.strideA;
    const int csign = has_c ? p.sign : 0;
    // instruction q of this wave fills column group 16 lw + q; lane L of it: column 4 (16 lw + q) + (L >> 4), row pair (L & 15) ^ (L >> 4)
    const int64_t lane_src = (int64_t)(lane >> 4) * p.lda + 2 * ((lane & 15) ^ (lane >> 4));
    const unsigned pan_lds = (unsigned)reinterpret_cast<uintptr_t>(sh.panel) + (unsigned)(lw * 16 * GLG * 8);       // LDS byte address (low half of the generic one)
    double cval = 0.0;
    auto issue_chunk = [&](int64_t inst, int ch, int buf) {
        const double *src0 = p.A + inst * p.strideA + (int64_t)(lw * 64) * p.lda + (int64_t)ch * CK + lane_src;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(pan_lds + (unsigned)(buf * PANEL * 8));
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const double *src = src0 + (int64_t)(4 * q) * p.lda;
            const unsigned dst = dst0 + (unsigned)(q * GLG * 8);
            unsigned keep;
            if (PMT_BS_NT & 1)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
        const double *cs = csrc + inst * cstride + (int64_t)ch * CK + (lt & (CK - 1));
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(cval) : "v"(cs) : "memory");
    };
    auto land_chunk = [&](int buf) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cval) : : "memory");
        sh.cvec[buf * CK + (lt & (CK - 1))] = signed_const(cval, csign);                      // four threads write the same value: benign
    };
    // q as in the register form; (column c, rows 4 u + 2 lkp, + 1) is one aligned 16-byte read at gl_off(c, 4 u + 2 lkp): the pair index
    // 2 u + lkp meets cg = c & 3, whose bit 1 meets the parity of u
    double qpart[4] = {0.0, 0.0, 0.0, 0.0};
    const int lkp = (lane >> 4) & 1, cg = lane >> 5;
    const int c0 = 64 * lw + 16 * cg + lm, cgl = c0 & 3;
    const int qbase = (c0 >> 2) * GLG + cgl * 32 + ((lkp ^ (cgl & 1)) << 1);
    const int qE = qbase + (cgl >> 1) * 4, qO = qbase + (1 - (cgl >> 1)) * 4;
    auto next_of = [&](int64_t i, int c, int64_t &ni, int &nc) { nc = c + 1; ni = i; if (nc == nchunk) { nc = 0; ni = i + G; } };

    int64_t inst = blockIdx.x;
    int ch = 0;
    if (inst < p.B) {
        issue_chunk(inst, 0, 0);
        land_chunk(0);
    }
    phase_barrier();
    int par = 0;
    while (inst < p.B) {
        const double *pan = sh.panel + par * PANEL;
        int64_t n1i; int n1c;
        next_of(inst, ch, n1i, n1c);
        const bool more = n1i < p.B;
        const bool last = (ch == nchunk - 1);
        issue_chunk(more ? n1i : inst, more ? n1c : ch, par ^ 1);          // (past the end: a valid chunk again, never used)
        if (!(PMT_BS_SKIP & 2)) {
#pragma unroll
            for (int u = 0; u < CK / 4; ++u) {
                const f64x2 c2 = *reinterpret_cast<const f64x2 *>(sh.cvec + par * CK + 4 * u + 2 * lkp);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f64x2 a2 = *reinterpret_cast<const f64x2 *>(pan + ((u & 1) ? qO : qE) + j * 8 * GLG + (u >> 1) * 8);
                    const double p0 = c2.x * a2.x, p1 = c2.y * a2.y;
                    qpart[2 * j] = qpart[2 * j] + p0;
                    qpart[2 * j + 1] = qpart[2 * j + 1] + p1;
                }
            }
        }
        if (last) {
            phase_barrier();                                  // the previous slab has left the staging buffer
            double *outp = p.out + inst * p.out_stride;
            double *st = sh.stage + (int)((reinterpret_cast<uintptr_t>(outp) >> 3) & 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double p2 = __shfl(qpart[2 * j], lane + 16, 64), p3 = __shfl(qpart[2 * j + 1], lane + 16, 64);
                const int col = 64 * lw + 32 * j + 16 * cg + lm;
                if (lkp == 0 && col < n) st[nq + col] = 2 * (((qpart[2 * j] + qpart[2 * j + 1]) + p2) + p3);
                qpart[2 * j] = 0.0;
                qpart[2 * j + 1] = 0.0;
            }
        }
        land_chunk(par ^ 1);
        phase_barrier();
        inst = n1i; ch = n1c;
        par ^= 1;
    }
}

template <bool FAST>
__device__ __forceinline__ void loader_waves(const SmallArgs &p, const Shared &sh, int lt, int n, int nchunk) {
    const int lane = lt & 63, lw = lt >> 6;
    const int lm = lane & 15;
    const int kp = lt & 15, cc0 = lt >> 4;                    // chunk loads: 16-byte piece kp of columns cc0 + 8 q
    const int nq = n * (n + 1) / 2;
    const int64_t G = gridDim.x;
    const bool has_c = p.b && p.sign;
    const double *csrc = has_c ? p.b : p.A;
    const int64_t cstride = has_c ? p.strideb : p.strideA;
    const int csign = has_c ? p.sign : 0;

    // chunk (inst, ch): global -> registers (set S); registers -> LDS panel `buf`.
    // Two register sets, parity of the phase: in phase g the loads of chunk g + 2 are issued FIRST (into the set chunk g has just left),
    // then chunk g + 1 — loaded during phase g - 1, a whole phase ago — goes from the other set into LDS.  Issuing a chunk's 17 loads takes
    // ~3400 cycles (the CU's vector-memory path accepts ~10 B/cycle when the chip streams), so they must not sit behind a wait.
    // FAST: the loads are issued through inline asm and waited for by hand (wait_chunk): the compiler's waitcnt pass answers loop-carried
    // loads with s_waitcnt vmcnt(0), which here would wait for the loads issued a moment ago.  These waves issue no other vector-memory
    // operation, loads return in order, so "at most NLOADS outstanding" means exactly: the older set has landed.
    constexpr int NLOADS = NPL + 1;
    f64x2 R[2][NPL] = {};
    double cval[2] = {0.0, 0.0};
    auto load_chunk = [&](auto set_t, int64_t inst, int ch) {
        constexpr int S = decltype(set_t)::value;
        const double *A = p.A + inst * p.strideA;
        const int64_t row = (int64_t)ch * CK + 2 * kp;
        if (PMT_BS_SKIP & 8) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) { R[S][q].x = 1.0; R[S][q].y = 2.0; }
            cval[S] = 0.5;
        } else if (FAST) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const double *src = A + (int64_t)(cc0 + 8 * q) * p.lda + row;
                f64x2 &dst = R[S][q];
                if (PMT_BS_NT & 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(src) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
            }
            // every thread reads c of row lt & 31 (no branch); without a b (sign 0) the read goes to A and signed_const(., 0) = 0
            const double *src = csrc + inst * cstride + (int64_t)ch * CK + (lt & (CK - 1));
            double &dst = cval[S];
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
        } else {
            const int64_t rmax = max(p.rows - 1, (int64_t)0), cmax = max(p.cols - 1, (int64_t)0);
            const int64_t r0 = min(row, rmax), r1 = min(row + 1, rmax);
            const bool ok0 = row < p.rows, ok1 = row + 1 < p.rows;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int col = cc0 + 8 * q;
                const double *src = A + min((int64_t)col, cmax) * p.lda;
                const bool okc = col < p.cols && p.rows > 0;
                const double x = okc ? src[r0] : 0.0, y = okc ? src[r1] : 0.0;
                R[S][q].x = (okc && ok0) ? x : 0.0;
                R[S][q].y = (okc && ok1) ? y : 0.0;
            }
            cval[S] = 0.0;
            if (lt < CK) {
                const int64_t rr = (int64_t)ch * CK + lt;
                if (has_c && rr < p.rows) cval[S] = p.b[inst * p.strideb + rr];
            }
        }
    };
    // wait until register set S has landed; `younger` = number of loads issued after it that may stay in flight (0 or NLOADS)
    auto wait_chunk = [&](auto set_t, auto younger_t) {
        constexpr int S = decltype(set_t)::value;
        constexpr int YOUNGER = decltype(younger_t)::value;
        if (FAST && !(PMT_BS_SKIP & 8)) {
            f64x2 (&r)[NPL] = R[S];                           // (asm operands inside a generic lambda must name locals)
            double &cv = cval[S];
            asm volatile("s_waitcnt vmcnt(%17)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                         "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(cv) : "n"(YOUNGER) : "memory");
        }
    };
    auto store_chunk = [&](auto set_t, int buf) {
        constexpr int S = decltype(set_t)::value;
        double *pan = sh.panel + buf * PANEL;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            double *d = pan + (cc0 + 8 * q) * SGP + 2 * kp;
            if (SGP % 2 == 0) *reinterpret_cast<f64x2 *>(d) = R[S][q];
            else { d[0] = R[S][q].x; d[1] = R[S][q].y; }
        }
        if (FAST) sh.cvec[buf * CK + (lt & (CK - 1))] = signed_const(cval[S], csign);         // four threads write the same value: benign
        else if (lt < CK) sh.cvec[buf * CK + lt] = signed_const(cval[S], csign);
    };

    // q = 2 A'c: column col of a chunk adds c[row] * A[row, col] into one partial sum per row class (row mod 4), rows ascending; the four classes
    // are added in class order at the end of the instance (round 1's summation order, = pmt_quad_gram_f64's).  Lane (cg, lkp, lm) owns columns
    // 64 lw + 32 j + 16 cg + lm (j = 0, 1) and the classes 2 lkp, 2 lkp + 1: qpart[2 j + k] = class 2 lkp + k of column j.  Rows 4u + 2 lkp and
    // 4u + 2 lkp + 1 of a column are one aligned 16-byte LDS read (even pitch): 24 LDS reads per phase instead of 40, and 16 lanes x 16 bytes
    // at a pitch of 272 bytes touch every bank exactly once.
    static_assert(SGP % 2 == 0, "the q reads are 16-byte reads of row pairs");
    double qpart[4] = {0.0, 0.0, 0.0, 0.0};
    const int lkp = (lane >> 4) & 1, cg = lane >> 5;
    const int qoff = (64 * lw + 16 * cg + lm) * SGP + 2 * lkp;
    auto next_of = [&](int64_t i, int c, int64_t &ni, int &nc) { nc = c + 1; ni = i; if (nc == nchunk) { nc = 0; ni = i + G; } };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using None = std::integral_constant<int, 0>;
    using OneSet = std::integral_constant<int, NLOADS>;
    // prime the pipeline: chunk 0 -> panel 0 (through set 0), chunk 1 -> set 1.  Phase g then loads chunk g + 2 into set g & 1 and stores
    // chunk g + 1 from set (g + 1) & 1.
    int64_t inst = blockIdx.x;
    int ch = 0;
    if (inst < p.B) {
        load_chunk(P0{}, inst, 0);
        wait_chunk(P0{}, None{});
        store_chunk(P0{}, 0);
        int64_t n1i; int n1c;
        next_of(inst, 0, n1i, n1c);
        if (n1i >= p.B) { n1i = inst; n1c = 0; }
        load_chunk(P1{}, n1i, n1c);
    }
    phase_barrier();
    // phase g (parity PAR): the matrix waves multiply chunk g out of panel PAR
    auto phase = [&](auto par_t) {
        constexpr int PAR = decltype(par_t)::value;
        using Other = std::integral_constant<int, PAR ^ 1>;
        const double *pan = sh.panel + PAR * PANEL;
        int64_t n1i, n2i; int n1c, n2c;
        next_of(inst, ch, n1i, n1c);
        next_of(n1i, n1c, n2i, n2c);
        if (n2i >= p.B) { n2i = inst; n2c = ch; }             // past the end: re-load a valid chunk, never used
        const bool last = (ch == nchunk - 1);
        load_chunk(par_t, n2i, n2c);                          // chunk g + 2 -> set PAR (chunk g left it during phase g - 1)
        wait_chunk(Other{}, OneSet{});                        // chunk g + 1 (set PAR ^ 1, issued a phase ago) has landed ...
        store_chunk(Other{}, PAR ^ 1);                        // ... and goes into the other panel
        if (!(PMT_BS_SKIP & 2)) {                             // q (vector ALU): two rows of two columns per 16-byte LDS read
#pragma unroll
            for (int u = 0; u < CK / 4; ++u) {
                const f64x2 c2 = *reinterpret_cast<const f64x2 *>(sh.cvec + PAR * CK + 4 * u + 2 * lkp);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f64x2 a2 = *reinterpret_cast<const f64x2 *>(pan + qoff + j * 32 * SGP + 4 * u);
                    const double p0 = c2.x * a2.x, p1 = c2.y * a2.y;
                    qpart[2 * j] = qpart[2 * j] + p0;
                    qpart[2 * j + 1] = qpart[2 * j + 1] + p1;
                }
            }
        }
        if (last) {
            phase_barrier();                                  // the previous slab has left the staging buffer
            double *outp = p.out + inst * p.out_stride;
            double *st = sh.stage + (int)((reinterpret_cast<uintptr_t>(outp) >> 3) & 1);
            // q: the four row classes of a column, added in class order, x2 (classes 2, 3 live 16 lanes up)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double p2 = __shfl(qpart[2 * j], lane + 16, 64), p3 = __shfl(qpart[2 * j + 1], lane + 16, 64);
                const int col = 64 * lw + 32 * j + 16 * cg + lm;
                if (lkp == 0 && col < n) st[nq + col] = 2 * (((qpart[2 * j] + qpart[2 * j + 1]) + p2) + p3);
                qpart[2 * j] = 0.0;
                qpart[2 * j + 1] = 0.0;
            }
        }
        phase_barrier();
        inst = n1i; ch = n1c;
    };
    while (inst < p.B) {
        phase(P0{});
        if (inst >= p.B) break;
        phase(P1{});
    }
    // the clamped re-loads of the last two phases are still in flight INTO the register sets: nothing may reuse those registers before
    // they have landed
    wait_chunk(P0{}, None{});
    wait_chunk(P1{}, None{});
}

// ---- storer waves 6, 7 (st_ = 0..127): slab copy-out, constraint block, c'c
__device__ __forceinline__ void storer_waves(const SmallArgs &p, const Shared &sh, int st_, int n, int nchunk) {
    const int lane = st_ & 63, sw = st_ >> 6;
    const int nq = n * (n + 1) / 2;
    const int64_t G = gridDim.x;
    const int m = p.Cm ? (int)min(p.m, (int64_t)1 << 20) : 0;
    const int64_t mn = (int64_t)m * n;
    const bool creg_path = p.Cm && p.stage_all && mn <= (int64_t)CREG * NS && m <= NS;
    const int L = (p.Cm && p.stage_all) ? nq + n + 1 + (int)mn + m : nq + n + 1;         // staged (contiguous) doubles per instance
    const bool has_c = p.b && p.sign;

    // copy-out of the slab staged for the PREVIOUS instance: 16 bytes per thread per piece, stage and HBM co-aligned; pair indices past
    // the end are clamped to the last pair (rewritten with the same bytes), so a batch needs no branch.  Four pieces at a time keep four
    // LDS reads in flight (a lone read queues behind the matrix waves' operand reads).  The store ISSUE is what takes the time here — the
    // write path of a CU has a bounded number of stores in flight — which is why these waves do nothing else that the phase waits for.
    double *cp_out = nullptr; int cp_head = 0, cp_u = 0, cp_n = 0, cp_last = 0;
    auto copy_batch = [&]() {
        f64x2 v[4]; int pos[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pos[k] = cp_head + 2 * min(st_ + NS * (cp_u + k), cp_last);
            v[k] = *reinterpret_cast<const f64x2 *>(sh.stage + cp_head + pos[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (PMT_BS_NT & 2) __builtin_nontemporal_store(v[k], reinterpret_cast<f64x2 *>(cp_out + pos[k]));
            else *reinterpret_cast<f64x2 *>(cp_out + pos[k]) = v[k];
        }
        cp_u += 4;
    };
    auto copy_begin = [&](int64_t inst) {
        cp_out = p.out + inst * p.out_stride;
        cp_head = (int)((reinterpret_cast<uintptr_t>(cp_out) >> 3) & 1);
        const int npairs = (L - cp_head) / 2;                 // >= 1 (L >= 3)
        cp_last = npairs - 1;
        cp_u = 0;
        cp_n = (npairs + NS - 1) / NS;
        if (cp_head && st_ == 0) cp_out[0] = sh.stage[cp_head];                                         // unaligned first element
        if (cp_head + 2 * npairs < L && st_ == 1) cp_out[L - 1] = sh.stage[cp_head + L - 1];            // odd last element
    };

    // c'c of 64 of this workgroup's instances at a time, one instance per lane of the last wave: ((0 + c_0^2) + c_1^2) + ...
    // left to right (src/functions.jl:574), loads batched eight deep
    double cst = 0.0;
    auto const_chains = [&](int64_t li0) {
        const int64_t my = (int64_t)blockIdx.x + (li0 + lane) * G;
        double s = 0.0;
        if (my < p.B && has_c) {
            const double *bb = p.b + my * p.strideb;
            int64_t i = 0;
            for (; i + 8 <= p.rows; i += 8) {
                double v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = bb[i + t];
#pragma unroll
                for (int t = 0; t < 8; ++t) { const double c = signed_const(v[t], p.sign); const double pr = c * c; s = s + pr; }
            }
            for (; i < p.rows; ++i) { const double c = signed_const(bb[i], p.sign); const double pr = c * c; s = s + pr; }
        }
        cst = s;
    };

    double creg[CREG]; double dreg = 0.0;
#pragma unroll
    for (int u = 0; u < CREG; ++u) creg[u] = 0.0;
    int64_t li = 0;
    bool pending = false;
    if ((int64_t)blockIdx.x < p.B && sw == 1) const_chains(0);
    phase_barrier();
    for (int64_t inst = blockIdx.x; inst < p.B; inst += G) {
        for (int ch = 0; ch < nchunk; ++ch) {
            const bool last = (ch == nchunk - 1);
            if (ch == 0 && creg_path) {                       // constraint block of this instance: loaded in its first phase, staged in its last
                const double *Ci = p.Cm + inst * mn;
#pragma unroll
                for (int u = 0; u < CREG; ++u) {              // entry e of the ROW-major output: consecutive lanes -> consecutive staging addresses
                    const int e = min(st_ + NS * u, (int)mn - 1);     // (the column-major reads are strided, but hit L1/L2; entries past the end are
                    const int row = e / n, col = e - row * n;         //  clamped to the last one: written twice, same value)
                    creg[u] = Ci[col * m + row];
                }
                dreg = p.d[inst * m + min(st_, m - 1)];
            }
            if (pending) {                                    // the previous slab leaves during this instance's phases, a share per phase
                const int todo = (cp_n - cp_u + (nchunk - ch) - 1) / (nchunk - ch);
                for (int u = 0; u < todo; u += 4) copy_batch();          // (pieces past the end are clamped re-writes of the last pair)
                if (last) { while (cp_u < cp_n) copy_batch(); pending = false; }
            }
            if (last) {
                phase_barrier();                              // every wave: the previous slab has left the staging buffer, restaging may begin
                double *outp = p.out + inst * p.out_stride;
                double *st = sh.stage + (int)((reinterpret_cast<uintptr_t>(outp) >> 3) & 1);
                if (sw == 1 && lane == (int)(li & 63)) st[nq + n] = cst;
                if (p.Cm) {
                    double *sc = p.stage_all ? st + nq + n + 1 : outp + nq + n + 1;           // staged, or straight to HBM when too large
                    if (creg_path) {
#pragma unroll
                        for (int u = 0; u < CREG; ++u) sc[min(st_ + NS * u, (int)mn - 1)] = creg[u];
                        sc[mn + min(st_, m - 1)] = signed_const(dreg, p.sign_d);
                    } else {
                        const double *Ci = p.Cm + inst * mn;
                        for (int64_t e = st_; e < mn; e += NS) { const int64_t col = e / m, row = e - col * m; sc[row * n + col] = Ci[e]; }
                        for (int i = st_; i < m; i += NS) sc[mn + i] = signed_const(p.d[inst * m + i], p.sign_d);
                    }
                }
                ++li;
            }
            phase_barrier();
            if (last) {
                copy_begin(inst);
                pending = true;
                if ((li & 63) == 0 && sw == 1) const_chains(li);          // more than 64 instances per workgroup: next batch of chains
            }
        }
    }
    if (pending) { while (cp_u < cp_n) copy_batch(); }
}

}  // namespace

// FAST: cols == 128, rows a multiple of 32, 16-byte aligned columns — aligned 16-byte loads without bounds checks.
template <bool FAST, bool GL = false>
__global__ __launch_bounds__(NT, 2) void batch_small_kernel(SmallArgs p) {
    static_assert(!GL || FAST, "the LDS-DMA panels serve the FAST shapes only");
    __shared__ __attribute__((aligned(16))) double panel[2 * PANEL];
    __shared__ __attribute__((aligned(16))) double stage[STAGE_CAP + 2];
    __shared__ __attribute__((aligned(16))) double cvec[2 * CK];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int n = (int)p.cols;                                // <= 128
    const int nchunk = (int)max((int64_t)1, (p.rows + CK - 1) / CK);
    Shared sh{panel, stage, cvec};
    // every wave executes the same number of barriers: one to start, one per phase, one more in the last phase of every instance
    if (wave == 0) matrix_wave<0, FAST, GL>(p, sh, tid & 63, n, nchunk);
    else if (wave == 1) matrix_wave<1, FAST, GL>(p, sh, tid & 63, n, nchunk);
    else if (wave == 2) matrix_wave<2, FAST, GL>(p, sh, tid & 63, n, nchunk);
    else if (wave == 3) matrix_wave<3, FAST, GL>(p, sh, tid & 63, n, nchunk);
    else if (wave < 6) { if (GL) loader_waves_glds(p, sh, tid - NH, n, nchunk); else loader_waves<FAST>(p, sh, tid - NH, n, nchunk); }
    else storer_waves(p, sh, tid - NH - NL, n, nchunk);
}

static int cu_count() {
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 256; }
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        cus[dev] = v;
    }
    return cus[dev];
}

// out: slab base; the sections of an instance's slab are contiguous ([Q | q | const | C | d], batch.hip)
int launch_batch_small(const double *A, int64_t lda, int64_t rows, int64_t cols, int64_t strideA, const double *b, int64_t strideb, int sign,
                       int64_t B, double *out, int64_t out_stride, const double *Cm, int64_t m, const double *d, int sign_d, hipStream_t s) {
    SmallArgs p;
    p.Cm = (m > 0 && cols > 0) ? Cm : nullptr; p.m = p.Cm ? m : 0; p.d = d; p.sign_d = sign_d;
    p.A = A; p.lda = lda; p.rows = rows; p.cols = cols; p.strideA = strideA; p.b = b; p.strideb = strideb; p.sign = sign;
    p.out = out; p.out_stride = out_stride; p.B = B;
    const int64_t nq = cols * (cols + 1) / 2;
    p.stage_all = (nq + cols + 1 + p.m * cols + p.m + 1 <= STAGE_CAP) ? 1 : 0;
    const bool fast = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0 && (strideA & 1) == 0 && cols == SN && rows > 0 && (rows % CK) == 0;
    // one workgroup per CU (LDS-limited), each walks instances blockIdx.x, blockIdx.x + G, ...
    const dim3 grid((unsigned)std::min<int64_t>(B, cu_count()));
    if (fast) PMT_LAUNCH_NAMED("batch_small_kernel", (batch_small_kernel<true, PMT_BS_GLDS != 0>), grid, dim3(NT), 0, s, p);
    else PMT_LAUNCH_NAMED("batch_small_kernel", batch_small_kernel<false>, grid, dim3(NT), 0, s, p);
    return check_launch("batch_small_kernel");
}

#if PMT_BS_TRACE
extern "C" int pmt_debug_bs_trace(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bs_trace), sizeof(long long) * 2 * 8192) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace pmt
