// Canonical least-squares objective for WIDE shapes of 129 .. 2048 columns (from the sizes the reference is used at with a few hundred
// variables up to what the fast load path reaches: gram.hip, gram_mid_applies) in ONE launch: upper triangle of A'A, q = 2 A'c and c'c.
//
// Reference semantics replaced: _vecdot!/muladd! literal expansion (src/functions.jl:702-709,548-576) + canonicalize!
// (src/functions.jl:381-386, src/util.jl:9-26) + update!(::MOI.ScalarQuadraticFunction) (src/moi_interop.jl:45-62): SURVEY Appendix A.3.
//
// Why a third form beside gram_sk.hip / gram_tall.hip (profiles/r06_gram_mid.txt): on 128 x 128 tiles a split tile's partial is 128 KB — as
// large as the 64 rows x 256 columns a work unit reads.  4096 x 512 wrote and re-read 84 MB of partials through a second (fix-up) launch:
// 54.8 us for 13.7 us of flops.  Here
//   * tiles are 64 x 64 (a partial is 32 KB); a workgroup takes one (tile, row chunk); its four waves split the chunk's 8-row groups among
//     themselves (a split of the contraction index: each wave holds the WHOLE tile, 16 blocks x 4 rotations = 64 accumulators, so a
//     k-step's 4 + 4 operand loads feed 64 MFMAs) and stream their rows straight from global memory (L2 / Infinity Cache) in the MFMA
//     operand layout — no barrier, no shared panel; the rotated B operands come out of the wave's private piece of LDS (gram_tall.hip:
//     gram_stream_kernel); an 8-row group and the loads of the group D further on are one pinned instruction stream (mid_step);
//   * diagonal tiles compute their 10 upper blocks and carry q = A'c on the matrix pipe (B operand = c in every lane of the slot); they are
//     split into fewer chunks than the off-diagonal ones (40 against 64 MFMAs per k-step);
//   * the waves' sums are added through LDS in a fixed order; a split tile's partials go to the workspace and the LAST workgroup of the tile
//     to arrive (one atomic counter per tile, release / acquire fences at agent scope — nobody spins, no co-residency needed) adds them in
//     CHUNK order, whatever the arrival order: deterministic sums; it then writes the tile's terms as row-contiguous 16-byte chunks;
//   * c'c is one more workgroup of the same launch, in the order pmt_quad_gram_constant_order reports as 5 (gram_sk.hip: sk_lin_role),
//     restated bit for bit by tests/gpu_util.py.
// Summation order of a coefficient (fixed by (rows, cols) alone): wave w of chunk c adds the 8-row groups c gpc + w, + 4, .. in MFMA k
// order (rows 2 k, 2 k + 1 of a group in contraction slot k); waves (0 + 2) + (1 + 3); chunks in ascending order.
#include "gram_common.h"

#ifndef PMT_MID_WPS
#define PMT_MID_WPS 1              // waves per SIMD the register budget aims at
#endif
#ifndef PMT_MID_G
#define PMT_MID_G 256              // workgroups per round (one per CU: 202 + 128 registers per lane)
#endif
#ifndef PMT_MID_MAXWG
#define PMT_MID_MAXWG 2304         // workgroups at most (several rounds where the tiles alone are more than half of the CUs)
#endif
#ifndef PMT_MID_D
#define PMT_MID_D 2                // iterations (8-row groups) in flight per wave (pinned stream, PMT_MID_SCHED 2: 2 33.6 us at 4096 x 512, 3 34.0, 4 35.4; 4096 x 1024 97.7 / 100.3 / 103.7 — with mid_compute + mid_load it was 3)
#endif
#ifndef PMT_MID_GROUP_US
#define PMT_MID_GROUP_US 1.15      // the cost model's time of one 8-row group per wave (mid_plan)
#endif
#ifndef PMT_MID_TAIL_US
#define PMT_MID_TAIL_US 0.95       // the same in the plan of unsplit tiles with split tails (mid_plan)
#endif
#ifndef PMT_MID_XCD
#define PMT_MID_XCD 1              // 0: workgroup ids tile-major (id = tile * S + chunk) whatever the size
#endif
#ifndef PMT_MID_SUPER
#define PMT_MID_SUPER 1            // XCD-aware order: the tiles of a chunk in 8 x 8 super-tiles (mid_tile_of)
#endif
#ifndef PMT_MID_PERSIST
#define PMT_MID_PERSIST 1          // launches of at least two rounds of work items run as PMT_MID_G persistent workgroups (gram_mid_kernel)
#endif
#ifndef PMT_MID_SB
#define PMT_MID_SB 8               // edge of a super-tile in tiles
#endif
#ifndef PMT_MID_FB
#define PMT_MID_FB 4               // chunks whose partials the last arriver loads together (64 loads per thread: a wave may have 63 outstanding; 8, or 16-byte loads: no faster)
#endif
#ifndef PMT_MID_SCHED
#define PMT_MID_SCHED 2            // 2: an 8-row group and the loads of the group D further on as one pinned instruction stream (mid_step); 1: mid_compute + mid_load with scheduling barriers between the phases (37.3 -> 34.3 us at 4096 x 512, 1250 -> 1189 at 65536 x 1024)
#endif
#ifndef PMT_MID_ABL
#define PMT_MID_ABL 0              // ablations (wrong results): 1 no MFMAs, 2 no loads after the first D groups, 3 no fold of the partials, 4 no epilogue
#endif
// The hand-over of a partial to the tile's last arriver.  0 (ships): the partial goes out as agent-scope write-through stores (sc1),
// s_waitcnt vmcnt(0), then the relaxed agent-scope count; the last arriver reads the partials with agent-scope loads — per-access
// coherence, the same contract as the pair fold of gram_sk.hip (a property of gfx942 / gfx950, which is all this library is built for).
// 1: the memory model's form — release fence (buffer_wbl2 sc1: the XCD's whole L2) in every workgroup, acquire fence (buffer_inv sc1) in the
// last arriver; measured in profiles/r06_gram_mid.txt.
#ifndef PMT_MID_FORMAL
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#define PMT_MID_FORMAL 1
#else
#define PMT_MID_FORMAL 0
#endif
#endif

#ifdef PMT_MID_TRACE
// debugging aid (tools/mid_trace.py): 100 MHz wall-clock stamps of the phases of every workgroup
__device__ unsigned long long g_mid_trace[1024 * 8];
#define MID_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_mid_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
extern "C" int pmt_mid_trace_read(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mid_trace), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : 1;
}
#else
#define MID_STAMP(k) do { } while (0)
#endif

namespace pmt {

constexpr int MT = 64;                       // tile edge
constexpr int MACC = 64;                     // accumulators per lane: (block row tm, block column tn, rotation) = (a >> 4, (a >> 2) & 3, a & 3)
constexpr int MPART = MACC * 64;             // doubles of a tile partial, [a][lane]
constexpr int MSTRIDE = MPART + MT + 8;      // + q partial (diagonal tiles), padded to 64 bytes
constexpr int MPITCH = MT + 1;
constexpr int MFG = 8;                       // chunks per group of a two-level fold (mid_body)
constexpr int MCNT = 16;                     // counter words per tile: [0] the tile's, [1 + group] the groups' (at most 64 chunks)
constexpr int MQBUF = 4 * 2048;              // LDS (doubles): [0, 8192) rotation pieces / the waves' exchange pieces / the finished tile + q + row descriptors;
constexpr int MFLAG = MQBUF + 4 * MT;        // the waves' q (4 x 64); the count's old value; the tile's variable maps (2 x 64)
constexpr int MMAPS = MFLAG + 8;
constexpr int MSH = MMAPS + 2 * MT;          // 68.7 KB per workgroup

struct MidArgs {
    const double *A; int64_t lda, rows, cols;
    const double *b; int sign;               // c_i = 0.0 (+|-) b[i]; null / 0: c = 0
    const int64_t *xvar; const int64_t *varmap; int moi;
    QT *out_quad; double *out_csc; double alpha; LT *out_lin; double *out_const;
    int nb;                                  // 64-column panels
    int n_off, s_off, s_diag;                // strictly upper tiles, row chunks of one, row chunks of a diagonal tile
    int gpc_off, gpc_diag;                   // 8-row groups per chunk
    int n_tail, s_tail, gpc_tail;            // the LAST n_tail strictly upper tiles (in the order they are walked) in s_tail chunks instead (mid_plan)
    double *ws;                              // one MSTRIDE slot per workgroup
    unsigned *counters;                      // one per tile, zero between launches (the last arriver re-arms its tile's)
    int xcd;                                 // workgroup ids in the XCD-aware order (gram_mid_kernel)
    int persist, total;                      // PMT_MID_G persistent workgroups walk the `total` work items (gram_mid_kernel)
    unsigned *tickets;                       // 8 per-XCD tickets + the count of workgroups that have left; zero between launches
};

struct MidPlan { int nb, n_off, s_off, s_diag, gpc_off, gpc_diag, n_tail, s_tail, gpc_tail, wgs; };

// Row chunks per tile, from (rows, cols) alone (the summation order depends on them).  s chunks per off-diagonal tile and ceil(5 s / 8) per
// diagonal one (88 against 128 MFMAs per 8-row group) make W workgroups that run one per CU in ceil(W / PMT_MID_G) rounds of
// ceil(groups per chunk / 4) iterations of ~1.15 us each + ~5 us per workgroup (first loads, the waves' sums; + 1 for a partial); a
// split tile's last arriver then reads s partials (mid_fold_us: 5.5 us measured at s = 7 .. 9, tools/mid_trace.py) — once in the
// launch's tail and, summed over the tiles, as work of the CUs.  The s with the smallest estimate, at most PMT_MID_MAXWG workgroups (33 KB of workspace each).
// what adding `count` partials costs the last arriver: one round of loads per PMT_MID_FB partials (~2.6 us each, tools/mid_trace.py)
static double mid_fold_us(int count) { return 0.5 + 2.6 * (double)cdiv(count, PMT_MID_FB); }

static MidPlan mid_plan(int64_t rows, int64_t cols) {
    MidPlan p;
    p.nb = (int)cdiv(cols, MT);
    p.n_off = p.nb * (p.nb - 1) / 2;
    const int ngroups = (int)std::max<int64_t>(1, cdiv(rows, 8));
    const int maxs = std::max(1, ngroups / 4);
    auto sd = [&](int s) { return std::min(maxs, std::max(1, (5 * s + 7) / 8)); };
    int best = 1;
    double best_t = 1e300;
    for (int s = 1; s <= std::min(maxs, 64); ++s) {
        const int64_t wgs = (int64_t)p.n_off * s + (int64_t)p.nb * sd(s);
        if (s > 1 && wgs > PMT_MID_MAXWG) break;
        const double it_off = (double)cdiv(cdiv(ngroups, s), 4), it_diag = 0.69 * (double)cdiv(cdiv(ngroups, sd(s)), 4);
        const double fold = s <= 1 ? 0.0 : s <= MFG ? mid_fold_us(s) : mid_fold_us(MFG) + 1.5 + mid_fold_us((int)cdiv(s, MFG));
        const double t = (double)cdiv(wgs, PMT_MID_G) * (PMT_MID_GROUP_US * (p.n_off ? std::max(it_off, it_diag) : it_diag) + 5.0 + (s > 1 ? 1.0 : 0.0)) +
                         fold * (1.0 + (double)(p.n_off + p.nb) / PMT_MID_G);
        if (t < best_t) { best_t = t; best = s; }
    }
    int s = best;
#ifdef PMT_TUNING
    if (const char *e = getenv("PMT_MID_S")) { const int v = atoi(e); if (v > 0) s = std::min(v, maxs); }      // (measurement builds only)
#endif
    // UNSPLIT off-diagonal tiles in several rounds of PMT_MID_G workgroups (config 2: 2016 of them + 64 diagonal ones = 8.1 rounds): what the
    // rounds do not divide is split instead of costing a whole tile time on a few CUs —
    //   * the LAST n_off % G off-diagonal tiles (the partial round) in s_tail chunks each (4096 x 3072: 1128 tiles = 4 rounds + 104 tiles,
    //     152 CUs idle for a tile time; in two chunks they are one round of half the length);
    //   * the diagonal tiles (last in the workgroup order) in chunks that fill the CUs the last round leaves free, the rest short rounds
    //     (config 2: instead of a ninth round of 0.69 tile times for 32 workgroups).
    // Both counts by the shortest estimated tail; compared with the best uniform split above.
    int sdiag = sd(s), ntail = 0, stail = 1;
    if (p.n_off + p.nb > PMT_MID_G) {
        // (0.95 us per 8-row group and wave: the pinned loop's time in these long workgroups, tools/mid_trace.py — config 2 1101 -> 1085 us
        // against the uniform model's 1.15, which stays where it decides the summation order of the single-round shapes)
        auto wg_us = [&](int c, double w) { return PMT_MID_TAIL_US * w * (double)cdiv(cdiv(ngroups, c), 4) + 5.0 + (c > 1 ? 1.0 : 0.0); };
        const int rem = p.n_off % PMT_MID_G;
        int ct = 1;
        double tail_off = 0.0;
        if (rem) {
            tail_off = 1e300;
            for (int c = 1; c <= std::min(maxs, MFG); ++c) {
                const double t = (double)cdiv((int64_t)rem * c, PMT_MID_G) * wg_us(c, 1.0) + (c > 1 ? mid_fold_us(c) : 0.0);
                if (t < tail_off) { tail_off = t; ct = c; }
            }
        }
        const int last = rem ? (int)(((int64_t)rem * ct) % PMT_MID_G) : 0, free_cus = last ? PMT_MID_G - last : 0;
        const double t_last = rem ? wg_us(ct, 1.0) : wg_us(1, 1.0);
        int cd = 1;
        double tail_diag = 1e300;
        for (int c = 1; c <= std::min(maxs, MFG); ++c) {
            const double t_d = wg_us(c, 0.69);
            const int64_t inside = (int64_t)free_cus * (int64_t)(t_last / t_d);       // chunks done beside the last off-diagonal round
            const int64_t left = std::max<int64_t>(0, (int64_t)p.nb * c - inside);
            const double t = (double)cdiv(left, PMT_MID_G) * t_d + (c > 1 ? mid_fold_us(c) : 0.0);
            if (t < tail_diag) { tail_diag = t; cd = c; }
        }
        const double t1 = (double)(p.n_off / PMT_MID_G) * wg_us(1, 1.0) + tail_off + tail_diag;
        // (the uniform model counts PMT_MID_GROUP_US per group)
        if (s == 1 || t1 * (PMT_MID_GROUP_US / PMT_MID_TAIL_US) < best_t) { s = 1; sdiag = cd; ntail = ct > 1 ? rem : 0; stail = ct > 1 ? ct : 1; }
    }
#ifdef PMT_TUNING
    if (const char *e = getenv("PMT_MID_TAIL")) { if (atoi(e) == 0) { ntail = 0; stail = 1; } }          // (measurement builds only)
#endif
    p.gpc_off = (int)cdiv(ngroups, s);
    p.s_off = (int)cdiv(ngroups, p.gpc_off);
    p.gpc_diag = (int)cdiv(ngroups, sdiag);
    p.s_diag = (int)cdiv(ngroups, p.gpc_diag);
    p.n_tail = ntail;
    p.gpc_tail = (int)cdiv(ngroups, stail);
    p.s_tail = ntail ? (int)cdiv(ngroups, p.gpc_tail) : 1;
    if (p.s_tail <= 1) { p.n_tail = 0; p.s_tail = 1; p.gpc_tail = ngroups; }
    p.wgs = (p.n_off - p.n_tail) * p.s_off + p.n_tail * p.s_tail + p.nb * p.s_diag + 1;
    return p;
}

// one 8-row group of one wave: buf[t] = rows (row0 + 2 lk, + 1) of column (t < 4 ? cj0 : ck0) + 16 (t & 3) + lm; cb = the same rows of b.
// FAST (A 16-byte aligned with an even pitch, b 16-byte aligned, the matrix within 4 GiB): scalar base + 32-bit lane offset, no mask (a
// column beyond the matrix reads the LAST column: its products land in entries nobody stores).  Otherwise 8-byte loads from clamped
// addresses, what lies outside replaced by 0.0 — no branch around a load (gram_tall.hip: stream_load).
template <bool DIAG, bool FAST>
__device__ __forceinline__ void mid_load(const MidArgs &g, int64_t row0, int lane, int64_t cj0, int64_t ck0, const unsigned (&voff)[DIAG ? 4 : 8],
                                         f64x2 (&buf)[DIAG ? 4 : 8], f64x2 &cb) {
    constexpr int NG = DIAG ? 4 : 8;
    const int lm = lane & 15, lk = lane >> 4;
    if (FAST) {
        const char *base = reinterpret_cast<const char *>(g.A + row0);
#pragma unroll
        for (int t = 0; t < NG; ++t) buf[t] = *reinterpret_cast<const f64x2 *>(base + voff[t]);
        cb.x = 0.0; cb.y = 0.0;
        if (DIAG && g.b) cb = *reinterpret_cast<const f64x2 *>(g.b + row0 + 2 * lk);
        return;
    }
    const int64_t row = row0 + 2 * lk;
    const bool r0 = row < g.rows, r1 = row + 1 < g.rows;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        const int64_t col = (t < 4 ? cj0 : ck0) + 16 * (t & 3) + lm;
        const bool cv = col < g.cols;
        const bool v0 = cv && r0, v1 = cv && r1;
        const double *p0 = g.A + (v0 ? col * g.lda + row : 0), *p1 = g.A + (v1 ? col * g.lda + row + 1 : 0);
        const double x = *p0, y = *p1;
        buf[t].x = v0 ? x : 0.0;
        buf[t].y = v1 ? y : 0.0;
    }
    cb.x = 0.0; cb.y = 0.0;
    if (DIAG && g.b) {
        const double x = g.b[r0 ? row : 0], y = g.b[r1 ? row + 1 : 0];
        cb.x = r0 ? x : 0.0;
        cb.y = r1 ? y : 0.0;
    }
}

// the MFMAs of one 8-row group.  The B operand of rotation r of a block column is the value the lane 4 r further up its 16-lane row holds:
// the wave stores its four B groups into its private piece of LDS and reads them back rotated (a wave's LDS operations execute in order).
template <bool DIAG>
__device__ __forceinline__ void mid_compute(double *__restrict__ rot, const f64x2 (&buf)[DIAG ? 4 : 8], const f64x2 &cb, int sign, int lane,
                                            double (&acc)[MACC], double (&qacc)[4]) {
    constexpr int BO = DIAG ? 0 : 4;
    const int lm = lane & 15, lrow = lane & 48;
#pragma unroll
    for (int t = 0; t < (PMT_MID_ABL == 5 || PMT_MID_ABL == 6 ? 0 : 4); ++t) *reinterpret_cast<f64x2 *>(rot + (t * 64 + lane) * 2) = buf[BO + t];
    // ALL rotated operands are asked for up front, and the MFMAs that need none of them (rotation 0: the loaded values themselves) go
    // first: the LDS round trip hides behind 32 MFMAs.  Read rotation by rotation in front of their first use, every wait for an LDS
    // read stood in the wave's one issue stream (one wave per SIMD): ~10 exposed waits per 8-row group, the loop at 0.55 of the pipe.
    f64x2 bv[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c][0] = buf[BO + c];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            if (PMT_MID_ABL == 5 || PMT_MID_ABL == 6) { bv[c][r] = buf[BO + (PMT_MID_ABL == 6 ? 0 : c)]; continue; }      // (ablation: no rotations)
            bv[c][r] = *reinterpret_cast<const f64x2 *>(rot + (c * 64 + lrow + ((lm + 4 * r) & 15)) * 2);
        }
    }
#if PMT_MID_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    constexpr int NC = PMT_MID_ABL == 1 ? 0 : 4;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int tm = 0; tm < (DIAG ? c + 1 : 4); ++tm) {
            const int a = (tm * 4 + c) * 4;
            acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm].x, bv[c][0].x, acc[a], 0, 0, 0);
        }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int tm = 0; tm < (DIAG ? c + 1 : 4); ++tm) {
            const int a = (tm * 4 + c) * 4;
            acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm].y, bv[c][0].y, acc[a], 0, 0, 0);
        }
#if PMT_MID_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
#pragma unroll
            for (int tm = 0; tm < (DIAG ? c + 1 : 4); ++tm) {
                const int a = (tm * 4 + c) * 4 + r;
                acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm].x, bv[c][r].x, acc[a], 0, 0, 0);
            }
#pragma unroll
        for (int r = 1; r < 4; ++r)
#pragma unroll
            for (int tm = 0; tm < (DIAG ? c + 1 : 4); ++tm) {
                const int a = (tm * 4 + c) * 4 + r;
                acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[tm].y, bv[c][r].y, acc[a], 0, 0, 0);
            }
    }
    if (DIAG) {
        // q = A'c on the matrix pipe: the B operand is c in every lane of the contraction slot, so block b of the result holds q of the
        // columns 4 b .. 4 b + 3 of the group (four copies)
        const double c0 = signed_const(cb.x, sign), c1 = signed_const(cb.y, sign);
#pragma unroll
        for (int t = 0; t < 4; ++t) qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t].x, c0, qacc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t].y, c1, qacc[t], 0, 0, 0);
    }
}

// One 8-row group AND the loads of the group D further on, as ONE pinned instruction stream (PMT_MID_SCHED 2, the FAST load path).  The
// workgroup has one wave per SIMD: whatever is not an MFMA stands in the wave's only issue stream.  mid_compute + mid_load put the 16 LDS
// operations of a group in one run in front of the MFMAs and its 8 loads (with a 64-bit vector add each: the compiler hoists the zero
// extension of the lane offsets out of the loop and loses the scalar-base form) in runs behind them: ~18.5 clocks per MFMA against 16.6 for
// bare MFMAs.  Here every LDS operation and every load stands alone behind an MFMA (the matrix pipe is busy for 16 clocks after an issue):
//   * the 32 (20 + 8 of q on a diagonal tile) MFMAs that need no rotation first, one LDS operation behind each of the first 16;
//   * the rotated MFMAs block ROW by block row: an A operand's registers are free after its row — its reload is issued there, the B
//     operands' reloads (free after the first phase) behind the first MFMAs of the second;
//   * loads in the scalar-base form (the lane offset passes through an empty asm: nothing to hoist).
// Every accumulator sees the same operations in the same order as in mid_compute (x before y of each group): the same bits.
#define MID_SB() __builtin_amdgcn_sched_barrier(0)
template <bool DIAG>
__device__ __forceinline__ void mid_step(const MidArgs &g, double *__restrict__ rot, int64_t next_row0, unsigned (&voff)[DIAG ? 4 : 8],
                                         f64x2 (&buf)[DIAG ? 4 : 8], f64x2 &cb, int sign, int lane, double (&acc)[MACC], double (&qacc)[4]) {
    constexpr int BO = DIAG ? 0 : 4;
    const int lm = lane & 15, lrow = lane & 48, lk = lane >> 4;
    const char *base = reinterpret_cast<const char *>(g.A + next_row0);
    auto reload = [&](int t) {
        asm volatile("" : "+v"(voff[t]));
        buf[t] = *reinterpret_cast<const f64x2 *>(base + voff[t]);
    };
    f64x2 bv[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c][0] = buf[BO + c];
    auto lds_op = [&](int i) {                       // 0..3: the B groups out; 4..15: the rotated reads
        if (i < 4) *reinterpret_cast<f64x2 *>(rot + (i * 64 + lane) * 2) = buf[BO + i];
        else {
            const int c = (i - 4) / 3, r = 1 + (i - 4) % 3;
            bv[c][r] = *reinterpret_cast<const f64x2 *>(rot + (c * 64 + lrow + ((lm + 4 * r) & 15)) * 2);
        }
    };
    MID_SB();
    // phase 0: rotation 0 (operands as loaded), x then y; an LDS operation behind each of the first 16
    int k = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int tm = 0; tm < (DIAG ? c + 1 : 4); ++tm) {
                const int a = (tm * 4 + c) * 4;
                acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(pass ? buf[tm].y : buf[tm].x, pass ? bv[c][0].y : bv[c][0].x, acc[a], 0, 0, 0);
                if (k < 16) { MID_SB(); lds_op(k); MID_SB(); }
                ++k;
            }
    if (DIAG) {
        // q = A'c on the matrix pipe (mid_compute)
        const double c0 = signed_const(cb.x, sign), c1 = signed_const(cb.y, sign);
#pragma unroll
        for (int t = 0; t < 4; ++t) qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t].x, c0, qacc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) qacc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(buf[t].y, c1, qacc[t], 0, 0, 0);
        MID_SB();
        // (no b: sign is 0 and signed_const ignores what is loaded — any valid address keeps the block free of branches)
        cb = *reinterpret_cast<const f64x2 *>((g.b ? g.b : g.A) + next_row0 + 2 * lk);
    }
    MID_SB();
    // phase 1: the rotated blocks, block row by block row
    int nb = 0;                                      // B reloads issued (off-diagonal tiles: buf[4..7])
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int c = (DIAG ? tm : 0); c < 4; ++c)
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    const int a = (tm * 4 + c) * 4 + r;
                    acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(pass ? buf[tm].y : buf[tm].x, pass ? bv[c][r].y : bv[c][r].x, acc[a], 0, 0, 0);
                    if (!DIAG && nb < 4) { MID_SB(); reload(4 + nb); MID_SB(); ++nb; }
                }
        MID_SB();
        reload(tm);
        MID_SB();
    }
}

__host__ __device__ constexpr bool mid_used(bool diag, int a) { return !diag || (a >> 4) <= ((a >> 2) & 3); }

// (row, col) inside the tile of accumulator a of a lane (gram_common.h: sk_acc_pos, on the 4 x 4 block grid of the tile)
__device__ __forceinline__ void mid_pos(int lane, int a, int &row, int &col) {
    const int i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
    row = 16 * (a >> 4) + 4 * b + i;
    col = 16 * ((a >> 2) & 3) + 4 * ((b + (a & 3)) & 3) + j;
}

// the mapped variable indices of the tile's 64 columns / 64 rows, gathered at the TOP of the workgroup's life (two dependent loads that
// would otherwise stand in the tail of the tile's last arriver); the barriers of the wave sums order them before the epilogue
__device__ __forceinline__ void mid_maps(const MidArgs &g, double *sh, int tid, int jb, int kb) {
    u64 *cmap = reinterpret_cast<u64 *>(sh + MMAPS);
    if (tid >= 2 * MT || (!g.out_quad && jb != kb)) return;
    const int64_t i = (tid < MT ? (int64_t)kb * MT + tid : (int64_t)jb * MT + tid - MT);
    const int64_t v = i < g.cols ? g.xvar[i] : 1;
    cmap[tid] = (u64)(g.moi ? map_var(g.varmap, v) : v);
}

// The finished tile (sh[row * MPITCH + col], q of a diagonal tile in sh[MT * MPITCH ..]) -> terms.  Row j of the tile's entries
// k = max(j, k0) .. are consecutive QuadraticTerms of the canonical upper triangle: one wave per row, 16-byte chunks.
__device__ __forceinline__ void mid_epilogue(const MidArgs &g, double *sh, int tid, int jb, int kb) {
    const int wave = tid >> 6, lane = tid & 63;
    const int64_t n = g.cols, j0 = (int64_t)jb * MT, k0 = (int64_t)kb * MT;
    double *tile = sh;
    const double *qfin = sh + MT * MPITCH;
    const u64 *cmap = reinterpret_cast<const u64 *>(sh + MMAPS);          // (mid_maps, at the top of the kernel)
    const u64 *rmap = cmap + MT;
    if (g.out_csc) {
        const int64_t j = j0 + lane;
#pragma unroll 4
        for (int col = wave; col < MT; col += 4) {
            const int64_t k = k0 + col;
            double c = tile[lane * MPITCH + col];
            if (g.moi || j != k) c = 2 * c;
            if (k < n && j <= k) g.out_csc[k * (k + 1) / 2 + j] = g.alpha * c;
        }
    }
    if (g.out_quad) {
        // One descriptor per tile row (first word of its segment in the term array, term count, first column, diagonal flag), then a
        // wave per row, a lane per TERM: two unconditional LDS reads and three 8-byte stores (scalar row base + 24 * lane + immediate).
        // The workgroup is alone with the tile (one wave per SIMD), so what counts is the instruction count: 16-byte chunks assembled
        // word by word took 15 us per tile as a wave per row with conditional reads, 6.9 us as a flat chunk list, (tools/mid_trace.py).
        u64 *out = reinterpret_cast<u64 *>(g.out_quad);
        u64 *desc = reinterpret_cast<u64 *>(sh + MT * MPITCH + MT);
        if (tid < MT) {
            const int64_t j = j0 + tid;
            const int64_t kstart = j > k0 ? j : k0;
            const int64_t kend = (k0 + MT < n) ? k0 + MT : n;
            const int64_t nterms = (j < n && kend > kstart) ? kend - kstart : 0;
            desc[2 * tid] = (u64)(j * n - (j * (j - 1)) / 2 + (kstart - j)) * 3;
            desc[2 * tid + 1] = (u64)nterms | (u64)(kstart - k0) << 16 | (u64)(kstart == j ? 1 : 0) << 24;
        }
        __syncthreads();
        const bool moi = g.moi != 0;
        const int uwave = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll 4
        for (int i = 0; i < MT / 4; ++i) {
            const int row = 4 * i + uwave;
            const u64 w0 = desc[2 * row], info = desc[2 * row + 1];
            const int nterms = (int)(info & 0xffff), coff = (int)((info >> 16) & 0xff);
            const bool dg = ((info >> 24) & 1) != 0;          // the segment starts on the diagonal: its first term is not doubled in native mode
            double c = tile[row * MPITCH + coff + lane];
            const u64 m = cmap[(coff + lane) & (MT - 1)], rv = rmap[row];
            // off-diagonal: (j,k)+(k,j) combined; diagonal: MOI doubling (moi_interop.jl:58)
            if (moi || !(dg && lane == 0)) c = 2 * c;
            if (lane < nterms) {
                u64 *d = out + w0 + 3 * lane;
                d[0] = (u64)__double_as_longlong(c);
                d[1] = rv;
                d[2] = m;
            }
        }
    }
    if (jb == kb && tid < MT) {
        const int64_t j = j0 + tid;
        if (j < n) {
            LT t;
            t.coeff = 2 * qfin[tid];
            t.var = (int64_t)rmap[tid];
            g.out_lin[j] = t;
        }
    }
}

// c'c in order 5 (gram_sk.hip: sk_lin_role): virtual thread t of 512 adds rows t, t + 512, ..; a shuffle tree per virtual wave; the eight
// virtual waves in order.  Thread tid of this 256-thread workgroup is the virtual threads tid and tid + 256.
__device__ __forceinline__ void mid_constant(const MidArgs &g, double *sh, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    double s0 = 0.0, s1 = 0.0;
    if (g.b && g.sign) {
        for (int64_t i = tid; i < g.rows; i += 512) { const double c = signed_const(g.b[i], g.sign); s0 = s0 + c * c; }
        for (int64_t i = tid + 256; i < g.rows; i += 512) { const double c = signed_const(g.b[i], g.sign); s1 = s1 + c * c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 = s0 + __shfl_down(s0, off, 64); s1 = s1 + __shfl_down(s1, off, 64); }
    if (lane == 0) { sh[wave] = s0; sh[4 + wave] = s1; }
    __syncthreads();
    if (tid == 0) {
        double v = sh[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) v = v + sh[w];
        *g.out_const = v;
    }
}

__device__ __forceinline__ void mid_put(double *p, double v) {
#if PMT_MID_FORMAL
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ double mid_get(const double *p) {
#if PMT_MID_FORMAL
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Behind the main loop, as seen by wave W.  The four waves' sums, (0 + 2) + (1 + 3), by halving: wave W hands the half of its 64
// accumulators that its partner W ^ 2 keeps to that partner through LDS and adds the partner's other half to its own; then the same
// with quarters and the partner W ^ 1 — every wave ends up OWNING the finished sums of one block row (a = 16 W .. 16 W + 15), each
// the same bits whichever side added (a + b = b + a).  Two waves summing everything for the others took 2.5 us of the workgroup's tail
// (accumulator moves + adds in one issue stream, tools/mid_trace.py); here every wave moves 48 and adds 48.  The owned quarter then goes
// to the workspace (a split tile) or into the finished tile in LDS; q of a diagonal tile is added by wave 0 (lane = column).
template <bool DIAG, int W>
__device__ __forceinline__ void mid_tail(const MidArgs &g, double *sh, int lane, double (&acc)[MACC], const double (&qacc)[4], int nchunk, double *w) {
    constexpr int KEEP1 = (W >> 1) * 32, GIVE1 = 32 - KEEP1, KEEP2 = KEEP1 + (W & 1) * 16, GIVE2 = KEEP1 + 16 - (W & 1) * 16;
    double *mine = sh + W * 2048;
    const double *half = sh + (W ^ 2) * 2048, *quarter = sh + (W ^ 1) * 2048;
    const bool qlane = (lane & 3) == 0;
    const int qcol = 4 * ((lane >> 2) & 3) + (lane >> 4);          // q of column 16 t + qcol sits in qacc[t] of the lanes with lane & 3 == 0
#pragma unroll
    for (int a = 0; a < 32; a += 2)
        if (mid_used(DIAG, GIVE1 + a)) { f64x2 v; v.x = acc[GIVE1 + a]; v.y = acc[GIVE1 + a + 1]; *reinterpret_cast<f64x2 *>(mine + (a * 32 + lane) * 2) = v; }
    if (DIAG && qlane) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sh[MQBUF + W * MT + 16 * t + qcol] = qacc[t];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 32; a += 2)
        if (mid_used(DIAG, KEEP1 + a)) {
            const f64x2 v = *reinterpret_cast<const f64x2 *>(half + (a * 32 + lane) * 2);
            acc[KEEP1 + a] = acc[KEEP1 + a] + v.x; acc[KEEP1 + a + 1] = acc[KEEP1 + a + 1] + v.y;
        }
    double qsum = 0.0;
    if (DIAG && W == 0) qsum = (sh[MQBUF + lane] + sh[MQBUF + 2 * MT + lane]) + (sh[MQBUF + MT + lane] + sh[MQBUF + 3 * MT + lane]);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; a += 2)
        if (mid_used(DIAG, GIVE2 + a)) { f64x2 v; v.x = acc[GIVE2 + a]; v.y = acc[GIVE2 + a + 1]; *reinterpret_cast<f64x2 *>(mine + (a * 32 + lane) * 2) = v; }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; a += 2)
        if (mid_used(DIAG, KEEP2 + a)) {
            const f64x2 v = *reinterpret_cast<const f64x2 *>(quarter + (a * 32 + lane) * 2);
            acc[KEEP2 + a] = acc[KEEP2 + a] + v.x; acc[KEEP2 + a + 1] = acc[KEEP2 + a + 1] + v.y;
        }
    __syncthreads();                                               // (the finished tile overwrites the exchange pieces)
    if (nchunk == 1) {
#pragma unroll
        for (int a = KEEP2; a < KEEP2 + 16; ++a) {
            if (!mid_used(DIAG, a)) continue;
            int row, col;
            mid_pos(lane, a, row, col);
            sh[row * MPITCH + col] = acc[a];
        }
        if (DIAG && W == 0) sh[MT * MPITCH + lane] = qsum;
        return;
    }
#pragma unroll
    for (int a = KEEP2; a < KEEP2 + 16; ++a) if (mid_used(DIAG, a)) mid_put(&w[a * 64 + lane], acc[a]);
    if (DIAG && W == 0) mid_put(&w[MPART + lane], qsum);
    // the partial is visible device-wide before the tile's count goes up (mid_body: a barrier, then the count)
#if PMT_MID_FORMAL
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    __builtin_amdgcn_s_waitcnt(0);
#endif
}

// sum (in ascending order) of `count` partials `stride` doubles apart, LOADED PMT_MID_FB at a time (the sum is not a chain of round trips to
// the fabric); thread tid: the elements e = tid + 256 u of the [a][lane] layout, and q of column tid of a diagonal tile
template <bool DIAG>
__device__ __forceinline__ void mid_fold(const double *p, int count, int64_t stride, int tid, double (&sum)[16], double &qs) {
#pragma unroll
    for (int u = 0; u < 16; ++u) sum[u] = 0.0;
    qs = 0.0;
    for (int c0 = 0; c0 < count; c0 += PMT_MID_FB) {
        double v[PMT_MID_FB][16], qv[PMT_MID_FB];
#pragma unroll
        for (int cc = 0; cc < PMT_MID_FB; ++cc) {
            const double *pc = p + (int64_t)min(c0 + cc, count - 1) * stride;          // (beyond the last one: a repeated load that is not added)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int a = (tid >> 6) + 4 * u;
                v[cc][u] = 0.0;
                if (mid_used(DIAG, a)) v[cc][u] = mid_get(pc + tid + 256 * u);
            }
            qv[cc] = 0.0;
            if (DIAG && tid < MT) qv[cc] = mid_get(pc + MPART + tid);
        }
#pragma unroll
        for (int cc = 0; cc < PMT_MID_FB; ++cc) {
            const bool live = c0 + cc < count;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const double t = sum[u] + v[cc][u]; sum[u] = live ? t : sum[u]; }
            const double t = qs + qv[cc];
            qs = live ? t : qs;
        }
    }
}

template <bool DIAG, bool FAST>
__device__ __forceinline__ void mid_body(const MidArgs &g, double *sh, int tid, int jb, int kb, int chunk, int nchunk, int gpc, int first_wg, unsigned *counter) {
    constexpr int NG = DIAG ? 4 : 8;
    constexpr int D = PMT_MID_D;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, lk = lane >> 4;
    const int64_t cj0 = (int64_t)jb * MT, ck0 = (int64_t)kb * MT;
    const int ngroups = (int)((g.rows + 7) >> 3);
    const int g0 = chunk * gpc + wave, g1 = min((chunk + 1) * gpc, ngroups);
    const int my = g1 > g0 ? (g1 - g0 + 3) / 4 : 0;                       // groups g0, g0 + 4, .. of this wave
    // the one ragged group (rows % 8 rows) is the last group of the matrix: loaded with masks, outside the pipelined loop
    const bool ragged = FAST && (g.rows & 7) && my > 0 && g0 + 4 * (my - 1) == ngroups - 1;
    const int nfast = ragged ? my - 1 : my;

    MID_STAMP(0);
    mid_maps(g, sh, tid, jb, kb);
    double acc[MACC], qacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < MACC; ++a) acc[a] = 0.0;
    unsigned voff[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) voff[t] = (unsigned)((min((t < 4 ? cj0 : ck0) + 16 * (t & 3) + lm, g.cols - 1) * g.lda + 2 * lk) * 8);
    double *rot = sh + wave * 512;
    f64x2 buf[D][NG], cb[D];
    auto row_of = [&](int s) { return (int64_t)(g0 + 4 * min(s, max(nfast - 1, 0))) * 8; };
    if (nfast > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) mid_load<DIAG, FAST>(g, row_of(d), lane, cj0, ck0, voff, buf[d], cb[d]);
        if (PMT_MID_SCHED == 2 && FAST && PMT_MID_ABL == 0) {
            // whole rounds of D groups as one branch-free block (a branch inside it and the compiler's vmcnt waits fall back to "all
            // loads": the number of loads in flight must not depend on the path); the loads beyond the last group repeat the last group
            int s0 = 0;
            for (; s0 + D <= nfast; s0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) mid_step<DIAG>(g, rot, row_of(s0 + d + D), voff, buf[d], cb[d], g.sign, lane, acc, qacc);
            }
#pragma unroll
            for (int d = 0; d < D - 1; ++d)
                if (s0 + d < nfast) mid_compute<DIAG>(rot, buf[d], cb[d], g.sign, lane, acc, qacc);
        } else {
            for (int s0 = 0; s0 < nfast; s0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (s0 + d < nfast) mid_compute<DIAG>(rot, buf[d], cb[d], g.sign, lane, acc, qacc);
                    if (PMT_MID_ABL != 2) mid_load<DIAG, FAST>(g, row_of(s0 + d + D), lane, cj0, ck0, voff, buf[d], cb[d]);
                }
            }
        }
    }
    if (ragged) {
        mid_load<DIAG, false>(g, (int64_t)(ngroups - 1) * 8, lane, cj0, ck0, voff, buf[0], cb[0]);
        mid_compute<DIAG>(rot, buf[0], cb[0], g.sign, lane, acc, qacc);
    }

    MID_STAMP(1);
    unsigned *flag = reinterpret_cast<unsigned *>(sh + MFLAG);
    double *w = g.ws + (int64_t)(first_wg + chunk) * MSTRIDE;
    __syncthreads();                                               // every wave is done with its rotation piece
    if (wave == 0) mid_tail<DIAG, 0>(g, sh, lane, acc, qacc, nchunk, w);
    else if (wave == 1) mid_tail<DIAG, 1>(g, sh, lane, acc, qacc, nchunk, w);
    else if (wave == 2) mid_tail<DIAG, 2>(g, sh, lane, acc, qacc, nchunk, w);
    else mid_tail<DIAG, 3>(g, sh, lane, acc, qacc, nchunk, w);
    MID_STAMP(2);
    if (nchunk > 1) {
        // One level up to MFG chunks: the tile's last arriver adds them all.  Beyond (few tiles, tall: 65536 x 256 is 10 tiles of 29 chunks —
        // one CU reading 29 x 32 KB took 20 us of the node's 113), two levels: the last arriver of every GROUP of MFG consecutive chunks adds
        // its group and publishes the sum in the slot of the group's first chunk, the last of THOSE adds the groups' sums.  The order stays
        // fixed: chunks in order within a group, groups in order.
        const bool two = nchunk > MFG;
        const int gfirst = two ? (chunk / MFG) * MFG : 0;
        const int n1 = two ? min(MFG, nchunk - gfirst) : nchunk;
        unsigned *c1 = two ? counter + 1 + chunk / MFG : counter;
        __syncthreads();                                           // every wave's quarter of the partial has left (mid_tail waits for its stores)
        if (tid == 0) *flag = __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        MID_STAMP(3);
        if (PMT_MID_ABL == 3 || *flag != (unsigned)(n1 - 1)) return;                   // (workgroup-uniform) not the last one of this tile / group
#if PMT_MID_FORMAL
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        if (tid == 0) __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // re-armed for the next launch
        double sum[16], qs;
        double *p = g.ws + (int64_t)(first_wg + gfirst) * MSTRIDE;
        mid_fold<DIAG>(p, n1, MSTRIDE, tid, sum, qs);
        if (two) {
#pragma unroll
            for (int u = 0; u < 16; ++u) if (mid_used(DIAG, (tid >> 6) + 4 * u)) mid_put(p + tid + 256 * u, sum[u]);
            if (DIAG && tid < MT) mid_put(p + MPART + tid, qs);
#if PMT_MID_FORMAL
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
            __builtin_amdgcn_s_waitcnt(0);
#endif
            __syncthreads();                                       // (also: everyone has read the first count's old value)
            if (tid == 0) *flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int ng = (nchunk + MFG - 1) / MFG;
            if (*flag != (unsigned)(ng - 1)) return;
#if PMT_MID_FORMAL
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mid_fold<DIAG>(g.ws + (int64_t)first_wg * MSTRIDE, ng, (int64_t)MFG * MSTRIDE, tid, sum, qs);
        }
        __syncthreads();                                           // (the tile overwrites the exchange pieces)
        MID_STAMP(4);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int a = (tid >> 6) + 4 * u;
            if (!mid_used(DIAG, a)) continue;
            int row, col;
            mid_pos(lane, a, row, col);
            sh[row * MPITCH + col] = sum[u];
        }
        if (DIAG && tid < MT) sh[MT * MPITCH + tid] = qs;
    }
    __syncthreads();
    MID_STAMP(5);
    if (PMT_MID_ABL != 4) mid_epilogue(g, sh, tid, jb, kb);
#ifdef PMT_MID_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    MID_STAMP(6);
#endif
}

// position of workgroup `id` in the XCD-major order of the ids [base, base + n): the ids of XCD 0 (id % 8 == 0) first, in ascending order,
// then XCD 1's, ..
__device__ __forceinline__ int mid_xcd_rank(int id, int base, int n) {
    const int x = id & 7, end = base + n;
    int start = 0;
    for (int y = 0; y < x; ++y) {
        const int fy = base + ((y - base) & 7);                    // the first id of the range on XCD y
        start += fy < end ? ((end - 1 - fy) >> 3) + 1 : 0;
    }
    return start + ((id - (base + ((x - base) & 7))) >> 3);
}

// Strictly upper tile number t -> (jb, kb) in SUPER-TILE order (round 6c): 8 x 8 blocks of tiles, column by column; inside a block row by
// row (on the diagonal: column by column).  32 consecutive tiles — what an XCD runs at a time — are then 4 row panels x 8 column panels
// (12 panels through its L2) instead of one column panel with 32 row panels (33): config 2 moved 4.39 GB per launch over the fabric.
__device__ __forceinline__ void mid_tile_of(int t, int nb, int &jb, int &kb) {
    constexpr int SB = PMT_MID_SB;
    const int nsb = (nb + SB - 1) / SB;
    for (int K = 0; K < nsb; ++K) {
        const int wK = min(SB, nb - SB * K);
        for (int J = 0; J <= K; ++J) {
            const int cnt = J < K ? SB * wK : wK * (wK - 1) / 2;
            if (t >= cnt) { t -= cnt; continue; }
            if (J < K) { jb = SB * J + t / wK; kb = SB * K + t % wK; return; }
            int kk = 1;
            while (t >= kk) { t -= kk; ++kk; }
            jb = SB * K + t; kb = SB * K + kk;
            return;
        }
    }
    jb = 0; kb = 1;
}

// one work item of the launch: (tile, chunk) number `id`, or the constant's (the last one)
template <bool FAST>
__device__ __forceinline__ void mid_item(const MidArgs &g, double *sh, int tid, int id) {
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own 4 MB L2.  The (chunk, tile) list is walked
    // CHUNK-major and cut into 8 contiguous pieces, one per XCD: the 32 workgroups an XCD runs at a time work on the same rows of A and
    // on neighbouring tiles (shared 64-column panels), so a matrix larger than one L2 is still read mostly out of L2 — numbered tile-major
    // (id = tile * S + chunk) every XCD touched every row chunk of every panel and 4096 x 1024 ran out of the Infinity Cache at 1.76 us per
    // 8-row group instead of 1.15 (profiles/r06_gram_mid.txt).
    const int nbody = g.n_off - g.n_tail, wbody = nbody * g.s_off, noff = wbody + g.n_tail * g.s_tail;
    if (id < noff) {
        // body tiles (ranks 0 .. nbody - 1 of the walk) in s_off chunks, then the tail tiles in s_tail chunks: chunk-major inside each part
        const bool tail = id >= wbody;
        const int base = tail ? wbody : 0, ntile = tail ? g.n_tail : nbody, nch = tail ? g.s_tail : g.s_off;
        const int k = g.xcd ? mid_xcd_rank(id, base, ntile * nch) : ((id - base) % nch) * ntile + (id - base) / nch;
        const int chunk = k / ntile;
        const int rank = (tail ? nbody : 0) + (k - chunk * ntile);          // the tile's place in the walk: its workspace slots and its counters
        int t = rank, jb, kb;
        if (PMT_MID_SUPER && g.xcd) mid_tile_of(t, g.nb, jb, kb);
        else {
            kb = 1;
            while (t >= kb) { t -= kb; ++kb; }                     // strictly upper tiles, column by column: (0,1), (0,2), (1,2), (0,3), ..
            jb = t;
        }
        const int first = tail ? wbody + (rank - nbody) * g.s_tail : rank * g.s_off;
        mid_body<false, FAST>(g, sh, tid, jb, kb, chunk, nch, tail ? g.gpc_tail : g.gpc_off, first, g.counters + rank * MCNT);
        return;
    }
    if (id < noff + g.nb * g.s_diag) {
        const int k = g.xcd ? mid_xcd_rank(id, noff, g.nb * g.s_diag) : ((id - noff) % g.s_diag) * g.nb + (id - noff) / g.s_diag;
        const int chunk = k / g.nb, jb = k - chunk * g.nb;
        mid_body<true, FAST>(g, sh, tid, jb, jb, chunk, g.s_diag, g.gpc_diag, noff + jb * g.s_diag, g.counters + (g.n_off + jb) * MCNT);
        return;
    }
    mid_constant(g, sh, tid);
}

// A launch of more than PMT_MID_G work items runs PERSISTENT (round 6c): PMT_MID_G workgroups, each walking items until none is left,
// instead of one workgroup per item — between two workgroups on a CU lay 2.5 us (the first one's stores drain, the dispatcher starts the
// next: tools/mid_trace.py, 8 times per CU at config 2); in a loop the next item's loads go out while the stores of the last drain.  The
// items keep their XCD: a workgroup with blockIdx % 8 = x takes the ids x, x + 8, x + 16, .. in order through ticket word x (what the
// hardware's round-robin over the XCDs gives a launch of one workgroup per item).  Nobody waits for anybody; the last workgroup to leave
// re-arms the tickets.
template <bool FAST>
__global__ __launch_bounds__(256, PMT_MID_WPS) void gram_mid_kernel(MidArgs g) {
    __shared__ double sh[MSH];
    __shared__ int next_id;
    const int tid = threadIdx.x;
    const int x = blockIdx.x & 7, per = gridDim.x >> 3;
    int id = blockIdx.x;
    for (;;) {
        mid_item<FAST>(g, sh, tid, id);
        if (!g.persist) return;                                    // (one workgroup per item)
        __syncthreads();                                           // the item's last LDS reads, and next_id's last readers
        if (tid == 0) next_id = x + 8 * (per + (int)__hip_atomic_fetch_add(g.tickets + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        id = next_id;
        if (id >= g.total) break;
    }
    if (tid == 0) {
        const unsigned gone = __hip_atomic_fetch_add(g.tickets + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(g.tickets + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// c'c alone, in the same order 5 (the staged host delivery of a shape the one-launch form otherwise takes: gram.hip)
__global__ __launch_bounds__(256) void gram_mid_constant_kernel(MidArgs g) {
    __shared__ double sh[8];
    mid_constant(g, sh, threadIdx.x);
}
int launch_gram_mid_constant(const double *b, int sign, int64_t rows, double *out_const, hipStream_t s) {
    MidArgs g = {};
    g.rows = rows; g.b = (b && sign) ? b : nullptr; g.sign = g.b ? sign : 0; g.out_const = out_const;
    PMT_LAUNCH_NAMED("gram_mid_constant_kernel", gram_mid_constant_kernel, dim3(1), dim3(256), 0, s, g);
    return check_launch("gram_mid_constant_kernel");
}

size_t gram_mid_workspace_bytes(int64_t rows, int64_t cols) {
    if (cols <= 0) return 0;
    const MidPlan p = mid_plan(rows, cols);
    return sizeof(double) * (size_t)p.wgs * MSTRIDE;
}
int gram_mid_counters(int64_t cols) { const int nb = (int)cdiv(cols, MT); return MCNT * (nb * (nb + 1) / 2) + 16; }      // (+ the tickets)

// the whole node in one launch; `counters`: gram_mid_counters(cols) zeroed words owned by the calling stream (gram.hip: SideStream)
int launch_gram_mid(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign, int moi,
                    const int64_t *varmap, pmt_quadratic_term *out_quad, double *out_csc, double alpha, pmt_linear_term *out_lin,
                    double *out_const, void *workspace, unsigned *counters, hipStream_t s) {
    if (!workspace || !counters) return fail(PMT_INVALID_ARGUMENT, "quad_gram: workspace required");
    const MidPlan p = mid_plan(rows, cols);
    MidArgs g;
    g.A = A; g.lda = lda; g.rows = rows; g.cols = cols; g.b = (b && sign) ? b : nullptr; g.sign = g.b ? sign : 0;
    g.xvar = xvar; g.varmap = varmap; g.moi = moi; g.out_quad = reinterpret_cast<QT *>(out_quad); g.out_csc = out_csc; g.alpha = alpha;
    g.out_lin = reinterpret_cast<LT *>(out_lin); g.out_const = out_const;
    g.nb = p.nb; g.n_off = p.n_off; g.s_off = p.s_off; g.s_diag = p.s_diag; g.gpc_off = p.gpc_off; g.gpc_diag = p.gpc_diag;
    g.n_tail = p.n_tail; g.s_tail = p.s_tail; g.gpc_tail = p.gpc_tail;
    g.ws = reinterpret_cast<double *>(workspace); g.counters = counters;
    g.xcd = PMT_MID_XCD && rows * cols * 8 > ((int64_t)4 << 20);      // (a matrix that fits one L2 is all there on every XCD: 1024 x 512 23.0 against 25.6 us)
    const bool fast = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(g.b) & 15) == 0 &&
                      (uint64_t)lda * (uint64_t)cols * 8 < (1ull << 32);
    g.total = p.wgs;
    g.persist = PMT_MID_PERSIST && p.wgs >= 2 * PMT_MID_G;
    g.tickets = counters + gram_mid_counters(cols) - 16;
    const unsigned grid = g.persist ? (unsigned)PMT_MID_G : (unsigned)p.wgs;
    if (fast) PMT_LAUNCH_NAMED("gram_mid_kernel", (gram_mid_kernel<true>), dim3(grid), dim3(256), 0, s, g);
    else PMT_LAUNCH_NAMED("gram_mid_kernel", (gram_mid_kernel<false>), dim3(grid), dim3(256), 0, s, g);
    return check_launch("gram_mid_kernel");
}

}  // namespace pmt
