// Canonical least-squares objective, SPECIALISED-WAVE form of the stream-K contraction (gram_sk.hip): the fast path for aligned shapes
// (whole 128-column tiles, 16-byte aligned columns, a multiple of 16 rows); everything else takes gram_sk.hip.  Same work split (phase A:
// whole tiles per workgroup in the XCD-aware order; phase B: stream-K units of the remainder tiles + gram_sk_fixup_kernel), same MFMA
// lane map and k order, hence the same bits.  What changes is WHO does what inside the persistent workgroup of 12 waves (3 per SIMD):
//   * waves 0-7   matrix waves (2 per SIMD, 64x32 wave tiles): LDS operand reads and MFMAs, nothing else.  In gram_sk.hip the same waves
//                 also issue the global loads, the LDS stores and their address arithmetic, and wait on them.
//   * waves 8, 9  loader waves: the K-contiguous column panels of stage g + 2 go global -> registers (two register sets, inline-asm loads
//                 with a hand-counted s_waitcnt: these waves issue nothing but loads) and stage g + 1 registers -> LDS, one barrier per stage.
//   * waves 10,11 formatter waves: the tile EPILOGUE.  When a tile's contraction ends the matrix waves only dump their accumulators into
//                 a per-workgroup scratch slot (coalesced 8-byte stores, ~3 us) and start the next tile; the formatters turn the slot into
//                 MOI.ScalarQuadraticTerms (x2, canonical upper-triangular position, varmap; 16-byte chunk stores) or CSC values WHILE the
//                 next tile is being multiplied.  In gram_sk.hip the epilogue stops the matrix pipe for ~40 us per tile (the store path of
//                 a CU takes that long for 393 KB), ~7 % of the launch at n = r = 4096.
// The lesson is config 4's (batch_small.hip, profiles/r02_batch_small.txt): a wave that waits for the vector-memory path should not be
// the wave that feeds the matrix pipe.
#include <type_traits>

#include "gram_common.h"

#ifndef PMT_G2_SKIP
#define PMT_G2_SKIP 0      // profiling builds only: 1 = no formatting (results missing), 2 = no loads
#endif

namespace pmt {

int launch_gram_fixup(const SKArgs &g, int64_t R, hipStream_t s);      // gram_sk.hip

namespace {

constexpr int BK = 16;                 // contraction depth per LDS stage
constexpr int GP = BK + 2;             // LDS pitch of a panel column: 18 = 2 (mod 4)... (lm * 18 + lk) mod 32 is injective over a half wave
constexpr int PANEL = ST * GP;         // doubles per panel
constexpr int NMAT = 512, NLOAD = 128, NFMT = 128, NTHREADS = NMAT + NLOAD + NFMT;
constexpr int NPL = (2 * ST * (BK / 2)) / NLOAD;      // 16-byte pieces per loader thread per stage (both panels): 16
constexpr int NACC = 32;

struct Job {
    int jb, kb;
    int64_t ibeg, iend;
    int full;          // 1: whole contraction -> formatted by this workgroup; 0: partial -> workspace slot + fix-up kernel
    int pslot;         // partial: workspace slot
    int valid;
};

// the idx-th job of workgroup `bid`: phase A whole tiles first, then the workgroup's stream-K range of (tile, chunk) units
__device__ __forceinline__ Job g2_job(const SKArgs &g, int bid, int idx) {
    Job j;
    j.valid = 0; j.jb = j.kb = 0; j.ibeg = j.iend = 0; j.full = 0; j.pslot = 0;
    if (idx < g.tfull) {
        sk_seq_unrank(sk_phase_a_index(g, bid, idx), g.ntiles, j.jb, j.kb);
        j.ibeg = 0; j.iend = g.rows; j.full = 1; j.valid = 1;
        return j;
    }
    const int64_t u0 = sk_unit_begin(g, bid), u1 = sk_unit_begin(g, bid + 1);
    int k = idx - g.tfull;
    for (int64_t u = u0; u < u1;) {
        const int rtile = (int)(u / g.nchunk);
        const int c0 = (int)(u - (int64_t)rtile * g.nchunk);
        const int c1 = (int)min((int64_t)g.nchunk, (int64_t)c0 + (u1 - u));
        if (k == 0) {
            sk_seq_unrank(g.tfull * g.G + rtile, g.ntiles, j.jb, j.kb);
            j.ibeg = (int64_t)c0 * SKC; j.iend = min(g.rows, (int64_t)c1 * SKC);
            j.full = (c0 == 0 && c1 == g.nchunk) ? 1 : 0;
            j.pslot = 2 * bid + (u == u0 ? 0 : 1);
            j.valid = 1;
            return j;
        }
        --k;
        u += (c1 - c0);
    }
    return j;
}

__device__ __forceinline__ int g2_stages(const Job &j) { return (int)((j.iend - j.ibeg) / BK); }

__device__ __forceinline__ double *g2_scratch(const SKArgs &g, int bid, int jobidx) {
    // scratch slots for whole tiles live behind the 2 * MAXG partial slots of the fix-up protocol: two per workgroup, alternating
    return g.ws + ((int64_t)2 * MAXG + 2 * bid + (jobidx & 1)) * SLOT;
}

// ---- matrix waves ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void g2_matrix(const SKArgs &g, const double *lds, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int lm = lane & 15, lk = lane >> 4;
    const int bid = blockIdx.x;
    double acc[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.0;
    int rc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rc[r] = (((((lm >> 2) + r) & 3) << 2) | (lm & 3)) * GP;          // column group rotated by r blocks
    const int aoff = (wr * 64 + lm) * GP + lk;                  // J panel: + tm * 16 * GP
    const int boff = PANEL + (wc * 32) * GP + lk;               // K panel: + tn * 16 * GP + rc[r]
    int gstage = 0;
    double a[2][4], b[2][2][4];
    auto read_operands = [&](int set, const double *pan, int ks) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) a[set][tm] = pan[aoff + tm * 16 * GP + ks * 4];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) b[set][tn][r] = pan[boff + tn * 16 * GP + rc[r] + ks * 4];
    };
    __syncthreads();                                            // buffer 0 holds the first stage
    read_operands(0, lds, 0);
    for (int jobidx = 0;; ++jobidx) {
        const Job job = g2_job(g, bid, jobidx);
        if (!job.valid) break;
        const int nst = g2_stages(job);
        for (int s = 0; s < nst; ++s, ++gstage) {
            const double *pan = lds + (gstage & 1) * 2 * PANEL, *pan_next = lds + ((gstage + 1) & 1) * 2 * PANEL;
            // One continuous software pipeline over the k-steps of ALL stages: the operands of the next k-step are read before the MFMAs of
            // the current one are issued — across the stage boundary too: the barrier sits between "my reads of this buffer are complete" and
            // "the first reads of the next buffer", and the last k-step's 32 MFMAs are issued BEHIND it, so the matrix pipe has work while the
            // waves meet at the barrier and the first operands of the next stage are in flight.  (With the barrier after the MFMAs the pipe
            // drained at every stage: 1.30 ms instead of 1.18 for the unspecialised kernel.)
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int set = ks & 1;                          // BK / 4 is even: the parity runs on across stages
                if (ks + 1 < BK / 4) {
                    read_operands(set ^ 1, pan, ks + 1);
                } else if (s + 1 < nst) {
                    __syncthreads();                            // (waits for this wave's outstanding LDS reads first)
                    read_operands(set ^ 1, pan_next, 0);
                    __builtin_amdgcn_sched_barrier(0);          // the MFMAs below stay below: MFMAs are not memory operations, and the
                }                                               // scheduler would otherwise lift them back above the barrier
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int tm = 0; tm < 4; ++tm)
                            acc[(tm * 2 + tn) * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[set][tm], b[set][tn][r], acc[(tm * 2 + tn) * 4 + r], 0, 0, 0);
                asm volatile("" ::: "memory");                 // operand reads are hoisted one k-step ahead, not further
            }
            if (s == nst - 1) {
                // the tile (or partial tile) leaves the registers: [accumulator index][thread], coalesced — the layout the fix-up kernel and
                // the formatter waves both know (sk_acc_pos)
                double *w = (job.full ? g2_scratch(g, bid, jobidx) : g.ws + (int64_t)job.pslot * SLOT) + tid;
#pragma unroll
                for (int r = 0; r < NACC; ++r) { w[r * NMAT] = acc[r]; acc[r] = 0.0; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // in L2 before the barrier lets the formatters at it
                __syncthreads();
                read_operands(0, pan_next, 0);                  // first operands of the next job (unused after the last one)
            }
        }
    }
}

// ---- loader waves ------------------------------------------------------------------------------------------------------------------
struct StageCursor { int jobidx; int s; Job job; };

__device__ __forceinline__ void g2_loader(const SKArgs &g, double *lds, int lt) {
    const int bid = blockIdx.x;
    const int kp = lt & 7, cc = lt >> 3;                        // 16-byte piece kp of columns cc + 16 q of a panel
    const unsigned voff = (unsigned)(((int64_t)cc * g.lda + 2 * kp) * 8);
    const int loff = cc * GP + 2 * kp;
    f64x2 R[2][NPL] = {};

    auto issue = [&](auto set_t, const StageCursor &c) {
        constexpr int S = decltype(set_t)::value;
        if (PMT_G2_SKIP & 2) return;
        const int64_t i0 = c.job.ibeg + (int64_t)c.s * BK;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t c0 = (int64_t)(p == 0 ? c.job.jb : c.job.kb) * ST;
#pragma unroll
            for (int q = 0; q < NPL / 2; ++q) {
                const double *base = g.A + (c0 + 16 * q) * g.lda + i0;            // wave-uniform: an SGPR pair
                f64x2 &dst = R[S][p * (NPL / 2) + q];
                const unsigned off = voff;                                         // (asm operands inside a generic lambda must name locals)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base) : "memory");
            }
        }
    };
    auto wait_set = [&](auto set_t, auto younger_t) {
        constexpr int S = decltype(set_t)::value;
        constexpr int YOUNGER = decltype(younger_t)::value;
        if (PMT_G2_SKIP & 2) return;
        f64x2 (&r)[NPL] = R[S];
        asm volatile("s_waitcnt vmcnt(%16)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                     "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "n"(YOUNGER) : "memory");
    };
    auto store = [&](auto set_t, int buf) {
        constexpr int S = decltype(set_t)::value;
        double *pan = lds + buf * 2 * PANEL;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < NPL / 2; ++q) *reinterpret_cast<f64x2 *>(pan + p * PANEL + loff + 16 * q * GP) = R[S][p * (NPL / 2) + q];
    };
    auto advance = [&](StageCursor &c) {                        // the next stage of the flat stream; past the end: stay (a valid, unused re-load)
        if (c.s + 1 < g2_stages(c.job)) { ++c.s; return; }
        const Job nj = g2_job(g, bid, c.jobidx + 1);
        if (nj.valid) { c.job = nj; ++c.jobidx; c.s = 0; }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using None = std::integral_constant<int, 0>;
    using OneSet = std::integral_constant<int, NPL>;

    StageCursor cur;                                             // the stage the matrix waves work on
    cur.jobidx = 0; cur.s = 0; cur.job = g2_job(g, bid, 0);
    if (!cur.job.valid) { __syncthreads(); return; }
    StageCursor pre = cur;                                       // the stage being loaded
    issue(S0{}, pre);
    wait_set(S0{}, None{});
    store(S0{}, 0);
    advance(pre);
    issue(S1{}, pre);                                            // stage 1 -> set 1
    __syncthreads();
    // phase g (parity PAR): loads of stage g + 2 -> set PAR (stage g left it a phase ago); stage g + 1 (set PAR ^ 1) -> LDS buffer PAR ^ 1
    auto phase = [&](auto par_t) {
        constexpr int PAR = decltype(par_t)::value;
        using Other = std::integral_constant<int, PAR ^ 1>;
        advance(pre);
        issue(par_t, pre);
        wait_set(Other{}, OneSet{});
        store(Other{}, PAR ^ 1);
        __syncthreads();
        // advance `cur`; returns false after the last stage
        if (cur.s + 1 < g2_stages(cur.job)) { ++cur.s; return true; }
        const Job nj = g2_job(g, bid, cur.jobidx + 1);
        if (!nj.valid) return false;
        cur.job = nj; ++cur.jobidx; cur.s = 0;
        return true;
    };
    for (;;) {
        if (!phase(S0{})) break;
        if (!phase(S1{})) break;
    }
    wait_set(S0{}, None{});                                      // the clamped re-loads of the last phases are still in flight into the sets
    wait_set(S1{}, None{});
}

// ---- formatter waves ---------------------------------------------------------------------------------------------------------------
// value of tile element (row, col) in a slot written by the matrix waves: the inverse of sk_acc_pos<2>
__device__ __forceinline__ int g2_slot_index(int row, int col) {
    const int wr = row >> 6, tm = (row & 63) >> 4, rr = row & 15, b = rr >> 2, i = rr & 3;
    const int wc = col >> 5, tn = (col & 31) >> 4, cq = col & 15, cg = cq >> 2, j = cq & 3;
    const int s = (cg - b) & 3;
    const int lane = j + 4 * b + 16 * i, wave = wr * 4 + wc;
    return ((tm * 2 + tn) * 4 + s) * NMAT + wave * 64 + lane;
}

__device__ __forceinline__ double g2_load_l2(const double *p) {
    // written by other waves of this workgroup a moment ago: read from L2, never from this CU's L1 (a slot is reused every other tile)
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct FmtState {
    int active;        // a tile is waiting to be formatted
    int jb, kb;
    const double *slot;
    int next_row, next_col;      // progress of the row passes (out_quad) / column passes (out_csc) of this wave
    int cmap_ready;
};

// one pass = one row of the tile as QuadraticTerms, or one column as CSC values; returns false when the tile is done
__device__ __forceinline__ bool g2_format_pass(const SKArgs &g, FmtState &f, u64 *s_val, u64 *s_cmap, int fw, int lane) {
    const int64_t n = g.cols, j0 = (int64_t)f.jb * ST, k0 = (int64_t)f.kb * ST;
    if (g.out_quad && f.next_row < ST) {
        if (!f.cmap_ready) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t kv = g.xvar[k0 + lane + 64 * h];
                s_cmap[lane + 64 * h] = (u64)(g.moi ? map_var(g.varmap, kv) : kv);
            }
            f.cmap_ready = 1;
            __builtin_amdgcn_wave_barrier();
        }
        const int jr = f.next_row;
        f.next_row += NFMT / 64;
        const int64_t j = j0 + jr;
        const int64_t kstart = j > k0 ? j : k0;
        const int nterms = (int)(k0 + ST - kstart);
        if (nterms <= 0) return true;
        const int coff = (int)(kstart - k0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = lane + 64 * h;
            double c = g2_load_l2(f.slot + g2_slot_index(jr, col));
            if (g.moi || j != k0 + col) c = 2 * c;             // off-diagonal: (j,k)+(k,j) combined; diagonal: MOI doubling (moi_interop.jl:58)
            s_val[col] = (u64)__double_as_longlong(c);
        }
        __builtin_amdgcn_wave_barrier();
        const int64_t jv = g.xvar[j];
        const u64 rv = (u64)(g.moi ? map_var(g.varmap, jv) : jv);
        const int64_t term0 = j * n - (j * (j - 1)) / 2 + (kstart - j);
        wave_write_words<3>(reinterpret_cast<u64 *>(g.out_quad) + term0 * 3, nterms, lane, [&](int q) -> u64 {
            const int t = q / 3, fld = q - 3 * t;
            return fld == 0 ? s_val[coff + t] : (fld == 1 ? rv : s_cmap[coff + t]);
        });
        __builtin_amdgcn_wave_barrier();
        return true;
    }
    if (g.out_csc && f.next_col < ST) {
        const int kc = f.next_col;
        f.next_col += NFMT / 64;
        const int64_t k = k0 + kc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int jr = lane + 64 * h;
            const int64_t j = j0 + jr;
            if (j <= k) {
                double c = g2_load_l2(f.slot + g2_slot_index(jr, kc));
                if (g.moi || j != k) c = 2 * c;
                g.out_csc[k * (k + 1) / 2 + j] = g.alpha * c;
            }
        }
        return true;
    }
    return false;
}

__device__ __forceinline__ void g2_formatter(const SKArgs &g, double *lds, int ft) {
    const int bid = blockIdx.x;
    const int lane = ft & 63, fw = ft >> 6;
    u64 *s_val = reinterpret_cast<u64 *>(lds + 4 * PANEL) + fw * 256;             // wave-local staging behind the panels
    u64 *s_cmap = s_val + 128;
    FmtState f;
    f.active = 0; f.jb = f.kb = 0; f.slot = nullptr; f.next_row = f.next_col = ST; f.cmap_ready = 0;
    const int passes_per_tile = (g.out_quad ? ST / (NFMT / 64) : 0) + (g.out_csc ? ST / (NFMT / 64) : 0);
    __syncthreads();
    Job prev; prev.valid = 0; prev.full = 0; prev.jb = prev.kb = 0;
    for (int jobidx = 0;; ++jobidx) {
        const Job job = g2_job(g, bid, jobidx);
        if (jobidx > 0 && prev.valid && prev.full && !(PMT_G2_SKIP & 1)) {           // the tile of the previous job is in its slot (dumped before the
            f.active = 1; f.jb = prev.jb; f.kb = prev.kb;                             // barrier that ended its last stage)
            f.slot = g2_scratch(g, bid, jobidx - 1);
            f.next_row = fw; f.next_col = fw; f.cmap_ready = 0;
        }
        if (!job.valid) break;
        const int nst = g2_stages(job);
        int left = f.active ? passes_per_tile : 0;
        for (int s = 0; s < nst; ++s) {
            if (f.active) {                                       // spread over this job's stages; all of it is out before the job's last barrier,
                int todo = (left + (nst - s) - 1) / (nst - s);    // i.e. before the matrix waves dump into the other slot... and this one a job later
                for (; todo > 0 && f.active; --todo, --left)
                    if (!g2_format_pass(g, f, s_val, s_cmap, fw, lane)) f.active = 0;
            }
            __syncthreads();
        }
        while (f.active) { if (!g2_format_pass(g, f, s_val, s_cmap, fw, lane)) f.active = 0; }
        prev = job;
    }
    while (f.active) { if (!g2_format_pass(g, f, s_val, s_cmap, fw, lane)) f.active = 0; }   // the last tile
}

}  // namespace

__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_num_vgpr(160))) void gram2_kernel(SKArgs g) {
    __shared__ __attribute__((aligned(16))) double lds[4 * PANEL + 2 * 256];
    const int tid = threadIdx.x;
    if (tid < NMAT) g2_matrix(g, lds, tid);
    else if (tid < NMAT + NLOAD) g2_loader(g, lds, tid - NMAT);
    else g2_formatter(g, lds, tid - NMAT - NLOAD);
}

size_t gram2_workspace_bytes() { return (size_t)(2 * MAXG + 2 * MAXG) * SLOT * sizeof(double); }

// returns PMT_OK and sets *taken when the shape qualifies; otherwise leaves *taken false (the caller runs gram_sk.hip)
int launch_gram2(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const int64_t *varmap, int moi,
                 pmt_quadratic_term *out_quad, double *out_csc, double alpha, void *workspace, hipStream_t s, bool *taken) {
    *taken = false;
    const bool aligned = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0;
    if (!aligned || cols <= 0 || (cols % ST) != 0 || rows < BK || (rows % BK) != 0 || !workspace || lda * ST >= ((int64_t)1 << 28)) return PMT_OK;
    SKArgs g;
    g.A = A; g.lda = lda; g.rows = rows; g.cols = cols; g.xvar = xvar; g.varmap = varmap; g.moi = moi; g.out_quad = out_quad;
    g.out_csc = out_csc; g.alpha = alpha;
    g.ntiles = (int)cdiv(cols, ST);
    g.nchunk = (int)std::max<int64_t>(1, cdiv(rows, SKC));
    const int64_t T = (int64_t)g.ntiles * (g.ntiles + 1) / 2;
    g.G = (int)std::min<int64_t>(T * g.nchunk, 256);
    g.tfull = (int)(T / g.G);
    const int64_t R = T - (int64_t)g.tfull * g.G;
    g.U = R * g.nchunk;
    g.vec_in = 1;
    g.ws = reinterpret_cast<double *>(workspace);
    PMT_LAUNCH_NAMED("gram2_kernel", gram2_kernel, dim3((unsigned)g.G), dim3(NTHREADS), 0, s, g);
    int rc = check_launch("gram2_kernel");
    if (rc) return rc;
    *taken = true;
    if (g.nchunk > 1 && R > 0) rc = launch_gram_fixup(g, R, s);
    return rc;
}

}  // namespace pmt
