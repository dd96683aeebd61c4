// Batched small instances (BASELINE config 4: n = r = 128, m = 16): ONE persistent workgroup per CU; the whole coefficient slab of an
// instance — Q = 2 A'A (upper triangle), q = 2 A'c, c'c, the constraint block C (row-major) and 0 (+|-) d — from a single pass over
// its A, written out as ONE contiguous stream.  In the reference a batch is many independent Models (src/model.jl:1-22); what is
// replaced per instance is _vecdot!/muladd! (src/functions.jl:702-709,548-576) + canonicalize! (:381-386) + the MOI copies
// (src/moi_interop.jl:45-81), coefficients only (the index arrays are identical for every instance, batch.hip).
//
// Both bounds of the step are ~0.3 ms for 8192 instances (1.9 GB of HBM traffic; 37.7 M v_mfma_f64_4x4x4_4b at 16 cycles on 1024 SIMDs),
// so the kernel is a software pipeline in which the matrix pipe, the loads and the stores all run at the same time:
//   * A is streamed in 32-row chunks through TWO LDS panels (K-contiguous columns, odd pitch: conflict-free for the ds_read2_b64 form
//     the compiler emits).  In phase g the waves multiply chunk g out of panel g & 1, write chunk g + 1 (loaded during phase g - 1) from
//     registers into the other panel and issue the global loads of chunk g + 2: one barrier per phase, a whole phase of flight time for
//     every load, and chunks run on across instance boundaries.
//   * only the 36 of the 64 16x16 sub-tiles that touch the upper triangle are computed, dealt to the 8 waves as "four of one column
//     strip + one of another" (9 per SIMD; the rotated B-operand reads of a strip are shared; unused slots compiled out per role class).
//   * the finished slab of an instance is assembled in LDS in its final memory order (Q packed row-major upper triangle | q | c'c |
//     C row-major | d-constants: 83.6 KB) and copied to HBM as 16-byte stores spread over the k-steps of the NEXT instance's phases, so
//     the write-out never stops the matrix pipe (round 1 wrote 8 bytes per lane, row by row, with the pipe idle).
//   * q is accumulated from the LDS chunk by the vector ALU (same summation order as round 1); c'c — a serial left-to-right chain per
//     instance (src/functions.jl:574) — is computed for 64 instances at a time by the lanes of one wave while the first chunk is in flight
//     (round 1: a second kernel, 19 us).
// Q uses the MFMA lane mapping and k order of gram_sk.hip: bit-identical to pmt_quad_gram_f64 on the same instance.
#include <type_traits>

#include "common.h"

#ifndef PMT_BS_SKIP
#define PMT_BS_SKIP 0      // profiling builds only: 1 = no contraction, 2 = no q, 4 = no slab staging / copy-out, 8 = no chunk loads
#endif
#ifndef PMT_BS_PITCH
#define PMT_BS_PITCH 33    // LDS pitch of a panel column in doubles
#endif
#ifndef PMT_BS_LOADKS
#define PMT_BS_LOADKS 1    // k-steps multiplied before the phase's LDS stores / global loads are issued
#endif
#ifndef PMT_BS_COPYEVERY
#define PMT_BS_COPYEVERY 2 // one 16-byte copy-out piece per thread every this many k-steps
#endif

namespace pmt {

namespace {

constexpr int SN = 128;                 // columns handled (smaller instances are zero padded)
constexpr int CK = 32;                  // rows per chunk
constexpr int SGP = PMT_BS_PITCH;
constexpr int NT = 512;
constexpr int PANEL = SN * SGP;         // doubles per panel
constexpr int STAGE_CAP = 10464;        // slab staging capacity in doubles (config 4: 10449 + 1 alignment shift)
constexpr int NP = 4;                   // 16-byte pieces per thread per chunk
constexpr int CREG = 4;                 // constraint-block entries prefetched per thread (m*n <= 2048)

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct SmallArgs {
    const double *A; int64_t lda, rows, cols, strideA;
    const double *b; int64_t strideb; int sign;
    double *out; int64_t out_stride;    // slab of instance i at out + i*out_stride: [Q | q | const | C | d]
    int64_t B;
    const double *Cm; int64_t m; const double *d; int sign_d;   // optional constraint block (m x cols, column-major)
    int stage_all;                      // the whole slab fits the LDS staging buffer (else C and d are written directly)
};

// wave w computes sub-tiles (tm = S_TM1[w][i], tn = S_TN1[w]) for i < 4 and (S_TM2[w], S_TN2[w]); the slots a role does not need (the
// fourth group-1 tile of waves 6, 7; the group-2 tile of waves 4, 5) are compiled out of its role class
__device__ __constant__ const signed char S_TN1[8] = {7, 7, 6, 5, 4, 3, 6, 2};
__device__ __constant__ const signed char S_TM1[8][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 2, 3}, {0, 1, 2, 3}, {0, 1, 2, 3}, {0, 1, 2, 3}, {4, 5, 6, 6}, {0, 1, 2, 2}};
__device__ __constant__ const signed char S_TN2[8] = {5, 5, 1, 1, 4, 3, 4, 0};
__device__ __constant__ const signed char S_TM2[8] = {4, 5, 0, 1, 0, 0, 4, 0};

}  // namespace

// FAST: cols == 128, rows a multiple of 32, 16-byte aligned columns — aligned 16-byte loads without bounds checks.
template <bool FAST>
__global__ __launch_bounds__(NT, 2) void batch_small_kernel(SmallArgs p) {
    __shared__ __attribute__((aligned(16))) double panel[2 * PANEL];
    __shared__ __attribute__((aligned(16))) double stage[STAGE_CAP + 2];
    __shared__ double cvec[2][CK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lk = lane >> 4;
    const int kp = tid & 15, cc0 = tid >> 4;                 // chunk loads: 16-byte piece kp of columns cc0 + 32 q
    const int64_t n = p.cols, nq = n * (n + 1) / 2;
    const int nchunk = (int)max((int64_t)1, (p.rows + CK - 1) / CK);
    const int64_t G = gridDim.x;
    const int64_t mn = p.Cm ? p.m * n : 0;
    const bool creg_path = p.Cm && p.stage_all && mn <= (int64_t)CREG * NT && p.m <= NT;
    const int64_t L = (p.Cm && p.stage_all) ? nq + n + 1 + mn + p.m : nq + n + 1;      // staged (contiguous) doubles per instance

    // ---- chunk (inst, ch): global -> registers; registers -> LDS panel `buf`
    f64x2 R[NP];
    double cval = 0.0;
    auto load_chunk = [&](int64_t inst, int ch) {
        const double *A = p.A + inst * p.strideA;
        const int64_t row = (int64_t)ch * CK + 2 * kp;
        if (PMT_BS_SKIP & 8) {
#pragma unroll
            for (int q = 0; q < NP; ++q) { R[q].x = 1.0; R[q].y = 2.0; }
        } else if (FAST) {
#pragma unroll
            for (int q = 0; q < NP; ++q) R[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)(cc0 + 32 * q) * p.lda + row);
        } else {
            const int64_t rmax = max(p.rows - 1, (int64_t)0), cmax = max(p.cols - 1, (int64_t)0);
            const int64_t r0 = min(row, rmax), r1 = min(row + 1, rmax);
            const bool ok0 = row < p.rows, ok1 = row + 1 < p.rows;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int col = cc0 + 32 * q;
                const double *src = A + min((int64_t)col, cmax) * p.lda;
                const bool okc = col < p.cols && p.rows > 0;
                const double x = okc ? src[r0] : 0.0, y = okc ? src[r1] : 0.0;
                R[q].x = (okc && ok0) ? x : 0.0;
                R[q].y = (okc && ok1) ? y : 0.0;
            }
        }
        cval = 0.0;
        if (tid < CK) {
            const int64_t rr = (int64_t)ch * CK + tid;
            if (p.b && p.sign && rr < p.rows) cval = signed_const(p.b[inst * p.strideb + rr], p.sign);
        }
    };
    auto store_chunk = [&](int buf) {
        double *pan = panel + buf * PANEL;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double *d = pan + (cc0 + 32 * q) * SGP + 2 * kp;
            d[0] = R[q].x; d[1] = R[q].y;
        }
        if (tid < CK) cvec[buf][tid] = cval;
    };

    // ---- copy-out of the slab staged for the PREVIOUS instance: piece u = 16 bytes per thread; stage and HBM are co-aligned
    double *cp_out = nullptr; int cp_head = 0, cp_u = 0, cp_n = 0;
    auto copy_piece = [&]() {
        if (cp_u < cp_n) {
            const int64_t pos = cp_head + 2 * ((int64_t)tid + (int64_t)NT * cp_u);
            const double *st = stage + cp_head;
            if (pos + 1 < L) *reinterpret_cast<f64x2 *>(cp_out + pos) = *reinterpret_cast<const f64x2 *>(st + pos);
            else if (pos < L) cp_out[pos] = st[pos];
            ++cp_u;
        }
    };
    auto copy_begin = [&](int64_t inst) {
        cp_out = p.out + inst * p.out_stride;
        cp_head = (int)((reinterpret_cast<uintptr_t>(cp_out) >> 3) & 1);
        cp_u = 0;
        cp_n = (PMT_BS_SKIP & 4) ? 0 : (int)((L - cp_head + 2 * NT - 1) / (2 * NT));
        if (cp_head && tid == 0 && !(PMT_BS_SKIP & 4)) cp_out[0] = stage[cp_head];
    };

    // ---- c'c of 64 of this workgroup's instances at a time, one instance per lane of wave 7: ((0 + c_0^2) + c_1^2) + ...
    // left to right (src/functions.jl:574), loads batched eight deep
    double cst = 0.0;
    auto const_chains = [&](int64_t li0) {
        const int64_t my = (int64_t)blockIdx.x + (li0 + lane) * G;
        double s = 0.0;
        if (my < p.B && p.b && p.sign) {
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

    auto run = [&](auto has4_t, auto has2_t) {
        constexpr bool H4 = decltype(has4_t)::value, H2 = decltype(has2_t)::value;
        const int role = wave;
        double acc[5][4];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
        double qpart = 0.0;                                  // lane: column 16*wave + lm, rows lk + 4u of every chunk
        int rc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rc[r] = (((((lm >> 2) + r) & 3) << 2) | (lm & 3)) * SGP;     // column group rotated by r blocks
        int aoff[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = (S_TM1[role][i] * 16 + lm) * SGP + lk;
        aoff[4] = (S_TM2[role] * 16 + lm) * SGP + lk;
        const int b1off = (S_TN1[role] * 16) * SGP + lk, b2off = (S_TN2[role] * 16) * SGP + lk;
        const int qoff = (16 * wave + lm) * SGP + lk;

        int64_t inst = blockIdx.x, li = 0;
        int ch = 0, cur = 0;
        bool pending = false;
        double creg[CREG]; double dreg = 0.0;
#pragma unroll
        for (int u = 0; u < CREG; ++u) creg[u] = 0.0;

        auto next_of = [&](int64_t i, int c, int64_t &ni, int &nc) { nc = c + 1; ni = i; if (nc == nchunk) { nc = 0; ni = i + G; } };

        // prime the pipeline: chunk 0 -> panel 0, chunk 1 in registers
        if (inst < p.B) {
            load_chunk(inst, 0);
            if (wave == 7) const_chains(0);
            store_chunk(0);
            int64_t ni; int nc;
            next_of(inst, 0, ni, nc);
            if (ni < p.B) load_chunk(ni, nc);
        }
        __syncthreads();

        while (inst < p.B) {
            const double *pan = panel + cur * PANEL;
            int64_t n1i, n2i; int n1c, n2c;
            next_of(inst, ch, n1i, n1c);
            next_of(n1i, n1c, n2i, n2c);
            if (n2i >= p.B) { n2i = inst; n2c = ch; }         // past the end: re-load a valid chunk, never used (no branch around the loads)
            const bool last = (ch == nchunk - 1);

            auto ksteps = [&](int k0, int k1, bool copy) {
#pragma unroll
                for (int ks = k0; ks < k1; ++ks) {
                    if (!(PMT_BS_SKIP & 1)) {
                        double a[5], b1[4], b2[4];
#pragma unroll
                        for (int i = 0; i < 3; ++i) a[i] = pan[aoff[i] + ks * 4];
                        if (H4) a[3] = pan[aoff[3] + ks * 4];
                        if (H2) a[4] = pan[aoff[4] + ks * 4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { b1[r] = pan[b1off + rc[r] + ks * 4]; if (H2) b2[r] = pan[b2off + rc[r] + ks * 4]; }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int i = 0; i < (H4 ? 4 : 3); ++i)
                                acc[i][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b1[r], acc[i][r], 0, 0, 0);
                            if (H2) acc[4][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[4], b2[r], acc[4][r], 0, 0, 0);
                        }
                    }
                    if (copy && (ks % PMT_BS_COPYEVERY) == 0) copy_piece();
                }
            };

            // the copy-out of the previous instance is spread over the phases before this instance's last one
            const bool copy_here = pending && !last;
            ksteps(0, PMT_BS_LOADKS, copy_here);
            if (!(PMT_BS_SKIP & 2)) {                         // q from the chunk (vector ALU, beside the partner wave's MFMAs)
#pragma unroll
                for (int u = 0; u < CK / 4; ++u) { const double pr = cvec[cur][lk + 4 * u] * pan[qoff + 4 * u]; qpart = qpart + pr; }
            }
            store_chunk(cur ^ 1);                             // chunk g + 1 (registers) -> the other panel
            load_chunk(n2i, n2c);                             // chunk g + 2 -> registers
            if (last && creg_path) {                          // constraint block of this instance: in flight during the phase
                const double *Ci = p.Cm + inst * mn;
#pragma unroll
                for (int u = 0; u < CREG; ++u) { const int64_t e = tid + (int64_t)NT * u; creg[u] = e < mn ? Ci[e] : 0.0; }
                dreg = tid < p.m ? p.d[inst * p.m + tid] : 0.0;
            }
            ksteps(PMT_BS_LOADKS, CK / 4, copy_here);
            if (pending && (ch == nchunk - 2 || nchunk == 1)) {     // whatever is left must be out before anyone restages
                while (cp_u < cp_n) copy_piece();
                pending = false;
                if (nchunk == 1) __syncthreads();
            }

            if (last) {
                // ---- the slab of this instance, in its final order, into the staging buffer
                double *outp = p.out + inst * p.out_stride;
                const int head = (int)((reinterpret_cast<uintptr_t>(outp) >> 3) & 1);
                double *st = stage + head;
                if (!(PMT_BS_SKIP & 4)) {
                    const int i_ = lane >> 4, bq = (lane >> 2) & 3, j_ = lane & 3;   // accumulator element -> (row, col) inside a sub-tile
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        if ((i == 3 && !H4) || (i == 4 && !H2)) continue;
                        const int tm = i < 4 ? S_TM1[role][i] : S_TM2[role];
                        const int tn = i < 4 ? S_TN1[role] : S_TN2[role];
                        const int row = tm * 16 + 4 * bq + i_;
                        const int rbase = row * (int)n - (row * (row - 1)) / 2 - row;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = tn * 16 + 4 * ((bq + r) & 3) + j_;
                            if (row <= col && col < n) st[rbase + col] = 2 * acc[i][r];
                        }
                    }
                    // q: the four row classes of a column, added in class order, x2
                    const double p1 = __shfl(qpart, lane + 16, 64), p2 = __shfl(qpart, lane + 32, 64), p3 = __shfl(qpart, lane + 48, 64);
                    if (lk == 0 && 16 * wave + lm < n) st[nq + 16 * wave + lm] = 2 * (((qpart + p1) + p2) + p3);
                    if (wave == 7 && lane == (int)(li & 63)) st[nq + n] = cst;
                    if (p.Cm) {
                        double *sc = p.stage_all ? st + nq + n + 1 : outp + nq + n + 1;       // staged, or straight to HBM when too large
                        if (creg_path) {
#pragma unroll
                            for (int u = 0; u < CREG; ++u) {
                                const int64_t e = tid + (int64_t)NT * u;
                                if (e < mn) { const int64_t col = e / p.m, row = e - col * p.m; sc[row * n + col] = creg[u]; }
                            }
                            if (tid < p.m) sc[mn + tid] = signed_const(dreg, p.sign_d);
                        } else {
                            const double *Ci = p.Cm + inst * mn;
                            for (int64_t e = tid; e < mn; e += NT) { const int64_t col = e / p.m, row = e - col * p.m; sc[row * n + col] = Ci[e]; }
                            for (int64_t i = tid; i < p.m; i += NT) sc[mn + i] = signed_const(p.d[inst * p.m + i], p.sign_d);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
                qpart = 0.0;
                ++li;
            }
            __syncthreads();
            if (last) {
                copy_begin(inst);
                pending = true;
                if ((li & 63) == 0 && wave == 7) const_chains(li);          // more than 64 instances per workgroup: next batch of chains
            }
            cur ^= 1;
            inst = n1i; ch = n1c;
        }
        if (pending) { while (cp_u < cp_n) copy_piece(); }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (wave >= 6) run(F{}, T{});
    else if (wave >= 4) run(T{}, F{});
    else run(T{}, T{});
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
    if (fast) PMT_LAUNCH_NAMED("batch_small_kernel", batch_small_kernel<true>, grid, dim3(NT), 0, s, p);
    else PMT_LAUNCH_NAMED("batch_small_kernel", batch_small_kernel<false>, grid, dim3(NT), 0, s, p);
    return check_launch("batch_small_kernel");
}

}  // namespace pmt
