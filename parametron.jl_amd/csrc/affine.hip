// Dense affine nodes: A*x (+|-) b  ->  Vector{AffineFunction} term block (LinearTerm AoS) or, fused with
// the MOI copy, MOI.VectorAffineTerm AoS.  HBM-bound streaming transpose (column-major in, row-major out).
//
// Reference loops replaced (see include/parametron_hip.h):
//   matvecmul!    src/functions.jl:775-798      vecadd!/vecsubtract!  src/functions.jl:751-764
//   update!(::MOI.VectorAffineFunction, ...)    src/moi_interop.jl:64-81
//
// Kernel shape (gfx950): 64x64 tile per 256-thread workgroup.  Loads: each half-wave reads one column
// segment of 64 rows as 32 x 16 B (512 B contiguous).  The tile is transposed through LDS (pitch 65
// doubles) and written row-major; every wave store instruction covers one contiguous 1 KiB (LinearTerm)
// or 1 KiB-chunks of the 1.5 KiB row segment (24-byte VectorAffineTerm, assembled as 16-byte chunks so
// that all stores are global_store_dwordx4).  Algorithmic bytes: 8 read + 16 (LT) / 24 (VAT) written
// per matrix entry.
#include "common.h"

namespace pmt {

// NTL (nontemporal loads of the matrix): measured (profiles/r04_batch_small.txt section 3, r04_affine_pack_nt.txt) — a matrix that is still in
// the Infinity Cache is read 16 % slower with the policy, a cold one faster.  The MOI pack of a LARGE constraint block runs behind the
// objective's contraction, which has streamed ~1 GB through the cache since the block was last touched: it is always cold there, and the
// policy takes the in-step launch of config 2's 512 x 4096 block from 15.4 to ~13.8 us.  Small blocks and the LinearTerm form keep plain loads.
constexpr int64_t NTL_MIN_BYTES = 8 << 20;

constexpr int TILE = 64;
constexpr int PITCH = TILE + 1;

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 f2u(double x) { return (u64)__double_as_longlong(x); }

template <bool NT>
__device__ __forceinline__ void store16(u64x2 *p, u64x2 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT>
__device__ __forceinline__ void store8(u64 *p, u64 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: LinearTerm output   MODE 1: VectorAffineTerm output
// TR: rows per tile (64, or 32 for blocks that would otherwise give fewer than ~4 workgroups per CU: the 512 x 4096 constraint block
// of config 2 is 512 tiles of 64 x 64 — two per CU, 8 waves, tail-dominated: 19 us inside the step against 13 us when it had the
// chip to itself; with 32-row tiles 1024 workgroups of half the LDS)
template <int MODE, bool NT, int TR, bool NTL = false>
__global__ __launch_bounds__(256) void affine_tile_kernel(
    const double *__restrict__ A, int64_t lda, int64_t rows, int64_t cols,
    const int64_t *__restrict__ xvar, const double *__restrict__ b, int sign,
    const int64_t *__restrict__ varmap, int64_t row_offset,
    u64 *__restrict__ out, double *__restrict__ out_consts, int vec_in, int vec_out, u64 *stamps, unsigned stamp_cap) {
    __shared__ double tile[TR * PITCH];
    __shared__ u64 vmx[TILE];

    const int t = threadIdx.x;
    // measurement hook (pmt_profile_kernel_stamps; null in production): this workgroup's start on the device's constant-rate clock, into
    // its own slot (plain stores — atomics of a thousand workgroups on one word serialise and triple the launch)
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (stamps && t == 0 && wg < stamp_cap) stamps[2 * wg] = (u64)wall_clock64();
    const int lane = t & 63;
    const int wave = t >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * TILE;
    const int64_t r0 = (int64_t)blockIdx.y * TR;
    const int nr = (int)min((int64_t)TR, rows - r0);
    const int nc = (int)min((int64_t)TILE, cols - c0);
    const bool full = (nr == TR) && (nc == TILE);

    // ---- load phase: column-major A tile -> LDS tile[row][col]
    if (full && vec_in) {
        constexpr int TPC = TR / 2;        // threads per column (16-byte pieces of a column segment)
        constexpr int CPI = 256 / TPC;     // columns per iteration
        const int cg = t / TPC;            // column within the group
        const int lr = (t % TPC) * 2;      // row pair
        const double *base = A + (c0 + cg) * lda + r0 + lr;
#pragma unroll
        for (int it = 0; it < TILE / CPI; ++it) {
            const f64x2 *src = reinterpret_cast<const f64x2 *>(base + (int64_t)it * CPI * lda);
            f64x2 v = NTL ? __builtin_nontemporal_load(src) : *src;
            const int c = it * CPI + cg;
            tile[lr * PITCH + c] = v.x;
            tile[(lr + 1) * PITCH + c] = v.y;
        }
    } else {
        const int r = t & 63;
        for (int c = t >> 6; c < nc; c += 4)
            if (r < nr) tile[r * PITCH + c] = A[(c0 + c) * lda + r0 + r];
    }
    if (t < nc) {
        const int64_t v = xvar[c0 + t];
        vmx[t] = (u64)(MODE == 1 ? map_var(varmap, v) : v);
    }
    // constants: one column of blocks writes 0.0 (+|-) b[row]
    if (blockIdx.x == 0 && t < nr && out_consts)
        out_consts[r0 + t] = signed_const(b ? b[r0 + t] : 0.0, b ? sign : 0);
    __syncthreads();

    // ---- store phase
    if (MODE == 0) {
        // 16 B per term: one wave store = 64 terms = 1 KiB contiguous
        if (lane < nc) {
            const u64 var = vmx[lane];
            for (int r = wave; r < nr; r += 4) {
                u64x2 v;
                v.x = f2u(tile[r * PITCH + lane]);
                v.y = var;
                store16<NT>(reinterpret_cast<u64x2 *>(out + ((r0 + r) * cols + c0 + lane) * 2), v);
            }
        }
    } else {
        if (full && vec_out) {
            // rows in pairs: 3 full-wave 16-byte stores per pair (row segment = 192 qwords = 96 chunks)
            for (int rp = wave * 2; rp < TR; rp += 8) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    int r, chunk;
                    if (s == 0) { r = rp; chunk = lane; }
                    else if (s == 1) { r = rp + (lane >> 5); chunk = 64 + (lane & 31); }
                    else { r = rp + 1; chunk = lane; }
                    const int q0 = chunk * 2;
                    const u64 rowidx = (u64)(row_offset + r0 + r + 1);
                    u64 w[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = q0 + h;
                        const int term = q / 3;
                        const int f = q - term * 3;
                        w[h] = (f == 0) ? rowidx : (f == 1 ? f2u(tile[r * PITCH + term]) : vmx[term]);
                    }
                    u64x2 v; v.x = w[0]; v.y = w[1];
                    store16<NT>(reinterpret_cast<u64x2 *>(out + ((r0 + r) * cols + c0) * 3 + q0), v);
                }
            }
        } else {
            if (lane < nc) {
                const u64 var = vmx[lane];
                for (int r = wave; r < nr; r += 4) {
                    u64 *p = out + ((r0 + r) * cols + c0 + lane) * 3;
                    store8<NT>(p, (u64)(row_offset + r0 + r + 1));
                    store8<NT>(p + 1, f2u(tile[r * PITCH + lane]));
                    store8<NT>(p + 2, var);
                }
            }
        }
    }
    if (stamps) {                    // ... and the latest workgroup end, behind this workgroup's stores
        __syncthreads();
        if (t == 0 && wg < stamp_cap) {
            __builtin_amdgcn_s_waitcnt(0);
            stamps[2 * wg + 1] = (u64)wall_clock64();
        }
    }
}

// x (+|-) v for x::Vector{Variable}
__global__ void vars_addsub_kernel(const int64_t *__restrict__ xvar, int64_t n, const double *__restrict__ v, int sign,
                                   const int64_t *__restrict__ varmap, int64_t row_offset,
                                   LT *__restrict__ out_lt, VAT *__restrict__ out_vat, double *__restrict__ out_consts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t var = xvar[i];
    if (out_lt) { LT t; t.coeff = 1.0; t.var = var; out_lt[i] = t; }
    if (out_vat) { VAT t; t.output_index = row_offset + i + 1; t.coeff = 1.0; t.var = map_var(varmap, var); out_vat[i] = t; }
    if (out_consts) out_consts[i] = signed_const(v ? v[i] : 0.0, v ? sign : 0);
}

__global__ void consts_kernel(const double *__restrict__ d, int64_t n, int sign, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = signed_const(d[i], sign);
}

// "Background" form of the MOI constraint pack (same output as affine_tile_kernel<VAT>): at most 16 VGPRs and no LDS, so that its
// waves are CO-RESIDENT with the persistent Gram kernel, which leaves 16 of a SIMD's 512 VGPRs free (gram.hip) — on the plan's side
// lane it then runs inside the contraction instead of behind it.  One wave per (row, 256-column chunk); lanes read one element of
// 64 different columns (8-byte reads, one cache line each — the 8 rows that share those lines are handled by the next 7 waves, so
// they come from L2) and write their 24-byte term.  Slow on its own (it is not the kernel to use on an idle chip), free where it runs.
__global__ __launch_bounds__(256) void affine_pack_background_kernel(const double *__restrict__ A, int64_t lda, int rows, int cols,
                                                                     const int64_t *__restrict__ xvar, const int64_t *__restrict__ varmap,
                                                                     int64_t row_offset, unsigned long long *__restrict__ out) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int chunk = wave / rows, row = wave - chunk * rows;
    const int c0 = chunk * 256 + lane;
    const double *a = A + row;
    unsigned long long *o = out + (int64_t)row * cols * 3;
    const unsigned long long rowword = (unsigned long long)(row_offset + row + 1);
#pragma unroll 1
    for (int c = c0; c < min(cols, chunk * 256 + 256); c += 64) {
        const double v = a[(int64_t)c * lda];
        const int64_t var = map_var(varmap, xvar[c]);
        unsigned long long *p = o + (int64_t)c * 3;
        p[0] = rowword;
        p[1] = (unsigned long long)__double_as_longlong(v);
        p[2] = (unsigned long long)var;
    }
}

// ---- MOI copies of materialised native functions (src/moi_interop.jl:35-81)
__global__ void pack_scalar_affine_kernel(const LT *__restrict__ in, int64_t n, const int64_t *__restrict__ varmap, LT *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LT t = in[i];
    t.var = map_var(varmap, t.var);
    out[i] = t;
}
// 256 consecutive terms per workgroup: the 24-byte structs come in and go out as 16-byte chunks through LDS (a per-thread struct
// copy is three strided 8-byte accesses each way); falls back to the per-thread form for unaligned buffers
__global__ __launch_bounds__(256) void pack_scalar_quadratic_kernel(const QT *__restrict__ in, int64_t n, const int64_t *__restrict__ varmap,
                                                                    QT *__restrict__ out, int aligned) {
    typedef unsigned long long u64;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    __shared__ u64 words[256 * 3];
    const int64_t base = (int64_t)blockIdx.x * 256;
    const int cnt = (int)min((int64_t)256, n - base);
    const int tid = threadIdx.x;
    if (aligned && cnt == 256) {
        const u64x2 *src = reinterpret_cast<const u64x2 *>(in + base);
        u64x2 *sw = reinterpret_cast<u64x2 *>(words);
        sw[tid] = src[tid];
        if (tid < 128) sw[256 + tid] = src[256 + tid];
        __syncthreads();
        const double c = __longlong_as_double((long long)words[3 * tid]);
        const int64_t r = (int64_t)words[3 * tid + 1], cl = (int64_t)words[3 * tid + 2];
        const double cm = (r == cl) ? 2 * c : c;                 // moi_interop.jl:58 (each thread rewrites its own three words)
        words[3 * tid] = (u64)__double_as_longlong(cm);
        words[3 * tid + 1] = (u64)map_var(varmap, r);
        words[3 * tid + 2] = (u64)map_var(varmap, cl);
        __syncthreads();
        u64x2 *dst = reinterpret_cast<u64x2 *>(out + base);
        dst[tid] = sw[tid];
        if (tid < 128) dst[256 + tid] = sw[256 + tid];
        return;
    }
    if (tid < cnt) {
        QT t = in[base + tid];
        QT o;
        o.coeff = (t.row == t.col) ? 2 * t.coeff : t.coeff;     // moi_interop.jl:58
        o.row = map_var(varmap, t.row);
        o.col = map_var(varmap, t.col);
        out[base + tid] = o;
    }
}
// one wave per row (handles ragged rows through row_ptr): 64 terms at a time through LDS, written as 16-byte chunks
__global__ __launch_bounds__(256) void pack_vector_affine_kernel(const LT *__restrict__ in, const int64_t *__restrict__ row_ptr, int64_t rows,
                                                                 int64_t row_len, const int64_t *__restrict__ varmap, int64_t row_offset,
                                                                 VAT *__restrict__ out) {
    typedef unsigned long long u64;
    __shared__ u64 s_c[4][64], s_v[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;                                     // (no workgroup barrier below: waves are independent)
    const int64_t beg = row_ptr ? row_ptr[row] : row * row_len;
    const int64_t end = row_ptr ? row_ptr[row + 1] : beg + row_len;
    const u64 rowword = (u64)(row_offset + row + 1);
    for (int64_t ts = beg; ts < end; ts += 64) {
        const int cnt = (int)min((int64_t)64, end - ts);
        if (lane < cnt) {
            const LT t = in[ts + lane];
            s_c[wave][lane] = (u64)__double_as_longlong(t.coeff);
            s_v[wave][lane] = (u64)map_var(varmap, t.var);
        }
        __builtin_amdgcn_wave_barrier();
        wave_write_words<3>(reinterpret_cast<u64 *>(out) + ts * 3, cnt, lane, [&](int q) -> u64 {
            const int t = q / 3, f = q - 3 * t;
            return f == 0 ? rowword : (f == 1 ? s_c[wave][t] : s_v[wave][t]);
        });
        __builtin_amdgcn_wave_barrier();
    }
}

static int validate_affine(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b,
                           int sign, const void *out_terms) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "affine: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "affine: lda < rows");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "affine: sign must be -1, 0 or +1");
    if (rows > 0 && cols > 0) {
        PMT_REQUIRE(A && xvar && out_terms, PMT_INVALID_ARGUMENT, "affine: null pointer");
    }
    PMT_REQUIRE(sign == 0 || b || rows == 0, PMT_INVALID_ARGUMENT, "affine: sign != 0 needs b");
    return PMT_OK;
}

// term buffers are write-once streams: non-temporal stores (tuning builds can switch them off with PMT_NONTEMPORAL=0)
static bool env_nt() {
#ifdef PMT_TUNING
    static int v = -1;
    if (v < 0) { const char *e = getenv("PMT_NONTEMPORAL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
#else
    return true;
#endif
}

// pmt_profile_kernel_stamps: device words {start, end} per workgroup the affine tile kernel of the NEXT launches reports into
static u64 *g_stamps = nullptr;
static unsigned g_stamp_cap = 0;

template <int MODE>
static int launch_affine(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b, int sign,
                         const int64_t *varmap, int64_t row_offset, void *out_terms, double *out_consts, hipStream_t s) {
    if (rows == 0 || cols == 0) return PMT_OK;   // cols == 0 (constants only) is handled by the callers
    const int vec_in = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
    const int vec_out = ((reinterpret_cast<uintptr_t>(out_terms) & 15) == 0 && (cols & 1) == 0) ? 1 : 0;
    const char *name = MODE == 0 ? "affine_tile_kernel<LT>" : "affine_tile_kernel<VAT>";
    u64 *out = reinterpret_cast<u64 *>(out_terms);
    // small blocks: 32-row tiles, twice the workgroups (see the kernel's comment); ~1024 = 4 per CU is where 64-row tiles start to fill the chip
    const bool small = cdiv(cols, TILE) * cdiv(rows, TILE) < 1024 && rows > 32;
    const bool ntl = MODE == 1 && rows * cols * (int64_t)sizeof(double) >= NTL_MIN_BYTES;
    u64 *stamps = MODE == 1 ? g_stamps : nullptr;
#define AFFINE_LAUNCH(NTV, TRV, NTLV)                                                                                                    \
    PMT_LAUNCH_NAMED(name, (affine_tile_kernel<MODE, NTV, TRV, NTLV>), dim3((unsigned)cdiv(cols, TILE), (unsigned)cdiv(rows, TRV)), dim3(256), 0, s, A, \
                     lda, rows, cols, xvar, b, sign, varmap, row_offset, out, out_consts, vec_in, vec_out, stamps, g_stamp_cap)
    if constexpr (MODE == 1) {
        if (ntl) {
            if (small) AFFINE_LAUNCH(true, 32, true); else AFFINE_LAUNCH(true, 64, true);
            return check_launch("affine_tile_kernel");
        }
    }
    if (env_nt()) { if (small) AFFINE_LAUNCH(true, 32, false); else AFFINE_LAUNCH(true, 64, false); }
    else { if (small) AFFINE_LAUNCH(false, 32, false); else AFFINE_LAUNCH(false, 64, false); }
#undef AFFINE_LAUNCH
    return check_launch("affine_tile_kernel");
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_profile_kernel_stamps(void *device_words, int64_t workgroups) {
    PMT_REQUIRE(workgroups >= 0 && workgroups < ((int64_t)1 << 31), PMT_INVALID_ARGUMENT, "profile_kernel_stamps: bad capacity");
    g_stamps = workgroups > 0 ? reinterpret_cast<u64 *>(device_words) : nullptr;
    g_stamp_cap = g_stamps ? (unsigned)workgroups : 0;
    return PMT_OK;
}

extern "C" int pmt_device_clock_khz(int device, int *khz) {
    PMT_REQUIRE(khz, PMT_INVALID_ARGUMENT, "device_clock_khz: null pointer");
    PMT_HIP_CHECK(hipDeviceGetAttribute(khz, hipDeviceAttributeWallClockRate, device));
    return PMT_OK;
}

extern "C" int pmt_affine_assemble_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b,
                                       int sign, pmt_linear_term *out_terms, double *out_consts, void *stream) {
    int rc = validate_affine(A, lda, rows, cols, xvar, b, sign, out_terms);
    if (rc) return rc;
    if (cols == 0 && rows > 0 && out_consts)
        return b && sign ? pmt_consts_f64(b, rows, sign, out_consts, stream)
                         : dispatch(stream, [=](hipStream_t s) { PMT_HIP_CHECK(hipMemsetAsync(out_consts, 0, rows * sizeof(double), s)); return PMT_OK; });
    SmallNode nd;
    nd.op = SOP_AFFINE_LT; nd.sign = sign; nd.d[0] = lda; nd.d[1] = rows; nd.d[2] = cols; nd.d[3] = 0;
    nd.in[0] = A; nd.in[1] = xvar; nd.in[2] = b; nd.out[0] = out_terms; nd.out[1] = out_consts; nd.work = rows * cols + rows;
    return dispatch(stream, [=](hipStream_t s) {
        return launch_affine<0>(A, lda, rows, cols, xvar, b, sign, nullptr, 0, out_terms, out_consts, s);
    }, nd);
}

extern "C" int pmt_affine_pack_vector_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b,
                                          int sign, const int64_t *varmap, int64_t row_offset, pmt_vector_affine_term *out_terms,
                                          double *out_consts, void *stream) {
    int rc = validate_affine(A, lda, rows, cols, xvar, b, sign, out_terms);
    if (rc) return rc;
    if (cols == 0 && rows > 0 && out_consts)
        return b && sign ? pmt_consts_f64(b, rows, sign, out_consts, stream)
                         : dispatch(stream, [=](hipStream_t s) { PMT_HIP_CHECK(hipMemsetAsync(out_consts, 0, rows * sizeof(double), s)); return PMT_OK; });
    SmallNode nd;
    nd.op = SOP_AFFINE_VAT; nd.sign = sign; nd.d[0] = lda; nd.d[1] = rows; nd.d[2] = cols; nd.d[3] = row_offset;
    nd.in[0] = A; nd.in[1] = xvar; nd.in[2] = b; nd.in[3] = varmap; nd.out[0] = out_terms; nd.out[1] = out_consts; nd.work = rows * cols + rows;
    return dispatch(stream, [=](hipStream_t s) {
        return launch_affine<1>(A, lda, rows, cols, xvar, b, sign, varmap, row_offset, out_terms, out_consts, s);
    }, nd);
}

extern "C" int pmt_affine_pack_vector_background_f64(const double *A, int64_t lda, int64_t rows, int64_t cols, const int64_t *xvar, const double *b,
                                                     int sign, const int64_t *varmap, int64_t row_offset, pmt_vector_affine_term *out_terms,
                                                     double *out_consts, void *stream) {
    int rc = validate_affine(A, lda, rows, cols, xvar, b, sign, out_terms);
    if (rc) return rc;
    PMT_REQUIRE(rows < (1 << 30) && cols < (1 << 30) && cdiv(cols, 256) * rows < ((int64_t)1 << 31), PMT_DIMENSION_MISMATCH,
                "affine_pack_vector_background: block too large");
    if (rows > 0 && out_consts) {
        rc = b && sign ? pmt_consts_f64(b, rows, sign, out_consts, stream)
                       : dispatch(stream, [=](hipStream_t s) { PMT_HIP_CHECK(hipMemsetAsync(out_consts, 0, rows * sizeof(double), s)); return PMT_OK; });
        if (rc) return rc;
    }
    if (rows == 0 || cols == 0) return PMT_OK;
    return dispatch(stream, [=](hipStream_t s) {
        const int64_t waves = cdiv(cols, 256) * rows;
        PMT_LAUNCH(affine_pack_background_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, s, A, lda, (int)rows, (int)cols, xvar, varmap, row_offset,
                   reinterpret_cast<unsigned long long *>(out_terms));
        return check_launch("affine_pack_background_kernel");
    });
}

extern "C" int pmt_vars_addsub_f64(const int64_t *xvar, int64_t n, const double *v, int sign, const int64_t *varmap, int64_t row_offset,
                                   pmt_linear_term *out_terms_lt, pmt_vector_affine_term *out_terms_vat, double *out_consts, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "vars_addsub: negative length");
    PMT_REQUIRE(sign >= -1 && sign <= 1, PMT_INVALID_ARGUMENT, "vars_addsub: sign must be -1, 0 or +1");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(xvar, PMT_INVALID_ARGUMENT, "vars_addsub: null xvar");
    PMT_REQUIRE(sign == 0 || v, PMT_INVALID_ARGUMENT, "vars_addsub: sign != 0 needs v");
    SmallNode nd;
    nd.op = SOP_VARS_ADDSUB; nd.sign = sign; nd.d[0] = n; nd.d[1] = row_offset; nd.in[0] = xvar; nd.in[1] = v; nd.in[2] = varmap;
    nd.out[0] = out_terms_lt; nd.out[1] = out_terms_vat; nd.out[2] = out_consts; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(vars_addsub_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, xvar, n, v, sign, varmap, row_offset,
                           out_terms_lt, out_terms_vat, out_consts);
        return check_launch("vars_addsub_kernel");
    }, nd);
}

extern "C" int pmt_consts_f64(const double *d, int64_t n, int sign, double *out, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "consts: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(d && out, PMT_INVALID_ARGUMENT, "consts: null pointer");
    SmallNode nd;
    nd.op = SOP_CONSTS; nd.sign = sign; nd.d[0] = n; nd.in[0] = d; nd.out[0] = out; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(consts_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, d, n, sign, out);
        return check_launch("consts_kernel");
    }, nd);
}

extern "C" int pmt_pack_scalar_affine_f64(const pmt_linear_term *terms, int64_t n, const int64_t *varmap, pmt_linear_term *out_terms,
                                          void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "pack_scalar_affine: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(terms && out_terms, PMT_INVALID_ARGUMENT, "pack_scalar_affine: null pointer");
    SmallNode nd;
    nd.op = SOP_PACK_SA; nd.d[0] = n; nd.in[0] = terms; nd.in[1] = varmap; nd.out[0] = out_terms; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(pack_scalar_affine_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, terms, n, varmap, out_terms);
        return check_launch("pack_scalar_affine_kernel");
    }, nd);
}

extern "C" int pmt_pack_scalar_quadratic_f64(const pmt_quadratic_term *quad, int64_t nq, const int64_t *varmap,
                                             pmt_quadratic_term *out_quad, void *stream) {
    PMT_REQUIRE(nq >= 0, PMT_DIMENSION_MISMATCH, "pack_scalar_quadratic: negative length");
    if (nq == 0) return PMT_OK;
    PMT_REQUIRE(quad && out_quad, PMT_INVALID_ARGUMENT, "pack_scalar_quadratic: null pointer");
    SmallNode nd;
    nd.op = SOP_PACK_SQ; nd.d[0] = nq; nd.in[0] = quad; nd.in[1] = varmap; nd.out[0] = out_quad; nd.work = nq;
    return dispatch(stream, [=](hipStream_t s) {
        const int aligned = (((reinterpret_cast<uintptr_t>(quad) | reinterpret_cast<uintptr_t>(out_quad)) & 15) == 0) ? 1 : 0;
        PMT_LAUNCH(pack_scalar_quadratic_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, quad, nq, varmap, out_quad, aligned);
        return check_launch("pack_scalar_quadratic_kernel");
    }, nd);
}

extern "C" int pmt_pack_vector_affine_f64(const pmt_linear_term *terms, const int64_t *row_ptr, int64_t rows, int64_t row_len,
                                          const int64_t *varmap, int64_t row_offset, pmt_vector_affine_term *out_terms, void *stream) {
    PMT_REQUIRE(rows >= 0 && row_len >= 0, PMT_DIMENSION_MISMATCH, "pack_vector_affine: negative dimension");
    if (rows == 0) return PMT_OK;
    PMT_REQUIRE((terms && out_terms) || (!row_ptr && row_len == 0), PMT_INVALID_ARGUMENT, "pack_vector_affine: null pointer");
    SmallNode nd;
    nd.op = SOP_PACK_VA; nd.d[0] = rows; nd.d[1] = row_len; nd.d[2] = row_offset; nd.in[0] = terms; nd.in[1] = row_ptr; nd.in[2] = varmap;
    nd.out[0] = out_terms;
    // (ragged rows: the term count is on the device; such a node joins a group only when its row count alone bounds it)
    nd.work = row_ptr ? SMALL_NODE_WORK_MAX + 1 : rows * row_len;
    return dispatch(stream, [=](hipStream_t s) {
        PMT_LAUNCH(pack_vector_affine_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, terms, row_ptr, rows, row_len, varmap,
                           row_offset, out_terms);
        return check_launch("pack_vector_affine_kernel");
    }, nd);
}
