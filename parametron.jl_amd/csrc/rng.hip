// Device-side Parameter update callback for synthetic inputs: counter-based U[0,1) stream identical
// bit for bit to oracle/parametron_oracle.c:pmo_fill_uniform and tests/golden/make_fill_golden.py
// (SURVEY.md §8d).  The analogue of `Parameter(rand!, zeros(n, n), model)` (README.md:36-43) with the
// value buffer resident in HBM.
#include "common.h"

namespace pmt {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double uniform_at(uint64_t base, uint64_t i, double scale) {
    return scale * ((double)(splitmix64(base + i) >> 11) * 0x1.0p-53);
}

typedef double f64x2 __attribute__((ext_vector_type(2)));

// two consecutive elements per thread and store (16 bytes per lane: a wave store covers 1 KiB); an unaligned head / odd tail go singly
__global__ void fill_uniform_kernel(double *__restrict__ dst, int64_t n, uint64_t base, double scale) {
    const int64_t head = (int64_t)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);           // dst + head is 16-byte aligned
    const int64_t npairs = (n - head) / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 == 0 && head && n > 0) dst[0] = uniform_at(base, 0, scale);
    if (t0 == 0 && head + 2 * npairs < n) dst[n - 1] = uniform_at(base, (uint64_t)(n - 1), scale);
    for (int64_t p = t0; p < npairs; p += stride) {
        const int64_t i = head + 2 * p;
        f64x2 v;
        v.x = uniform_at(base, (uint64_t)i, scale);
        v.y = uniform_at(base, (uint64_t)i + 1, scale);
        *reinterpret_cast<f64x2 *>(dst + i) = v;
    }
}

// column-major matrix with a padded leading dimension: dst[c*lda + i] = scale * U(seed, c*rows + i) — the VALUES are those of the
// contiguous stream (identical to pmt_fill_uniform_f64 on a rows x cols array); only the placement in HBM differs.
// blockIdx.y walks the columns (no division per element), a thread writes a row pair of its column as one 16-byte store when the column
// starts on a 16-byte boundary (even lda, aligned base), singly otherwise.
__global__ void fill_uniform_matrix_kernel(double *__restrict__ dst, int64_t rows, int64_t cols, int64_t lda, uint64_t base, double scale, int vec) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = 2 * t;
    if (r >= rows) return;
    for (int64_t c = blockIdx.y; c < cols; c += gridDim.y) {
        double *col = dst + c * lda;
        const uint64_t i = (uint64_t)(c * rows + r);
        const double x = uniform_at(base, i, scale);
        if (r + 1 < rows) {
            const double y = uniform_at(base, i + 1, scale);
            if (vec) { f64x2 v; v.x = x; v.y = y; *reinterpret_cast<f64x2 *>(col + r) = v; }
            else { col[r] = x; col[r + 1] = y; }
        } else {
            col[r] = x;
        }
    }
}

}  // namespace pmt

using namespace pmt;

extern "C" int pmt_fill_uniform_matrix_f64(double *dst, int64_t rows, int64_t cols, int64_t lda, uint64_t seed, double scale, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform_matrix: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "fill_uniform_matrix: lda < rows");
    if (rows == 0 || cols == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform_matrix: null pointer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull;
    SmallNode nd;
    nd.op = SOP_FILL; nd.d[0] = rows; nd.d[1] = cols; nd.d[2] = lda; nd.out[0] = dst; nd.scale = scale; nd.seed = seed; nd.work = rows * cols;
    return dispatch(stream, [=](hipStream_t s) {
        const int vec = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
        const int64_t bx = cdiv(cdiv(rows, 2), 256);
        // enough workgroups to fill the chip (8 per CU) without more than the grid's y limit; the rest of the columns are walked in the kernel
        const unsigned by = (unsigned)std::min<int64_t>(cols, std::max<int64_t>(1, std::min<int64_t>(65535, cdiv(2048, bx))));
        PMT_LAUNCH(fill_uniform_matrix_kernel, dim3((unsigned)bx, by), dim3(256), 0, s, dst, rows, cols, lda, base, scale, vec);
        return check_launch("fill_uniform_matrix_kernel");
    }, nd);
}

// The same fill with the seed read from a HOST word at every launch / replay: a recorded Parameter callback (README.md:36-43 rand!, which
// draws new values at every update!) whose stream advances from one re-evaluation to the next without the host re-issuing the call —
// the host stores the next seed into *seed_word before pmt_plan_update.  cols == 1, lda == rows: a vector.
extern "C" int pmt_fill_uniform_dyn_f64(double *dst, int64_t rows, int64_t cols, int64_t lda, const uint64_t *seed_word, double scale, void *stream) {
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform_dyn: negative dimension");
    PMT_REQUIRE(lda >= rows, PMT_DIMENSION_MISMATCH, "fill_uniform_dyn: lda < rows");
    PMT_REQUIRE(seed_word, PMT_INVALID_ARGUMENT, "fill_uniform_dyn: null seed word");
    if (rows == 0 || cols == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform_dyn: null pointer");
    SmallNode nd;
    nd.op = SOP_FILL; nd.d[0] = rows; nd.d[1] = cols; nd.d[2] = lda; nd.out[0] = dst; nd.scale = scale; nd.seed_host = seed_word; nd.work = rows * cols;
    return dispatch(stream, [=](hipStream_t s) {
        const uint64_t base = *seed_word * 0x9E3779B97F4A7C15ull;
        const int vec = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (lda & 1) == 0) ? 1 : 0;
        const int64_t bx = cdiv(cdiv(rows, 2), 256);
        const unsigned by = (unsigned)std::min<int64_t>(cols, std::max<int64_t>(1, std::min<int64_t>(65535, cdiv(2048, bx))));
        PMT_LAUNCH(fill_uniform_matrix_kernel, dim3((unsigned)bx, by), dim3(256), 0, s, dst, rows, cols, lda, base, scale, vec);
        return check_launch("fill_uniform_matrix_kernel");
    }, nd);
}

// Same stream starting at element `index_offset`: dst[i] = scale * U(seed, index_offset + i).  Lets a shard of a larger
// array be generated independently of how the array is partitioned across GPUs (batched instances, config 4).
extern "C" int pmt_fill_uniform_offset_f64(double *dst, int64_t n, uint64_t seed, uint64_t index_offset, double scale, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform: null pointer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + index_offset;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(cdiv(n, 2), 256), 256 * 8);
        PMT_LAUNCH(fill_uniform_kernel, dim3(blocks), dim3(256), 0, s, dst, n, base, scale);
        return check_launch("fill_uniform_kernel");
    });
}

extern "C" int pmt_fill_uniform_f64(double *dst, int64_t n, uint64_t seed, double scale, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform: null pointer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull;
    SmallNode nd;
    nd.op = SOP_FILL; nd.d[0] = n; nd.d[1] = 1; nd.d[2] = n; nd.out[0] = dst; nd.scale = scale; nd.seed = seed; nd.work = n;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(cdiv(n, 2), 256), 256 * 8);
        PMT_LAUNCH(fill_uniform_kernel, dim3(blocks), dim3(256), 0, s, dst, n, base, scale);
        return check_launch("fill_uniform_kernel");
    }, nd);
}
