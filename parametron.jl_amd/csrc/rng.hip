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

__global__ void fill_uniform_kernel(double *__restrict__ dst, int64_t n, uint64_t base, double scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = scale * ((double)(splitmix64(base + (uint64_t)i) >> 11) * 0x1.0p-53);
}

// column-major matrix with a padded leading dimension: dst[c*lda + i] = scale * U(seed, c*rows + i) — the VALUES are those of the
// contiguous stream (identical to pmt_fill_uniform_f64 on a rows x cols array); only the placement in HBM differs
__global__ void fill_uniform_matrix_kernel(double *__restrict__ dst, int64_t rows, int64_t cols, int64_t lda, uint64_t base, double scale) {
    const int64_t n = rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t c = i / rows, r = i - c * rows;
        dst[c * lda + r] = scale * ((double)(splitmix64(base + (uint64_t)i) >> 11) * 0x1.0p-53);
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
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(rows * cols, 256), 256 * 8);
        PMT_LAUNCH(fill_uniform_matrix_kernel, dim3(blocks), dim3(256), 0, s, dst, rows, cols, lda, base, scale);
        return check_launch("fill_uniform_matrix_kernel");
    });
}

// Same stream starting at element `index_offset`: dst[i] = scale * U(seed, index_offset + i).  Lets a shard of a larger
// array be generated independently of how the array is partitioned across GPUs (batched instances, config 4).
extern "C" int pmt_fill_uniform_offset_f64(double *dst, int64_t n, uint64_t seed, uint64_t index_offset, double scale, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform: null pointer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + index_offset;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 8);
        PMT_LAUNCH(fill_uniform_kernel, dim3(blocks), dim3(256), 0, s, dst, n, base, scale);
        return check_launch("fill_uniform_kernel");
    });
}

extern "C" int pmt_fill_uniform_f64(double *dst, int64_t n, uint64_t seed, double scale, void *stream) {
    PMT_REQUIRE(n >= 0, PMT_DIMENSION_MISMATCH, "fill_uniform: negative length");
    if (n == 0) return PMT_OK;
    PMT_REQUIRE(dst, PMT_INVALID_ARGUMENT, "fill_uniform: null pointer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull;
    return dispatch(stream, [=](hipStream_t s) {
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 8);
        PMT_LAUNCH(fill_uniform_kernel, dim3(blocks), dim3(256), 0, s, dst, n, base, scale);
        return check_launch("fill_uniform_kernel");
    });
}
