// Shared host-side plumbing for libparametron_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>

#include "parametron_hip.h"

namespace pmt {

using LT = pmt_linear_term;
using QT = pmt_quadratic_term;
using VAT = pmt_vector_affine_term;
static_assert(sizeof(LT) == 16 && sizeof(QT) == 24 && sizeof(VAT) == 24, "Julia isbits layouts");

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define PMT_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::pmt::fail(PMT_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define PMT_REQUIRE(cond, code, msg)                 \
    do {                                             \
        if (!(cond)) return ::pmt::fail(code, msg);  \
    } while (0)

// A launch either runs now on `stream` or, when `stream` is a plan's recording handle, is
// appended to that plan's tape (plan.cpp).
using Launch = std::function<int(hipStream_t)>;
int dispatch(void *stream, Launch launch);

// Small plans (small.hip): the reference walks a tiny model's DAG in nanoseconds per hop (src/lazyexpression.jl:50-61); on the device every
// hop is a kernel launch of ~5 us.  An entry point whose work can also be described by a SmallNode records the description beside its
// closure; pmt_plan_end_record then replaces every run of consecutive small nodes by ONE launch of an interpreter kernel that executes
// them in tape order with a workgroup barrier between dependent nodes.  Outputs are bit-identical to the separate kernels.
enum SmallOp : int {
    SOP_FILL = 1, SOP_AFFINE_LT, SOP_AFFINE_VAT, SOP_QUAD_EXPAND, SOP_VARS_ADDSUB, SOP_CONSTS, SOP_PACK_SA, SOP_PACK_SQ, SOP_PACK_VA, SOP_COPY8, SOP_GRAM,
    SOP_AFFVEC_COMBINE, SOP_AFFVEC_SCALE, SOP_MATVEC_AFFS, SOP_VECDOT_NUM_VARS, SOP_VECDOT_NUM_AFFS, SOP_TRANSPOSE, SOP_QUAD_COMBINE,
    SOP_QUAD_SCALE, SOP_SCALE_VARS, SOP_SCALE_NUMBERS, SOP_BILINEAR, SOP_VECDOT_TERMS, SOP_VECDOT_AFFS_VARS
};
struct SmallNode {
    int op = 0, sign = 0, moi = 0;
    int dyn = -1;                   // slot of the launch's dynamic-seed table (SOP_FILL with a host seed word), or -1
    int sync = 1;                   // a barrier in front of the node (set by small_plan_phases, small.hip)
    int64_t d[4] = {0, 0, 0, 0};
    const void *in[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *out[3] = {nullptr, nullptr, nullptr};
    double scale = 0.0;
    uint64_t seed = 0;
    // host side only
    int64_t work = 0;               // elements written (the grouping bound)
    const uint64_t *seed_host = nullptr;
};
constexpr int SMALL_MAX_DYN = 16;
constexpr int64_t SMALL_NODE_WORK_MAX = 32768;       // a node writing more than this many elements is launched on its own
constexpr int64_t SMALL_GROUP_WORK_MAX = 65536;      // ... and a group is cut before it exceeds this
int dispatch(void *stream, Launch launch, const SmallNode &node);
bool is_recording_handle(void *stream);        // `stream` is a plan's recording handle, not a HIP stream

// Per-kernel timing (the analogue of the reference's per-node `findallocs` report, src/debug.jl:4-23): when enabled
// through pmt_profile_enable(), every launch is bracketed by HIP events on its own stream.
struct ProfScope {
    ProfScope(const char *name, hipStream_t s);
    ~ProfScope();
    const char *name_; hipStream_t s_; hipEvent_t e0_ = nullptr, e1_ = nullptr;
};
#define PMT_LAUNCH_NAMED(name, kernel, grid, block, shmem, s, ...)          \
    do {                                                                    \
        ::pmt::ProfScope _prof(name, s);                                    \
        hipLaunchKernelGGL(kernel, grid, block, shmem, s, __VA_ARGS__);     \
    } while (0)
#define PMT_LAUNCH(kernel, grid, block, shmem, s, ...) PMT_LAUNCH_NAMED(#kernel, kernel, grid, block, shmem, s, __VA_ARGS__)

inline int check_launch(const char *name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PMT_HIP_ERROR, std::string(name) + ": " + hipGetErrorString(e));
    return PMT_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 0.0 (+|-) v : the constant a zero!'d AffineFunction ends up with after add!/subtract! of a Number
// (src/functions.jl:244,452,474).  sign == 0 -> 0.0.
__device__ __forceinline__ double signed_const(double v, int sign) {
    return sign > 0 ? 0.0 + v : (sign < 0 ? 0.0 - v : 0.0);
}
__device__ __forceinline__ int64_t map_var(const int64_t *__restrict__ varmap, int64_t var) {
    return varmap ? varmap[var - 1] : var;
}

// One wave writes `nterms` consecutive terms of W 8-byte words each, starting at `seg`, as 16-byte chunks (global_store_dwordx4)
// instead of W strided 8-byte stores per lane; word(q) returns word q of the segment (term q / W, field q % W).  `seg` is 8-byte
// aligned only, so a leading / trailing single word is handled.  All lanes of the wave call this together.
template <int W, typename F>
__device__ __forceinline__ void wave_write_words(unsigned long long *__restrict__ seg, int nterms, int lane, F word) {
    typedef unsigned long long u64w;
    typedef u64w u64w2 __attribute__((ext_vector_type(2)));
    const int nwords = nterms * W;
    const int lead = (int)((reinterpret_cast<uintptr_t>(seg) >> 3) & 1);
    if (lead && lane == 0) seg[0] = word(0);
    for (int c = lane; lead + 2 * c < nwords; c += 64) {
        const int q0 = lead + 2 * c;
        if (q0 + 1 < nwords) {
            u64w2 v;
            v.x = word(q0);
            v.y = word(q0 + 1);
            *reinterpret_cast<u64w2 *>(seg + q0) = v;
        } else {
            seg[q0] = word(q0);
        }
    }
}

}  // namespace pmt
