// Delivery to a HOST solver while the re-evaluation is still running (the reference's boundary is a host optimizer: MOI.set,
// src/moi_interop.jl:131-137,168-175; OSQP's update takes P's and A's CSC values, q, l, u).
//
// Measured first (tools/deliver_probe.hip, profiles/r03_host_delivery.txt): on this stack hipMemcpyAsync device -> page-locked host is a
// blit KERNEL (__amd_rocclr_copyBuffer), and hipStreamWaitValue64 is a KERNEL that spins (__amd_rocclr_streamOpsWait) — neither is a
// copy-engine / command-processor operation, both take wave slots and registers the persistent contraction (2 x 248 of a SIMD's 512 VGPRs)
// does not leave, so "wait for a band group, then copy it" from the runtime slowed the contraction from 1.24 to 1.66 ms and starved the
// node's side kernels.  A store from a kernel into page-locked host memory runs at the same 54 GB/s as the runtime's copy.  Hence two
// kernels of at most 16 VGPRs and no LDS, i.e. CO-RESIDENT with the contraction (like the node's own small reductions).  They are the
// FALLBACK: the preferred path hands the copies to the copy engine, gated by signals the kernels set (hsadma.hip) — a courier that shares
// CUs with the contraction costs it dearly too (its stores to the host queue up in the CU's memory pipeline in front of the
// contraction's panel loads: 1.19 -> 1.95 ms).  They run when the process's HSA runtime cannot be reached:
//   courier_kernel   one launch per delivery: for every band group, ONE lane per workgroup polls the group's flag (s_sleep between
//                    polls; a one-thread kernel behind the stage that completes the group clears it, gram.hip), then the workgroup's
//                    slice of the group is copied straight into the host array;
//   to_host_kernel   a plain device -> host copy (recorded fetches: q, A's values, bounds).
// Loads are agent-scope relaxed atomic loads (`sc1`): the producing launches have ended, and no line of this XCD's L2 may serve a stale copy
// (a 128-byte line can straddle two band groups).  The courier gives up after ~2 s without progress and raises an error flag instead of
// hanging the GPU.
#include "gram_common.h"

namespace pmt {

struct CourierArgs {
    const double *src;                    // out_csc (device)
    double *dst;                          // device address of the page-locked host array
    long long *ready;                     // per band group: 1 = in the making, 0 = complete (cleared by the one-thread kernel behind the group's stage)
    unsigned *done;                       // workgroups that have finished; the last one re-arms the flags for the next launch
    int *error;                           // set to 1 on timeout (page-locked host word, gram.hip SideStream::err_host)
    int ngroups;
    long long gbeg[MAXGROUPS], gend[MAXGROUPS];      // the groups' ranges of the array (doubles)
};

constexpr long long COURIER_TIMEOUT_TICKS = 200000000LL;      // wall_clock64 runs at 100 MHz: 2 s

__global__ __launch_bounds__(256) void courier_kernel(CourierArgs a) {
    __shared__ int s_abort;
    const int tid = threadIdx.x;
    for (int g = 0; g < a.ngroups; ++g) {
        if (tid == 0) {
            int abort = 0;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&a.ready[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
                __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t0 > COURIER_TIMEOUT_TICKS) { abort = 1; break; }
            }
            s_abort = abort;
        }
        __syncthreads();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        const long long n = a.gend[g] - a.gbeg[g];
        const double *src = a.src + a.gbeg[g];
        double *dst = a.dst + a.gbeg[g];
        // (32-bit offsets from uniform bases, two loads in flight: the kernel has to stay within 16 VGPRs)
        // BYTE offsets (a group is below 4 GiB: checked by the launcher) so that every access is uniform base + 32-bit lane offset
        const char *sb = reinterpret_cast<const char *>(src);
        char *db = reinterpret_cast<char *>(dst);
        const unsigned stride = gridDim.x * 2048u, nb = (unsigned)n * 8u;
        unsigned o = (blockIdx.x * 256u + (unsigned)tid) * 8u;
#pragma unroll 1
        for (; o + stride < nb; o += 2 * stride) {
            const double v0 = __hip_atomic_load(reinterpret_cast<const double *>(sb + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double v1 = __hip_atomic_load(reinterpret_cast<const double *>(sb + o + stride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<double *>(db + o) = v0;
            *reinterpret_cast<double *>(db + o + stride) = v1;
        }
        if (o < nb) *reinterpret_cast<double *>(db + o) = __hip_atomic_load(reinterpret_cast<const double *>(sb + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                        // s_abort is rewritten by the next group's poll
    }
    // the last workgroup to finish re-arms the flags (all polls are over): ready for the next launch
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            for (int g = 0; g < a.ngroups; ++g) __hip_atomic_store(&a.ready[g], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dst (host, 8-byte words) = src (device)
__global__ __launch_bounds__(256) void to_host_kernel(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, int n) {
    // (two loads in flight, uniform bases + 32-bit byte offsets: at most 16 VGPRs; the launcher keeps a launch below 4 GiB)
    const char *sb = reinterpret_cast<const char *>(src);
    char *db = reinterpret_cast<char *>(dst);
    const unsigned stride = gridDim.x * 2048u, nb = (unsigned)n * 8u;
    unsigned o = (blockIdx.x * 256u + threadIdx.x) * 8u;
#pragma unroll 1
    for (; o + stride < nb; o += 2 * stride) {
        const unsigned long long v0 = *reinterpret_cast<const unsigned long long *>(sb + o);
        const unsigned long long v1 = *reinterpret_cast<const unsigned long long *>(sb + o + stride);
        *reinterpret_cast<unsigned long long *>(db + o) = v0;
        *reinterpret_cast<unsigned long long *>(db + o + stride) = v1;
    }
    if (o < nb) *reinterpret_cast<unsigned long long *>(db + o) = *reinterpret_cast<const unsigned long long *>(sb + o);
}

// pitched: `height` rows of `wwords` 8-byte words; row r at src + r * spitch / dst + r * dpitch (pitches in words)
__global__ __launch_bounds__(256) void to_host_2d_kernel(const unsigned long long *__restrict__ src, unsigned spitch, unsigned long long *__restrict__ dst,
                                                         unsigned dpitch, unsigned wwords, unsigned height) {
    const unsigned total = wwords * height;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned r = i / wwords, c = i - r * wwords;
        dst[(size_t)r * dpitch + c] = src[(size_t)r * spitch + c];
    }
}

// device-visible address of a page-locked host buffer, or null when it is not mapped (pageable memory: the runtime's copy is used)
void *host_device_pointer(void *host) {
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

int launch_courier(const double *src, double *dst_dev, long long *ready, unsigned *done, int *error, int ngroups, const int64_t *gbeg, const int64_t *gend,
                   hipStream_t s) {
    for (int g = 0; g < ngroups; ++g)
        PMT_REQUIRE((gend[g] - gbeg[g]) * 8 < ((int64_t)1 << 31), PMT_DIMENSION_MISMATCH, "host delivery: a band group of 2 GiB or more (use more groups)");
    CourierArgs a;
    a.src = src; a.dst = dst_dev; a.ready = ready; a.done = done; a.error = error; a.ngroups = ngroups;
    for (int g = 0; g < MAXGROUPS; ++g) { a.gbeg[g] = g < ngroups ? gbeg[g] : 0; a.gend[g] = g < ngroups ? gend[g] : 0; }
    // 64 workgroups: page-locked stores saturate PCIe from 64 workgroups on (tools/deliver_probe.hip: 53.6 GB/s), and at most one courier
    // wave sits beside the contraction's two on a quarter of the SIMDs
    PMT_LAUNCH(courier_kernel, dim3(64), dim3(256), 0, s, a);
    return check_launch("courier_kernel");
}

int launch_to_host(const void *src, void *dst_dev, size_t bytes, hipStream_t s) {
    constexpr size_t PIECE = (size_t)1 << 31;                    // 32-bit byte offsets inside a launch
    for (size_t o = 0; o < bytes; o += PIECE) {
        const int n = (int)(std::min(PIECE, bytes - o) / 8);
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, cdiv(n, 1024)));
        PMT_LAUNCH(to_host_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const unsigned long long *>(static_cast<const char *>(src) + o),
                   reinterpret_cast<unsigned long long *>(static_cast<char *>(dst_dev) + o), n);
    }
    return check_launch("to_host_kernel");
}

int launch_to_host_2d(const void *src, size_t src_pitch, void *dst_dev, size_t dst_pitch, size_t width_bytes, size_t height, hipStream_t s) {
    PMT_REQUIRE(width_bytes / 8 * height < ((size_t)1 << 32) && src_pitch / 8 < ((size_t)1 << 32) && dst_pitch / 8 < ((size_t)1 << 32), PMT_DIMENSION_MISMATCH,
                "host delivery: a pitched block of 2^32 words or more");
    if (!width_bytes || !height) return PMT_OK;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, cdiv((int64_t)(width_bytes / 8 * height), 1024)));
    PMT_LAUNCH(to_host_2d_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned long long *>(src), (unsigned)(src_pitch / 8),
               static_cast<unsigned long long *>(dst_dev), (unsigned)(dst_pitch / 8), (unsigned)(width_bytes / 8), (unsigned)height);
    return check_launch("to_host_2d_kernel");
}

}  // namespace pmt
