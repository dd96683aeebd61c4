// Host -> host pitched copy on a few worker threads.
//
// A host solver's A (CSC values, handoff "host_csc") contains, for every DENSE constraint block, the block's Parameter matrix column by
// column.  When that Parameter is host-updated — `Parameter(model, val=buf)` or a callback `f(val)`, src/parameter.jl:88,101-102 — the
// values are ALREADY on the host: they need not come back from the device over the same PCIe link the objective's 67 MB are crossing.
// They are copied on the host instead, from the Parameter's buffer into the block's row range of every column of the solver's array,
// while the device re-evaluates the objective.  16.8 MB at config 3: ~0.3 ms on eight threads, hidden behind 1.7 ms of device work; one
// Python-level numpy copy of the same bytes takes longer than the whole solve.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

extern "C" int pmt_host_copy_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t height, int threads) {
    if (width_bytes == 0 || height == 0) return PMT_OK;
    PMT_REQUIRE(dst && src && dst_pitch >= width_bytes && src_pitch >= width_bytes, PMT_INVALID_ARGUMENT, "host_copy_2d: bad argument");
    PMT_REQUIRE(threads >= 0 && threads <= 64, PMT_INVALID_ARGUMENT, "host_copy_2d: threads must be 0 (default) .. 64");
    const size_t total = width_bytes * height;
    int nt = threads > 0 ? threads : 8;
    nt = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, total >> 20));          // at least 1 MiB per thread
    nt = (int)std::min<size_t>((size_t)nt, height);
    auto work = [=](size_t r0, size_t r1) {
        char *d = static_cast<char *>(dst) + r0 * dst_pitch;
        const char *s = static_cast<const char *>(src) + r0 * src_pitch;
        if (dst_pitch == width_bytes && src_pitch == width_bytes) { memcpy(d, s, (r1 - r0) * width_bytes); return; }
        for (size_t r = r0; r < r1; ++r, d += dst_pitch, s += src_pitch) memcpy(d, s, width_bytes);
    };
    if (nt <= 1) { work(0, height); return PMT_OK; }
    // The pool lives OUTSIDE the try block: when a thread cannot be started (std::system_error) the ones already running are joinable and
    // must be joined, not destroyed (std::terminate).  Rows whose thread never started are copied by the calling thread; nothing is copied twice.
    std::vector<std::thread> pool;
    const size_t per = (height + (size_t)nt - 1) / (size_t)nt;
    // rows [0, per) are the calling thread's; `unassigned` = first row beyond them that no STARTED thread owns — it moves forward only behind
    // a successful emplace_back, so whatever throws (reserve's std::bad_alloc, a thread that cannot be started) leaves the rest to the caller
    size_t unassigned = std::min(height, per);
    try {
        pool.reserve((size_t)nt - 1);
        for (int t = 1; t < nt; ++t) {
            const size_t r0 = std::min(height, (size_t)t * per), r1 = std::min(height, r0 + per);
            if (r0 < r1) pool.emplace_back(work, r0, r1);
            unassigned = r1;
        }
        unassigned = height;
    } catch (const std::exception &) {
    }
    work(0, std::min(height, per));
    if (unassigned < height) work(unassigned, height);
    for (auto &th : pool) th.join();
    return PMT_OK;
}
