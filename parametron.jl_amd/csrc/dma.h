// Copy-engine delivery to the host (hsadma.hip): an Engine per HIP device, signals whose value word kernels may write, copies that start
// when their dependency signal reads 0.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pmt {
namespace dma {

struct Engine;
struct Signal { uint64_t handle = 0; int64_t *value = nullptr; };     // value: the signal's 64-bit word (host and device address)

Engine *get(int device);                                             // null: no HSA runtime / no unambiguous agent match -> kernel copies
// pmt_set_host_delivery: 0 = copy engine when there is one (default), 1 = copy engine or an error, 2 = kernel copies (deliver.hip)
int delivery_mode();
// pmt_set_fault_injection (test hook): bit 0 = the first halves of a pair fold never raise their flag (gram_sk.hip);
// bit 1 = the grid barrier of a small plan's run on several workgroups waits for an arrival that never comes, with a 20 ms bound (small.hip)
int fault_injection();
int signal_create(Engine *e, int64_t initial, Signal *out);
void signal_destroy(Engine *e, Signal s);
void signal_set(Engine *e, Signal s, int64_t v);
// queue 0 / 1: two independent first-in-first-out queues (two SDMA engines) when the runtime offers a choice, else one
int copy_to_host(Engine *e, void *host_dst, const void *device_src, size_t bytes, const Signal *dep, Signal completion, int queue = 0);
// `height` rows of `width_bytes` bytes, row r at base + r * pitch on either side (the engine's rectangle copy)
int copy_rect_to_host(Engine *e, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes, size_t height,
                      const Signal *dep, Signal completion);
int wait(Engine *e, Signal completion, double timeout_s);
int launch_signal_store(Signal s, hipStream_t stream);

}  // namespace dma

// a pitched transfer (height == 0: a linear one)
struct FetchRect { size_t dst_pitch = 0, src_pitch = 0, height = 0; };

// per recorded fetch (plan.hip): the dependency / completion signals of its copy-engine transfer, created on first replay
struct FetchState {
    dma::Engine *eng = nullptr;
    dma::Signal dep, done;
    bool tried = false, created = false, pending = false;
    int pinned = -1;                       // host_dst is page-locked and device-mapped (the engine must not be given pageable memory); -1: not looked up yet
    ~FetchState();
};

}  // namespace pmt
