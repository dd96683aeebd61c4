// Copy-engine delivery to the host (hsadma.hip): an Engine per HIP device, signals whose value word kernels may write, copies that start
// when their dependency signal reads 0.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pmt {
namespace dma {

struct Engine;
struct Signal { uint64_t handle = 0; int64_t *value = nullptr; };     // value: the signal's 64-bit word (host and device address)

Engine *get(int device);                                             // null: no HSA runtime / no matching agent -> kernel copies
int signal_create(Engine *e, int64_t initial, Signal *out);
void signal_destroy(Engine *e, Signal s);
void signal_set(Engine *e, Signal s, int64_t v);
// queue 0 / 1: two independent first-in-first-out queues (two SDMA engines) when the runtime offers a choice, else one
int copy_to_host(Engine *e, void *host_dst, const void *device_src, size_t bytes, const Signal *dep, Signal completion, int queue = 0);
int wait(Engine *e, Signal completion, double timeout_s);
int launch_signal_store(Signal s, hipStream_t stream);

}  // namespace dma

// per recorded fetch (plan.hip): the dependency / completion signals of its copy-engine transfer, created on first replay
struct FetchState {
    dma::Engine *eng = nullptr;
    dma::Signal dep, done;
    bool tried = false, created = false, pending = false;
    ~FetchState();
};

}  // namespace pmt
