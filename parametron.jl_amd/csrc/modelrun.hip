// update!(m::Model) of a SMALL model behind ONE C call: what a host (Julia, Python, C) otherwise walks itself every solve —
//   setdirty!(m) + the Parameter refresh (src/model.jl:132-133, src/parameter.jl:93-104): the value of every host-updated Parameter into its
//     page-locked mailbox in the device layout, the next seed of every device-regenerated one into its seed word;
//   the re-evaluation of the objective's and the constraints' DAGs and their MOI copies (src/model.jl:134-143, src/moi_interop.jl:131-137,
//     168-175): one replay of the plan's tape (pmt_plan_update; for a small model ONE launch, small.hip);
//   the wait for the MOI buffers (page-locked arrays of the function objects the kernels store into) and the scalar functions' constants.
// Host-side code only: no kernel lives here.  The mailboxes, seed words and MOI buffers are the caller's: a pmt_model
// holds pointers and strides, never memory.
#include <cstring>
#include <vector>

#include "common.h"

struct pmt_plan;
extern "C" int pmt_plan_update(pmt_plan *plan);
extern "C" int pmt_plan_synchronize(pmt_plan *plan);
extern "C" int pmt_plan_fetch(pmt_plan *plan, void *host_dst, const void *device_src, size_t bytes);

namespace {

struct Mailbox {
    const double *host;              // the Parameter's value array as the host language holds it
    int64_t rows, cols;              // cols == 0: a vector (or a scalar: rows == 1)
    int64_t row_stride, col_stride;  // in doubles (Julia Matrix: 1, size(A, 1); numpy C order: shape[1], 1)
    double *mailbox; int64_t ld;     // page-locked, the device layout: column-major, `ld` doubles per column
};
struct Seed { uint64_t *word; uint64_t base, stride, updates; };
struct Constant { const double *src; double *dst; };
struct Fetch { void *host; const void *device; size_t bytes; };
struct Slot { int kind; size_t index; };      // 0 mailbox, 1 seed

}  // namespace

struct pmt_model {
    pmt_plan *plan = nullptr;
    std::vector<Slot> slots;
    std::vector<Mailbox> mailboxes;
    std::vector<Seed> seeds;
    std::vector<Constant> constants;
    std::vector<Fetch> fetches;
    bool replay_pending = false;     // a replay that may still be reading the mailboxes has not been waited for
};

using namespace pmt;

extern "C" int pmt_model_create(pmt_plan *plan, pmt_model **out) {
    PMT_REQUIRE(plan && out, PMT_INVALID_ARGUMENT, "model_create: null argument");
    pmt_model *m = new (std::nothrow) pmt_model;
    PMT_REQUIRE(m, PMT_OUT_OF_MEMORY, "model_create: out of memory");
    m->plan = plan;
    *out = m;
    return PMT_OK;
}

extern "C" int pmt_model_destroy(pmt_model *model) {
    delete model;                    // (the plan, the mailboxes and the seed words belong to the caller)
    return PMT_OK;
}

extern "C" int pmt_model_add_mailbox(pmt_model *model, const double *host, int64_t rows, int64_t cols, int64_t row_stride, int64_t col_stride,
                                     double *mailbox, int64_t ld, int *out_slot) {
    PMT_REQUIRE(model && mailbox, PMT_INVALID_ARGUMENT, "model_add_mailbox: null argument");
    PMT_REQUIRE(rows >= 0 && cols >= 0, PMT_DIMENSION_MISMATCH, "model_add_mailbox: negative dimension");
    PMT_REQUIRE(ld >= rows, PMT_DIMENSION_MISMATCH, "model_add_mailbox: ld < rows");
    PMT_REQUIRE(host || rows == 0, PMT_INVALID_ARGUMENT, "model_add_mailbox: null host array");
    model->mailboxes.push_back(Mailbox{host, rows, cols, row_stride, col_stride, mailbox, ld});
    model->slots.push_back(Slot{0, model->mailboxes.size() - 1});
    if (out_slot) *out_slot = (int)model->slots.size() - 1;
    return PMT_OK;
}

extern "C" int pmt_model_add_seed(pmt_model *model, uint64_t *seed_word, uint64_t base, uint64_t stride, int *out_slot) {
    PMT_REQUIRE(model && seed_word, PMT_INVALID_ARGUMENT, "model_add_seed: null argument");
    model->seeds.push_back(Seed{seed_word, base, stride, 0});
    model->slots.push_back(Slot{1, model->seeds.size() - 1});
    if (out_slot) *out_slot = (int)model->slots.size() - 1;
    return PMT_OK;
}

extern "C" int pmt_model_set_host(pmt_model *model, int slot, const double *host) {
    PMT_REQUIRE(model && slot >= 0 && (size_t)slot < model->slots.size() && model->slots[(size_t)slot].kind == 0, PMT_INVALID_ARGUMENT,
                "model_set_host: not a mailbox slot");
    model->mailboxes[model->slots[(size_t)slot].index].host = host;
    return PMT_OK;
}

extern "C" int pmt_model_add_constant(pmt_model *model, const double *src, double *dst) {
    PMT_REQUIRE(model && src && dst, PMT_INVALID_ARGUMENT, "model_add_constant: null argument");
    model->constants.push_back(Constant{src, dst});
    return PMT_OK;
}

// a result that does NOT live in host memory (a scalar function's constant that an expression node left in HBM): copied out behind every
// replay, in front of the wait (pmt_plan_fetch: asynchronous on the plan's stream; `host_dst` should be page-locked)
extern "C" int pmt_model_add_fetch(pmt_model *model, void *host_dst, const void *device_src, size_t bytes) {
    PMT_REQUIRE(model && host_dst && device_src, PMT_INVALID_ARGUMENT, "model_add_fetch: null argument");
    model->fetches.push_back(Fetch{host_dst, device_src, bytes});
    return PMT_OK;
}

extern "C" int pmt_model_num_slots(const pmt_model *model) { return model ? (int)model->slots.size() : 0; }

static void write_mailbox(const Mailbox &mb) {
    const int64_t ncols = mb.cols > 0 ? mb.cols : 1;
    for (int64_t c = 0; c < ncols; ++c) {
        double *d = mb.mailbox + c * mb.ld;
        const double *s = mb.host + c * mb.col_stride;
        if (mb.row_stride == 1) std::memcpy(d, s, sizeof(double) * (size_t)mb.rows);
        else
            for (int64_t i = 0; i < mb.rows; ++i) d[i] = s[i * mb.row_stride];
    }
}

// dirty: one byte per slot in registration order (non-zero: the Parameter's value changed since the last update — its callback ran, or the
// user overwrote a `val=` buffer), or null: every slot (setdirty!(model) with callbacks that always produce new values).
// synchronize != 0: returns when the MOI buffers are complete on the host (and the registered constants are stored); 0: the replay is
// enqueued only — pmt_model_wait finishes it.
extern "C" int pmt_model_update(pmt_model *model, const unsigned char *dirty, int nslots, int synchronize) {
    PMT_REQUIRE(model, PMT_INVALID_ARGUMENT, "model_update: null model");
    PMT_REQUIRE(!dirty || nslots == (int)model->slots.size(), PMT_DIMENSION_MISMATCH, "model_update: the dirty mask does not have one byte per slot");
    bool waited = !model->replay_pending;
    for (size_t k = 0; k < model->slots.size(); ++k) {
        if (dirty && !dirty[k]) continue;
        const Slot &sl = model->slots[k];
        if (sl.kind == 0) {
            if (!waited) {                       // the previous replay may still be reading the mailboxes
                if (int rc = pmt_plan_synchronize(model->plan)) return rc;
                model->replay_pending = false;
                waited = true;
            }
            write_mailbox(model->mailboxes[sl.index]);
        } else {
            Seed &sd = model->seeds[sl.index];
            // (a seed word is a kernel ARGUMENT of the launch, read when the launch is made: no wait)
            *sd.word = sd.base + sd.stride * sd.updates;
            ++sd.updates;
        }
    }
    if (int rc = pmt_plan_update(model->plan)) return rc;
    model->replay_pending = true;
    for (const Fetch &f : model->fetches)
        if (int rc = pmt_plan_fetch(model->plan, f.host, f.device, f.bytes)) return rc;
    if (!synchronize) return PMT_OK;
    if (int rc = pmt_plan_synchronize(model->plan)) return rc;
    model->replay_pending = false;
    for (const Constant &c : model->constants) *c.dst = *c.src;
    return PMT_OK;
}

extern "C" int pmt_model_wait(pmt_model *model) {
    PMT_REQUIRE(model, PMT_INVALID_ARGUMENT, "model_wait: null model");
    if (int rc = pmt_plan_synchronize(model->plan)) return rc;
    model->replay_pending = false;
    for (const Constant &c : model->constants) *c.dst = *c.src;
    return PMT_OK;
}
