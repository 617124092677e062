// api_internal.hpp -- what api.cpp (the drop-in boundary, include/mmdx.h) and bench_api.cpp (the measurement / A-B entry points,
// include/mmdx_bench.h) share: the model handle, its device buffers, the error mapping and the launch-shape overrides.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mmdx.h"
#include "../../include/mmdx_bench.h"
#include "error.hpp"
#include "graph_pin.hpp"
#include "kernels.hpp"
#include "plan.hpp"
#include "rig_kernels.hpp"

namespace mmdx {

// HIP error -> status + message (also clears the runtime's sticky error)
mmdx_status hip_fail(hipError_t e, const char *what);

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);      \
    } while (0)

struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    const GraphPin *pin = nullptr;           // per-call scratch of a handle: may not move while a recorded graph holds it
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (graph_recording()) return hipErrorStreamCaptureUnsupported;   // run the sequence once un-captured first
        if (graph_pinned(pin)) return hipErrorIllegalState;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&ptr, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        device_free_or_defer(ptr);
        ptr = nullptr; bytes = 0;
    }
};

// Launch-shape overrides for A/B runs (tools/): read ONCE, at the first deform call of the process -- the
// per-frame call has a budget of a few microseconds and getenv walks the whole environment.
struct LaunchOverrides {
    int interleave, threads, lds_target, group, placement_log, placement_park;
    int frame_kernel;   // MMDX_FRAME_KERNEL: 0 = a single frame always runs the tile kernel, 1 = models of fewer than 256 tiles run the
                        // frame kernel (default), 2 = always (A/B); MMDX_FRAME_THREADS: 128 / 256 lanes per workgroup
    int frame_threads;
    int shared_fused;   // MMDX_SHARED_FUSED: crowds with a shared facial state gather the morphs inside the deform kernel: 0 never,
                        // 1 up to 8 instances (default), 2 always (A/B, tests)
    int store_wt;       // MMDX_STORE_WT: 0 / 1 force cached / write-through stores where the caller gave no hint (A/B); -1 default
    int morph_autoskip; // MMDX_MORPH_AUTOSKIP: 0 turns the automatic "shared rates unchanged" detection off (A/B); 1 default
    int fused_pack;     // MMDX_FUSED_PACK: 0 = per-instance morph weights run deform_kernel<512, ., kMorphFused4> (default),
                        // 1 = pack_kernel (round 4's higher-occupancy shape: measured slower, kept for the A/B)
    int stagger;        // MMDX_STAGGER: start offset between the workgroups of a CU, in units of 64 cycles per residency slot (A/B)
};
LaunchOverrides read_launch_overrides();
LaunchOverrides &launch_overrides();
int env_int(const char *name, int dflt);
// average ms of `iters` store-pattern launches on the default stream (after one warm-up launch); bench_api.cpp
hipError_t time_store_pattern(void *a, void *b, uint32_t nv, uint32_t ni, uint32_t bpva, uint32_t bpvb, int iters, float *avg_ms);

}  // namespace mmdx

struct mmdx_model_s {
    mmdx::Plan plan;
    int device = -1;  // -1: host-only
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    // per-call kernel timing (mmdx_profile_*): event quadruples {skin0, skin1, morph0, morph1},
    // recorded on the launch stream without any host sync; read back by mmdx_profile_collect
    std::vector<hipEvent_t> prof_events;
    std::vector<uint8_t> prof_has_morph;
    size_t prof_calls = 0;
    bool profile = false;
    uint32_t profile_stride = 1, prof_seen = 0;     // time every profile_stride-th call
    uint64_t device_bytes = 0;
    // static streams
    mmdx::DevBuf tiles, spos, snrm, suv, perm, skin1, skin2_ids, skin2_w, skin4_ids, skin4_w, bone_list,
        ell, entries, slot_top, chain_off, chain_rate;
    // per-call scratch (grown on demand, reused)
    mmdx::DevBuf pal, rates, wslot, morphed, out_a, out_b;
    mmdx::DevBuf seen;              // RatesSeen record (kernels.hpp): the rates `morphed` was last computed from, device side
    bool morphed_valid = false;     // `morphed` holds the result of a shared morph pass (MMDX_MORPH_UNCHANGED)
    std::vector<float> host_rates;  // ... and the host's copy of those rates when they came from host memory
    bool host_rates_valid = false;
    uint32_t host_skips = 0;        // crowd calls whose morph pass the host-side comparison skipped (mmdx_debug_morph_pass_stats)
    bool capturing = false;         // between mmdx_graph_begin and mmdx_graph_end: the stream records
    std::thread::id capture_thread; // ... begun on this thread (thread-local capture mode: it must end there too)
    mmdx::GraphPin pin;                   // graphs that hold addresses of this model's scratch buffers
    std::vector<mmdx::GraphPin *> rec_pins;   // handles that took part in the recording in progress (incl. this model)
    bool rec_poisoned = false;          // one of them was destroyed before mmdx_graph_end: the recording cannot become a graph
    bool last_write_through = false;          // store flavour of the last crowd launch (mmdx_debug_last_store_policy)
    mmdx_model_s() {
        for (mmdx::DevBuf *b : {&pal, &rates, &wslot, &morphed, &out_a, &out_b}) b->pin = &pin;
    }
    // page-locked bounce buffer for small outputs bound for pageable host memory (see mmdx_deform_batched)
    void *bounce = nullptr, *bounce_dev = nullptr;  // host address, device-side address
    size_t bounce_bytes = 0;
    void *bounce_in = nullptr;                     // the same for small pageable inputs (palette, rates)
};
