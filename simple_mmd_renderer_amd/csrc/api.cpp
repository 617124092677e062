// api.cpp -- the C ABI of include/mmdx.h over the HIP runtime.  Host side of the drop-in boundary:
// validates and compiles the model once (plan.cpp), keeps the static streams resident in HBM, and
// turns one mmdx_deform*() call into at most three launches on the handle's stream:
//     [flatten group morphs -> slot weights] -> [shared morph pass] -> deform (skin + write-out).
// There is NO CPU fallback: without a usable HIP device every compute entry point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "api_internal.hpp"

using namespace mmdx;

namespace {

// Device selection follows hipSetDevice's model: mmdx_device_select() binds the CALLING THREAD (one host thread per
// device may each select its own and create / run its models concurrently) and also becomes the default of
// threads that never selected one (a process that selects once in its main thread and works from a pool).
std::atomic<int> g_default_device{0};
thread_local int tl_device = -1;
int current_device() { return tl_device >= 0 ? tl_device : g_default_device.load(std::memory_order_relaxed); }
std::once_flag g_prepare_once[16];
hipError_t g_prepare_status[16];



// Streams of this thread that are recording a graph (hipStreamCaptureModeThreadLocal: begin and end happen on one thread,
// mmdx_graph_end enforces it): while > 0 nothing on this thread may allocate, copy from the host or wait.
thread_local int tl_recording_depth = 0;


template <typename T>
hipError_t upload(DevBuf &b, const std::vector<T> &v, uint64_t &total) {
    const size_t n = std::max<size_t>(v.size() * sizeof(T), 16);  // never a null stream pointer
    hipError_t e = b.ensure(n);
    if (e != hipSuccess) return e;
    total += n;
    if (!v.empty()) return hipMemcpy(b.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return hipSuccess;
}

}  // namespace


struct mmdx_graph_s {
    void *exec = nullptr;      // hipGraphExec_t
    void *stream = nullptr;    // the model's stream at recording time
    int device = 0;
    std::atomic<bool> valid{true};        // false once a handle it was recorded from has been destroyed
    std::vector<GraphPin *> pins;         // those handles (guarded by g_graph_mu)
};

namespace {
std::mutex g_graph_mu;          // graph <-> handle links (rare operations: record, destroy)

// Store flavour of a crowd launch (kernels.hip CopyFast), decided from the CALL alone -- no table of addresses, no state behind the
// boundary: the caller's hint first (mmdx_placement_info.store_flags hands it the probe's verdict for arrays from
// mmdx_crowd_output_alloc), then MMDX_STORE_WT=0 / 1 (A/B runs), then the default for arrays nothing is known about: outputs large
// enough to stream through the caches (>= 512 MB per call) are written through, because six plain allocations in seven are not in
// the fast store mode (expected cost of the wrong guess: 2 % on a fast pair against 4.6-5 % on the others).
bool write_through_for(size_t out_bytes, uint32_t flags) {
    if (flags & MMDX_OUT_STORES_WRITE_THROUGH) return true;
    if (flags & MMDX_OUT_STORES_CACHED) return false;
    const int env = launch_overrides().store_wt;
    if (env == 0 || env == 1) return env == 1;
    return out_bytes >= (size_t(512) << 20);
}
}
void mmdx::graph_note_handle(mmdx_model_s *model, GraphPin *pin) {
    if (!model || !model->capturing || !pin) return;
    std::lock_guard<std::mutex> lk(g_graph_mu);
    if (std::find(model->rec_pins.begin(), model->rec_pins.end(), pin) == model->rec_pins.end()) {
        model->rec_pins.push_back(pin);
        pin->recorders.push_back(model);
    }
}
void mmdx::graph_drop_handle(GraphPin *pin) {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (mmdx_model_s *rec : pin->recorders) {        // a recording in progress used this handle: it can no longer become a graph
        rec->rec_pins.erase(std::remove(rec->rec_pins.begin(), rec->rec_pins.end(), pin), rec->rec_pins.end());
        rec->rec_poisoned = true;
    }
    pin->recorders.clear();
    for (mmdx_graph_s *g : pin->graphs) {
        g->valid.store(false, std::memory_order_release);
        g->pins.erase(std::remove(g->pins.begin(), g->pins.end(), pin), g->pins.end());
    }
    pin->graphs.clear();
    pin->pins.store(0, std::memory_order_release);
}

namespace {

constexpr size_t kMaxProfiledCalls = 1 << 16;

// The device-side address of a host pointer that lies in page-locked, device-mapped memory; nullptr for
// pageable memory (and for device memory: callers pass that with the *_ON_DEVICE flags).  MMDX_HOST_DIRECT=0
// turns the direct path off (A/B against the staging copy).
bool host_direct_enabled() {
    static const bool enabled = [] { const char *e = std::getenv("MMDX_HOST_DIRECT"); return !(e && e[0] == '0'); }();
    return enabled;
}
// What a pointer handed over WITHOUT an *_ON_DEVICE flag is: pageable host memory (unknown to the runtime),
// page-locked host memory (`*mapped` = its device-side address; interior pointers are fine, the offset carries
// over), or device memory -- a caller's mistake that must not reach a CPU memcpy.
enum class PtrKind { Pageable, Mapped, Device };
PtrKind classify_pointer(const void *p, void **mapped) {
    *mapped = nullptr;
    hipPointerAttribute_t attr;
    if (!p || hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();                    // pageable memory: "invalid value", not an error of ours
        return PtrKind::Pageable;
    }
    if (attr.type == hipMemoryTypeDevice) return PtrKind::Device;
    if (attr.type != hipMemoryTypeHost) return PtrKind::Pageable;
    if (hipHostGetDevicePointer(mapped, const_cast<void *>(p), 0) != hipSuccess) {
        (void)hipGetLastError();
        *mapped = nullptr;
        return PtrKind::Pageable;
    }
    return PtrKind::Mapped;
}
void *mapped_host_pointer(const void *host) {
    void *dev = nullptr;
    if (!host_direct_enabled()) return nullptr;
    (void)classify_pointer(host, &dev);
    return dev;
}

// Host-to-device copy of a per-call input.  Small pageable inputs (one frame's palette, its rates) are first
// copied by the CPU into the model's page-locked bounce buffer -- slot `slot_off` of it, the palette and the rates
// use different halves -- so the copy command is a plain DMA instead of the runtime's pageable path.  The previous
// call's copy out of the same slot has completed: every call with host inputs waits on the stream before it returns.
constexpr size_t kBounceInBytes = size_t(256) << 10;
hipError_t copy_in(mmdx_model_s *m, void *dst, const void *src, PtrKind kind, size_t bytes, size_t slot_off,
                   hipStream_t st) {
    if (!bytes) return hipSuccess;
    if (bytes <= kBounceInBytes / 2 && host_direct_enabled() && kind == PtrKind::Pageable) {
        if (!m->bounce_in && hipHostMalloc(&m->bounce_in, kBounceInBytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            m->bounce_in = nullptr;
        }
        if (m->bounce_in) {
            void *slot = static_cast<unsigned char *>(m->bounce_in) + slot_off;
            std::memcpy(slot, src, bytes);
            return hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, st);
        }
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
}



mmdx_status upload_model(mmdx_model_s *m) {
    Plan &p = m->plan;
    uint64_t &t = m->device_bytes;
    HIP_TRY(upload(m->tiles, p.tiles, t));
    if (p.f16) HIP_TRY(upload(m->spos, p.spos16, t)); else HIP_TRY(upload(m->spos, p.spos, t));
    HIP_TRY(upload(m->snrm, p.snrm, t));
    HIP_TRY(upload(m->suv, p.suv, t));
    HIP_TRY(upload(m->perm, p.perm, t));
    HIP_TRY(upload(m->skin1, p.skin1, t));
    HIP_TRY(upload(m->skin2_ids, p.skin2_ids, t));
    HIP_TRY(upload(m->skin2_w, p.skin2_w, t));
    HIP_TRY(upload(m->skin4_ids, p.skin4_ids, t));
    HIP_TRY(upload(m->skin4_w, p.skin4_w, t));
    HIP_TRY(upload(m->bone_list, p.bone_list, t));
    HIP_TRY(upload(m->ell, p.ell, t));
    if (p.f16) HIP_TRY(upload(m->entries, p.entries16, t)); else HIP_TRY(upload(m->entries, p.entries, t));
    HIP_TRY(upload(m->slot_top, p.slot_top, t));
    HIP_TRY(upload(m->chain_off, p.chain_off, t));
    HIP_TRY(upload(m->chain_rate, p.chain_rate, t));
    if (p.ns) {
        HIP_TRY(m->morphed.ensure(size_t(p.nv) * 12));
        t += size_t(p.nv) * 12;
        // which rates `morphed` was last computed from (kernels.hpp, RatesSeen): nothing yet
        const size_t seen_bytes = (size_t(kSeenRates) + p.nm) * 4;
        HIP_TRY(m->seen.ensure(seen_bytes));
        HIP_TRY(hipMemset(m->seen.ptr, 0, seen_bytes));
        HIP_TRY(hipStreamSynchronize(nullptr));
        t += seen_bytes;
    }
    return MMDX_OK;
}

void free_model(mmdx_model_s *m) {
    graph_drop_handle(&m->pin);              // graphs recorded from this model hold addresses that are about to be freed
    if (m->device >= 0) {
        (void)hipSetDevice(m->device);
        for (DevBuf *b : {&m->tiles, &m->spos, &m->snrm, &m->suv, &m->perm, &m->skin1, &m->skin2_ids,
                          &m->skin2_w, &m->skin4_ids, &m->skin4_w, &m->bone_list, &m->ell,
                          &m->entries, &m->slot_top, &m->chain_off, &m->chain_rate, &m->pal, &m->rates,
                          &m->wslot, &m->morphed, &m->seen, &m->out_a, &m->out_b})
            b->release();
        if (m->bounce) (void)hipHostFree(m->bounce);
        if (m->bounce_in) (void)hipHostFree(m->bounce_in);
        for (hipEvent_t ev : {m->ev_t0, m->ev_t1})
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : m->prof_events) (void)hipEventDestroy(ev);
        if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    }
    delete m;
}

size_t out_bytes_a(uint32_t layout, uint64_t nvi) {
    return size_t(layout == MMDX_OUT_SOA ? nvi * 12 : (layout == MMDX_OUT_VERTEX32 ? nvi * 32 : nvi * 6));
}
size_t out_bytes_b(uint32_t layout, uint64_t nvi) {
    return size_t(layout == MMDX_OUT_VERTEX32 ? 0 : nvi * 12);
}

}  // namespace


// ---- shared with bench_api.cpp (api_internal.hpp) -----------------------------------------------------------------------
mmdx_status mmdx::hip_fail(hipError_t e, const char *what) {
    // leave no sticky error behind for the next call
    (void)hipGetLastError();
    // the two refusals of DevBuf::ensure / rig_api's Buf::ensure (no HIP call failed)
    if (e == hipErrorIllegalState)
        return fail(MMDX_ERR_INVALID_ARGUMENT, std::string(what) + ": a scratch buffer of this handle would have to grow, but a recorded "
                    "graph (mmdx_graph_*) holds its address -- destroy the graph first, or size the buffers with an un-recorded call of "
                    "the largest shape before recording");
    if (e == hipErrorStreamCaptureUnsupported)
        return fail(MMDX_ERR_INVALID_ARGUMENT, std::string(what) + ": this call would have to allocate device memory while a graph is "
                    "being recorded -- run the same sequence once un-recorded first");
    return fail(e == hipErrorOutOfMemory ? MMDX_ERR_OUT_OF_MEMORY
                                         : (e == hipErrorNoDevice ? MMDX_ERR_NO_DEVICE : MMDX_ERR_HIP),
                std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")");
}

int mmdx::env_int(const char *name, int dflt) {
    const char *s = std::getenv(name);
    return s && *s ? std::atoi(s) : dflt;
}

LaunchOverrides mmdx::read_launch_overrides() {
    return {env_int("MMDX_INTERLEAVE", 1), env_int("MMDX_THREADS", 0), env_int("MMDX_LDS_TARGET", 0),
            env_int("MMDX_GROUP", 0), env_int("MMDX_PLACEMENT_LOG", 0), env_int("MMDX_PLACEMENT_PARK", 0),
            env_int("MMDX_FRAME_KERNEL", 1), env_int("MMDX_FRAME_THREADS", 256), env_int("MMDX_SHARED_FUSED", 1),
            env_int("MMDX_STORE_WT", -1), env_int("MMDX_MORPH_AUTOSKIP", 1), env_int("MMDX_FUSED_PACK", 0), env_int("MMDX_STAGGER", 0)};
}
LaunchOverrides &mmdx::launch_overrides() {
    static LaunchOverrides o = read_launch_overrides();
    return o;
}

void mmdx::morph_motion_release_device(MorphMotionDevice &d) {
    graph_drop_handle(&d.pin);               // graphs that hold these addresses can no longer be replayed
    if (d.device >= 0) (void)hipSetDevice(d.device);
    for (void **p : {&d.key_off, &d.frames, &d.weights, &d.frames_in, &d.out}) {
        device_free_or_defer(*p);
        *p = nullptr;
    }
    d.frames_in_bytes = d.out_bytes = 0;
    d.device = -1;
}

// A stream that is recording takes calls from the recording thread only: the "nothing may allocate / copy from the host / wait"
// guards are per thread (thread-local capture mode), a call from another thread would slip past them.
static mmdx_status recording_thread_check(mmdx_model_t model) {
    if (model && model->capturing && model->capture_thread != std::this_thread::get_id())
        return fail(MMDX_ERR_INVALID_ARGUMENT, "this model's stream is recording a graph on another thread: recorded calls must come "
                                               "from the thread that called mmdx_graph_begin");
    return MMDX_OK;
}

mmdx_status mmdx::resolve_stream(mmdx_model_t model, int *device, hipStream_t *stream) {
    if (mmdx_status st = recording_thread_check(model)) return st;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MMDX_ERR_NO_DEVICE, "no HIP device available (motion and rig evaluation run on the GPU)");
    const bool on_model = model && model->device >= 0;
    *device = on_model ? model->device : current_device();
    *stream = on_model ? model->stream : nullptr;
    HIP_TRY(hipSetDevice(*device));
    return MMDX_OK;
}
mmdx_status mmdx::hip_status(hipError_t e, const char *what) { return hip_fail(e, what); }
bool mmdx::graph_recording() { return tl_recording_depth > 0; }
// hipFree is one of the calls the runtime refuses on a thread whose stream is recording (it would also invalidate the recording):
// a handle destroyed in that window parks its blocks here, mmdx_graph_end frees them.
static thread_local std::vector<void *> tl_deferred_free;
// ... and so are the stream / event / page-locked-memory calls of a whole handle's teardown: a model destroyed on a thread that is
// recording (another model's graph) is only cut off from its graphs here; its resources go at mmdx_graph_end.
static thread_local std::vector<mmdx_model_s *> tl_deferred_models;
void mmdx::device_free_or_defer(void *ptr) {
    if (!ptr) return;
    if (tl_recording_depth > 0) tl_deferred_free.push_back(ptr);
    else (void)hipFree(ptr);
}

// The wait at the end of a call that hands results back to the host.  A per-frame call is tens of
// microseconds of device work; hipStreamSynchronize may put the thread to sleep and then pays a wake-up that is
// several times that on some hosts (tools/archive/probes/host_io_probe.py), so poll the stream first and only fall back to the
// blocking wait when the work is long.  MMDX_SPIN_WAIT_US: polling budget in microseconds (default 2000, 0 = off).
hipError_t mmdx::wait_stream(hipStream_t stream) {
    static const int spin_us = env_int("MMDX_SPIN_WAIT_US", 2000);
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(stream);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
    }
    return hipStreamSynchronize(stream);
}

extern "C" {

uint32_t mmdx_abi_version(void) { return MMDX_ABI_VERSION; }

mmdx_status mmdx_device_count(int32_t *count) {
    if (!count) return fail(MMDX_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return MMDX_OK;
}

mmdx_status mmdx_device_select(int32_t ordinal) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (ordinal < 0 || ordinal >= n || ordinal >= 16)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(ordinal));
    tl_device = ordinal;
    g_default_device.store(ordinal, std::memory_order_relaxed);
    return MMDX_OK;
}

mmdx_status mmdx_device_name(int32_t ordinal, char *buf, size_t buf_size) {
    if (!buf || !buf_size) return fail(MMDX_ERR_INVALID_ARGUMENT, "buf is NULL");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
    std::snprintf(buf, buf_size, "%s %s cus=%d lds=%zu l2=%d memclk=%d buswidth=%d", prop.name,
                  prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock,
                  prop.l2CacheSize, prop.memoryClockRate, prop.memoryBusWidth);
    return MMDX_OK;
}

mmdx_status mmdx_model_create(const mmdx_model_desc *desc, mmdx_model_t *out_model) {
    if (!desc || !out_model) return fail(MMDX_ERR_INVALID_ARGUMENT, "desc / out_model is NULL");
    *out_model = nullptr;
    if (desc->struct_size == sizeof(mmdx_model_desc) &&
        (desc->flags & ~uint32_t(MMDX_CREATE_NORMALIZE | MMDX_CREATE_HOST_ONLY | MMDX_CREATE_F16_POSITIONS | MMDX_CREATE_FAST_MATH |
                                 MMDX_CREATE_TILE_ORDER)))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "unknown bits in mmdx_model_desc.flags");
    mmdx_model_s *m = new (std::nothrow) mmdx_model_s;
    if (!m) return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    std::string err;
    mmdx_status st;
    try {
        st = build_plan(*desc, m->plan, err);
    } catch (const std::bad_alloc &) {
        delete m;
        return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while compiling the model");
    }
    if (st != MMDX_OK) {
        delete m;
        return fail(st, err);
    }
    if (desc->flags & MMDX_CREATE_HOST_ONLY) {
        *out_model = m;
        return MMDX_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        delete m;
        return fail(MMDX_ERR_NO_DEVICE,
                    "no HIP device available (this engine has no CPU fallback; use "
                    "MMDX_CREATE_HOST_ONLY to validate a model without a GPU)");
    }
    m->device = current_device();
    auto bail = [&](mmdx_status s) { free_model(m); return s; };
    if ((e = hipSetDevice(m->device)) != hipSuccess) return bail(hip_fail(e, "hipSetDevice"));
    std::call_once(g_prepare_once[m->device], [&] {
        hipError_t pe = prepare_kernels();
        g_prepare_status[m->device] = pe != hipSuccess ? pe : prepare_kernels_fast();
    });
    if (g_prepare_status[m->device] != hipSuccess)
        return bail(hip_fail(g_prepare_status[m->device], "hipFuncSetAttribute(dynamic LDS)"));
    if ((e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(hip_fail(e, "hipStreamCreate"));
    m->stream = m->own_stream;
    for (hipEvent_t *ev : {&m->ev_t0, &m->ev_t1})
        if ((e = hipEventCreate(ev)) != hipSuccess) return bail(hip_fail(e, "hipEventCreate"));
    st = upload_model(m);
    if (st != MMDX_OK) return bail(st);
    *out_model = m;
    return MMDX_OK;
}

mmdx_status mmdx_model_destroy(mmdx_model_t model) {
    if (!model) return MMDX_OK;
    if (model->capturing) {              // destroyed in the middle of a recording: end it here so that no state is left behind
        if (model->capture_thread != std::this_thread::get_id())
            return fail(MMDX_ERR_INVALID_ARGUMENT, "this model is recording a graph on another thread: end the recording there first");
        mmdx_graph_t dropped = nullptr;
        (void)mmdx_graph_end(model, &dropped);
        mmdx_graph_destroy(dropped);
    }
    if (tl_recording_depth > 0 && model->device >= 0) {
        // this thread is (still) recording another model's graph: no wait, no stream / event / host-memory call now
        graph_drop_handle(&model->pin);
        tl_deferred_models.push_back(model);
        return MMDX_OK;
    }
    if (model->device >= 0) {
        (void)hipSetDevice(model->device);
        (void)hipStreamSynchronize(model->stream);
    }
    free_model(model);
    return MMDX_OK;
}

mmdx_status mmdx_model_get_vertex_order(mmdx_model_t m, uint32_t *engine_to_original, uint32_t *original_to_engine) {
    if (!m) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL");
    const Plan &p = m->plan;
    for (const TileHdr &t : p.tiles)
        for (uint32_t s = 0; s < t.nv; ++s) {
            const uint32_t e = t.v0 + s, o = t.v0 + p.perm[e];
            if (engine_to_original) engine_to_original[e] = o;
            if (original_to_engine) original_to_engine[o] = e;
        }
    return MMDX_OK;
}

mmdx_status mmdx_model_get_info(mmdx_model_t m, mmdx_model_info *info) {
    if (!m || !info) return fail(MMDX_ERR_INVALID_ARGUMENT, "model / info is NULL");
    if (info->struct_size != sizeof(mmdx_model_info))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_model_info.struct_size mismatch");
    const Plan &p = m->plan;
    info->n_vertices = p.nv; info->n_bones = p.nb; info->n_morphs = p.nm;
    info->n_slots = p.ns; info->n_entries = p.ne_real; info->n_entries_padded = p.ne;
    info->n_tiles = p.ntiles; info->tile_vertices = kTileVerts;
    info->n_bdef1 = p.n1; info->n_bdef2 = p.n2; info->n_bdef4 = p.n4;
    info->max_tile_bones = p.max_tile_bones;
    info->device_bytes = m->device_bytes;
    info->device_ordinal = m->device < 0 ? 0xffffffffu : uint32_t(m->device);
    info->flags = p.flags;
    return MMDX_OK;
}

mmdx_status mmdx_model_get_skin(mmdx_model_t m, int32_t *type, int32_t *ids, float *weights) {
    if (!m || !type || !ids || !weights) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    const Plan &p = m->plan;
    for (uint32_t i = 0; i < p.nv; ++i)
        type[i] = p.cls[i] == 0 ? MMDX_SKIN_BDEF1 : (p.cls[i] == 1 ? MMDX_SKIN_BDEF2 : MMDX_SKIN_BDEF4);
    std::memcpy(ids, p.ids.data(), size_t(p.nv) * 16);
    std::memcpy(weights, p.wts.data(), size_t(p.nv) * 16);
    return MMDX_OK;
}

mmdx_status mmdx_model_slot_weights(mmdx_model_t m, const float *rates, float *out) {
    if (!m || (m->plan.ns && (!rates || !out))) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    flatten_slot_weights(m->plan, rates, out);
    return MMDX_OK;
}

mmdx_status mmdx_model_set_stream(mmdx_model_t m, void *hip_stream) {
    if (!m) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL");
    if (m->device < 0) return fail(MMDX_ERR_NO_DEVICE, "host-only model");
    m->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->own_stream;
    return MMDX_OK;
}

mmdx_status mmdx_deform_batched(mmdx_model_t m, const mmdx_deform_args *a) {
    if (!m || !a) return fail(MMDX_ERR_INVALID_ARGUMENT, "model / args is NULL");
    if (a->struct_size != sizeof(mmdx_deform_args))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_deform_args.struct_size mismatch");
    if (m->device < 0)
        return fail(MMDX_ERR_NO_DEVICE, "model was created with MMDX_CREATE_HOST_ONLY: nothing to run on "
                                        "(this engine has no CPU fallback)");
    const Plan &p = m->plan;
    const uint32_t ni = a->n_instances, layout = a->out_layout;
    if (ni == 0) return fail(MMDX_ERR_INVALID_ARGUMENT, "n_instances must be >= 1");
    if (a->flags & ~uint32_t(MMDX_PALETTE_ON_DEVICE | MMDX_WEIGHTS_ON_DEVICE | MMDX_OUT_ON_DEVICE | MMDX_WEIGHTS_SHARED | MMDX_MORPH_UNCHANGED |
                             MMDX_OUT_STORES_WRITE_THROUGH | MMDX_OUT_STORES_CACHED))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "unknown bits in mmdx_deform_args.flags");
    if ((a->flags & MMDX_OUT_STORES_WRITE_THROUGH) && (a->flags & MMDX_OUT_STORES_CACHED))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "MMDX_OUT_STORES_WRITE_THROUGH and MMDX_OUT_STORES_CACHED exclude each other");
    if (layout > MMDX_OUT_SOA_POS16) return fail(MMDX_ERR_INVALID_ARGUMENT, "unknown out_layout");
    if (p.f16 != (layout == MMDX_OUT_SOA_POS16))
        return fail(MMDX_ERR_UNSUPPORTED, "MMDX_OUT_SOA_POS16 goes with MMDX_CREATE_F16_POSITIONS models "
                                          "(and only with them)");
    if (!a->palettes || !a->out_a || (layout != MMDX_OUT_VERTEX32 && !a->out_b))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "palettes / out_a / out_b is NULL");
    if (p.ns && !a->morph_weights && !(a->flags & MMDX_MORPH_UNCHANGED))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "morph_weights is NULL");
    // host arguments: what kind of memory they are (device memory without its *_ON_DEVICE flag is a caller's
    // mistake that would otherwise end in a CPU memcpy from / to a device address)
    void *map_a = nullptr, *map_b = nullptr, *map_unused = nullptr;
    PtrKind kind_pal = PtrKind::Pageable, kind_w = PtrKind::Pageable, kind_a = PtrKind::Pageable, kind_b = PtrKind::Pageable;
    if (!(a->flags & MMDX_PALETTE_ON_DEVICE) && (kind_pal = classify_pointer(a->palettes, &map_unused)) == PtrKind::Device)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "palettes points to device memory: pass MMDX_PALETTE_ON_DEVICE");
    if (p.ns && !(a->flags & MMDX_WEIGHTS_ON_DEVICE) &&
        (kind_w = classify_pointer(a->morph_weights, &map_unused)) == PtrKind::Device)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "morph_weights points to device memory: pass MMDX_WEIGHTS_ON_DEVICE");
    if (!(a->flags & MMDX_OUT_ON_DEVICE)) {
        kind_a = classify_pointer(a->out_a, &map_a);
        if (layout != MMDX_OUT_VERTEX32) kind_b = classify_pointer(a->out_b, &map_b);
        if (kind_a == PtrKind::Device || kind_b == PtrKind::Device)
            return fail(MMDX_ERR_INVALID_ARGUMENT, "out_a / out_b points to device memory: pass MMDX_OUT_ON_DEVICE");
        if (!host_direct_enabled()) map_a = map_b = nullptr;
    }
    if (m->capturing) {
        if (mmdx_status rst = recording_thread_check(m)) return rst;
        const uint32_t need = MMDX_PALETTE_ON_DEVICE | MMDX_OUT_ON_DEVICE | (p.ns ? uint32_t(MMDX_WEIGHTS_ON_DEVICE) : 0u);
        if ((a->flags & need) != need)
            return fail(MMDX_ERR_INVALID_ARGUMENT, "while a graph is being recorded every operand must be in device memory");
        if (m->profile) return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_profile_enable and graph recording exclude each other");
    }
    const bool shared = (a->flags & MMDX_WEIGHTS_SHARED) != 0 || ni == 1;
    const bool fast = (p.flags & MMDX_CREATE_FAST_MATH) != 0;      // contracted multiply-adds: this model opted out of bit-exactness
    const uint64_t nvi = uint64_t(ni) * p.nv;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    hipEvent_t *pev = nullptr;  // {skin0, skin1, morph0, morph1} of this call when profiling
    // a full recording (kMaxProfiledCalls without mmdx_profile_collect) stops recording; it never fails the call
    if (m->profile && m->prof_calls < kMaxProfiledCalls && m->prof_seen++ % m->profile_stride == 0) {
        while (m->prof_events.size() < 4 * (m->prof_calls + 1)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            m->prof_events.push_back(ev);
        }
        pev = m->prof_events.data() + 4 * m->prof_calls;
    }

    DeformParams dp;
    std::memset(&dp, 0, sizeof(dp));
    dp.tiles = static_cast<const TileHdr *>(m->tiles.ptr);
    dp.spos = m->spos.ptr;
    dp.snrm = static_cast<const float *>(m->snrm.ptr);
    dp.suv = static_cast<const float *>(m->suv.ptr);
    dp.perm = static_cast<const uint16_t *>(m->perm.ptr);
    dp.skin1 = static_cast<const uint16_t *>(m->skin1.ptr);
    dp.skin2_ids = static_cast<const uint32_t *>(m->skin2_ids.ptr);
    dp.skin2_w = static_cast<const float *>(m->skin2_w.ptr);
    dp.skin4_ids = static_cast<const uint2 *>(m->skin4_ids.ptr);
    dp.skin4_w = static_cast<const float4 *>(m->skin4_w.ptr);
    dp.bone_list = static_cast<const uint32_t *>(m->bone_list.ptr);
    dp.ell = static_cast<const uint2 *>(m->ell.ptr);
    dp.entries = m->entries.ptr;
    dp.nv = p.nv; dp.nb = p.nb; dp.ns = p.ns; dp.ni = ni;
    dp.pos_scale = a->pos_scale;
    dp.pal_stride = p.max_tile_bones * 3;
    dp.finite_offsets = p.finite_offsets ? 1u : 0u;
    const LaunchOverrides &ov = launch_overrides();
    dp.interleave = uint32_t(ov.interleave);
    dp.tile_order = (p.flags & MMDX_CREATE_TILE_ORDER) ? 1u : 0u;

    // ---- palettes -------------------------------------------------------------------------------
    const size_t pal_bytes = size_t(ni) * p.nb * 64;
    if (a->flags & MMDX_PALETTE_ON_DEVICE) {
        if (reinterpret_cast<uintptr_t>(a->palettes) & 15)
            return fail(MMDX_ERR_INVALID_ARGUMENT, "device palettes must be 16-byte aligned");
        dp.palettes = a->palettes;
    } else {
        HIP_TRY(m->pal.ensure(pal_bytes));
        HIP_TRY(copy_in(m, m->pal.ptr, a->palettes, kind_pal, pal_bytes, 0, st));
        dp.palettes = static_cast<const float *>(m->pal.ptr);
    }

    // ---- morph mode + slot weights ----------------------------------------------------------------
    // Morph mode.  Shared rates: a single frame and SMALL crowds with one facial state gather the morphs inside the deform
    // kernel (every workgroup repeats its tile's walk; no separate launch: 8.8 vs 10.1 us for 2 instances of the 50k
    // model, break-even at 8, tools/archive/probes/shared_ab.py); larger crowds run the morph pass once, in front (257 vs 269 us for
    // 1024 instances) -- or not at all when the caller declares the rates unchanged since the last such call.
    bool unchanged = shared && ni > 1 && (a->flags & MMDX_MORPH_UNCHANGED);
    if (unchanged && !m->morphed_valid)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "MMDX_MORPH_UNCHANGED without an earlier MMDX_WEIGHTS_SHARED crowd call on this model");
    int morph = kMorphNone;
    const int sf = launch_overrides().shared_fused;          // 0: never for crowds, 1: small crowds (default), 2: always
    // The same without the caller's promise: vertex_images_ depends on morph_rates_ only (poser_impl.inl:362-386), so a crowd call
    // whose shared rates are, bit for bit, those of the morph pass whose result this handle still holds needs no morph pass.
    // Rates in HOST memory are compared here (memcmp with the copy kept of the last pass's), the launch and the upload are skipped;
    // rates in DEVICE memory are compared by morph_apply_kernel itself, which then skips its walk (RatesSeen, kernels.hpp).
    const bool autoskip = launch_overrides().morph_autoskip != 0;
    if (m->pin.replayed.exchange(false, std::memory_order_acq_rel)) m->host_rates_valid = false;   // a graph replay ran a morph pass
    const bool crowd_pass = p.ns && shared && ni > 1 && !(p.ns <= kMaxFusedSlots && (sf == 2 || (sf == 1 && ni <= 8)));
    const bool host_rates_call = crowd_pass && !unchanged && !(a->flags & MMDX_WEIGHTS_ON_DEVICE);
    if (host_rates_call && autoskip && m->morphed_valid && m->host_rates_valid && m->host_rates.size() == p.nm &&
        std::memcmp(m->host_rates.data(), a->morph_weights, size_t(p.nm) * 4) == 0) {
        unchanged = true;
        ++m->host_skips;
    }
    const bool gather_in_kernel = ni == 1 || (p.ns <= kMaxFusedSlots && !unchanged && (sf == 2 || (sf == 1 && ni <= 8)));
    if (p.ns) morph = shared ? (gather_in_kernel ? kMorphFused1 : kMorphShared) : kMorphFused4;
    if (morph != kMorphNone) {
        const uint32_t niw = shared ? 1u : ni;
        const float *rates_dev;
        if ((a->flags & MMDX_WEIGHTS_ON_DEVICE) || unchanged) {
            rates_dev = a->morph_weights;                     // (unchanged: never read)
        } else {
            HIP_TRY(m->rates.ensure(size_t(niw) * p.nm * 4));
            HIP_TRY(copy_in(m, m->rates.ptr, a->morph_weights, kind_w, size_t(niw) * p.nm * 4, kBounceInBytes / 2, st));
            rates_dev = static_cast<const float *>(m->rates.ptr);
        }
        FlattenParams f;
        f.rates = rates_dev;
        f.slot_top = static_cast<const uint32_t *>(m->slot_top.ptr);
        f.chain_off = static_cast<const uint32_t *>(m->chain_off.ptr);
        f.chain_rate = static_cast<const float *>(m->chain_rate.ptr);
        f.nm = p.nm; f.ns = p.ns; f.niw = niw;
        f.quad = morph == kMorphFused4 ? 1u : 0u;
        f.seen = nullptr;
        const size_t rows = f.quad ? size_t((niw + 3) / 4) * 4 : niw;
        HIP_TRY(m->wslot.ensure(rows * (size_t(p.ns) + 1) * 4));
        f.out = static_cast<float *>(m->wslot.ptr);
        if (pev) HIP_TRY(hipEventRecord(pev[2], st));
        dp.wslot = f.out;
        dp.morphed = static_cast<float *>(m->morphed.ptr);
        if (morph == kMorphShared && unchanged) {
            // nothing to launch: `morphed` holds the positions
        } else if (morph == kMorphShared && p.ns <= kMaxFusedSlots) {
            if (autoskip) f.seen = static_cast<uint32_t *>(m->seen.ptr);
            HIP_TRY((fast ? launch_morph_apply_fast : launch_morph_apply)(p.f16, dp, &f, st));      // flatten fused in: one launch
        } else if (morph == kMorphFused1) {
            if (ni > 1) dp.morph_seen = static_cast<uint32_t *>(m->seen.ptr);   // it overwrites `morphed`: the record is void after it
            // A single frame (ni == 1) leaves the model's kept positions alone: they belong to the last SHARED CROWD call
            // (MMDX_MORPH_UNCHANGED is documented against that one), whichever kernel the frame takes.
            if (ni == 1) dp.morphed = nullptr;
            dp.fused_rates = rates_dev;                           // flatten inside the deform kernel
            dp.slot_top = f.slot_top; dp.chain_off = f.chain_off; dp.chain_rate = f.chain_rate;
            dp.nm = p.nm;
        } else {
            HIP_TRY(launch_flatten(f, st));
            if (morph == kMorphShared) {
                HIP_TRY((fast ? launch_morph_apply_fast : launch_morph_apply)(p.f16, dp, nullptr, st));
                HIP_TRY(hipMemsetAsync(m->seen.ptr, 0, 4, st));       // this (rare: > 8192 slots) pass keeps no record of its rates
            }
        }
        // the host's own record: the rates of the pass `morphed` now holds, when they were host memory
        if (morph == kMorphShared && !unchanged) {
            m->host_rates_valid = host_rates_call;
            if (host_rates_call) m->host_rates.assign(a->morph_weights, a->morph_weights + p.nm);
        } else if (morph == kMorphFused1 && ni > 1) {
            m->host_rates_valid = false;
        }
        if (pev) HIP_TRY(hipEventRecord(pev[3], st));
        if (morph == kMorphShared || (morph == kMorphFused1 && ni > 1)) m->morphed_valid = true;   // kept by either path
    }

    // ---- outputs ---------------------------------------------------------------------------------
    const size_t bytes_a = out_bytes_a(layout, nvi), bytes_b = out_bytes_b(layout, nvi);
    const bool out_dev = (a->flags & MMDX_OUT_ON_DEVICE) != 0;
    // Host outputs in page-locked, device-mapped memory (mmdx_host_malloc / hipHostMalloc / hipHostRegister): the
    // kernel stores straight into them over PCIe -- 16-byte coalesced stores, overlapped with the skinning -- and
    // the staging buffer plus the device-to-host copy command (~10 us of fixed cost per frame) drop out.
    // Anything else goes through the staging buffer as before.
    // Small outputs bound for pageable memory (a frame of one model) take the same route into a page-locked
    // bounce buffer of the model's and are copied out by the CPU after the wait: the copy command into pageable
    // memory is the one piece of the per-frame call whose cost varies by 100 us between hosts.
    bool out_direct = false, out_bounce = false;
    constexpr size_t kBounceMax = size_t(4) << 20;
    const size_t off_b = (bytes_a + 63) & ~size_t(63);
    if (out_dev) {
        dp.out_a = a->out_a; dp.out_b = a->out_b;
    } else {
        void *da = map_a, *db = bytes_b ? map_b : nullptr;
        out_direct = da && (!bytes_b || db);
        if (!out_direct && off_b + bytes_b <= kBounceMax && host_direct_enabled()) {
            if (m->bounce_bytes < off_b + bytes_b) {
                if (m->bounce) (void)hipHostFree(m->bounce);
                m->bounce = nullptr; m->bounce_bytes = 0; m->bounce_dev = nullptr;
                if (hipHostMalloc(&m->bounce, kBounceMax, hipHostMallocDefault) == hipSuccess) m->bounce_bytes = kBounceMax;
                else (void)hipGetLastError();
            }
            if (m->bounce) {
                if (!m->bounce_dev) m->bounce_dev = mapped_host_pointer(m->bounce);
                da = m->bounce_dev;
                out_bounce = da != nullptr;
                if (out_bounce) db = static_cast<unsigned char *>(da) + off_b;
            }
        }
        if (out_direct || out_bounce) {
            dp.out_a = da; dp.out_b = bytes_b ? db : nullptr;
        } else {
            HIP_TRY(m->out_a.ensure(bytes_a));
            if (bytes_b) HIP_TRY(m->out_b.ensure(bytes_b));
            dp.out_a = m->out_a.ptr; dp.out_b = m->out_b.ptr;
        }
    }
    dp.out_aligned = ((reinterpret_cast<uintptr_t>(dp.out_a) | reinterpret_cast<uintptr_t>(dp.out_b)) & 15) == 0;

    // ---- workgroup shape ------------------------------------------------------------------------------
    // 256 threads / two vertex slots per lane everywhere except the per-instance-morph path: there one slot
    // per lane (512 threads) leaves the registers to serve 8 instances per walk over a morph row.
    // A single frame (one instance) is latency-bound: one slot per lane and twice the waves per tile finish sooner
    // (config 2: 8.4 -> 6.9 us, config 5: 16.9 -> 14.9 us).
    const bool one_frame = ni == 1 && (morph == kMorphNone || morph == kMorphFused1);
    // (tile-order outputs: no LDS image, 80 VGPRs with one slot per lane -- 512 threads measured 215.5 vs 219.1 us on the crowd)
    int threads = (ov.threads ? ov.threads : (morph == kMorphFused4 || one_frame || dp.tile_order ? 512 : 256)) == 512 ? 512 : 256;
    if (morph == kMorphFused4 && threads == 512) {   // tiles with hundreds of bones: 8 palettes do not fit, 4 may
        uint32_t so, wo;
        if (deform_lds_bytes(512, layout, morph, 8, p.max_tile_bones, p.ns, &so, &wo) > 160 * 1024) threads = 256;
    }
    // ---- store flavour: only where the launch shape has the write-through flavour, and only for outputs in device memory (stores
    // into mapped host memory cross PCIe whatever their cache bits say) ------------------------------------------------------------
    dp.write_through = out_dev && deform_has_write_through(threads, int(layout), morph, p.f16, dp.tile_order != 0) &&
                       write_through_for(bytes_a + bytes_b, a->flags) ? 1u : 0u;
    m->last_write_through = dp.write_through != 0;
    // ---- group size (instances per workgroup) from the LDS budget ---------------------------------
    const uint32_t gmin = morph == kMorphFused4 ? (threads == 512 ? 8u : 4u) : 1u;
    uint32_t group = gmin;
    if (morph != kMorphFused1 || ni > 1) {
        const uint32_t target = uint32_t(ov.lds_target ? ov.lds_target : (morph == kMorphFused4 ? 64 : 42) * 1024);
        uint32_t so, wo;
        const size_t fixed = deform_lds_bytes(threads, layout, morph, 0, p.max_tile_bones, p.ns, &so, &wo, dp.tile_order != 0);
        const size_t per = size_t(p.max_tile_bones) * 48;
        uint32_t g = target > fixed ? uint32_t((target - fixed) / per) : 0u;
        g = std::min(g, (morph == kMorphFused4 || dp.tile_order) ? 16u : 32u);   // tile order: 16 219 us, 32 229 us, 8 244 us
        if (g >= 8) g &= ~3u;   // measured: 16 beats 17 (even split of 1024 instances, aligned strides)
        // write-through stores go to arrays that are not in the fast store mode; there 8 instances per workgroup (four workgroups
        // per CU, half the open output streams each) beat 16 by 3-8 % -- 218-224 vs 225-241 us on four such pairs, while on a fast
        // pair 16 wins (204 vs 211): profiles/r03/shape_sweep_write_through*.txt
        if (dp.write_through) g = std::min(g, 8u);
        g = std::max(g / gmin * gmin, gmin);
        const uint32_t ni_up = (ni + gmin - 1) / gmin * gmin;
        g = std::min(g, ni_up);
        // keep the grid large enough to fill 256 CUs several times over
        while (g > gmin && uint64_t(p.ntiles) * ((ni + g - 1) / g) < 2048) {
            const uint32_t half = std::max((g / 2) / gmin * gmin, gmin);
            if (half == g) break;
            g = half;
        }
        group = std::max(g, gmin);
        const int forced = ov.group;
        if (forced > 0) group = std::max(uint32_t(forced) / gmin * gmin, gmin);
    }
    dp.group = group;
    // Per-instance morph weights, second shape (kernels.hip pack_kernel), OPT-IN (MMDX_FUSED_PACK=1): packs of 4 instances, 80 registers,
    // three 8-wave workgroups per CU while a workgroup's LDS stays under a third of the CU's.  It runs at 5.7 waves per SIMD where
    // deform_kernel<512, ., kMorphFused4> runs at 3.8 -- and loses (config 3' 372-382 us against 335-348; profiles/r04/fused_pack_*):
    // the walk and the skinning of one CU do not overlap in either kernel (walk alone 132 us + skinning alone 254 us), and packs of 4
    // walk the table twice as often as packs of 8.  Kept for the A/B, not the default.  Two-array layouts in original vertex order.
    // The group: as many instances (multiple of 4, up to 16) as keep three workgroups on a CU, else as fit two.
    bool pack = morph == kMorphFused4 && ov.fused_pack != 0 && kTileVerts == 512 && !dp.tile_order && layout != MMDX_OUT_VERTEX32 && !ov.threads;
    size_t lds = 0;
    if (pack) {
        uint32_t so, wo, mo;
        const size_t third = (160 * 1024) / 3 - 64, half = 80 * 1024 - 64;
        uint32_t g = 0;
        for (uint32_t c = 16; c >= 4 && !g; c -= 4)
            if (pack_lds_bytes(c, p.max_tile_bones, p.ns, &so, &wo, &mo) <= third) g = c;
        for (uint32_t c = 16; c >= 4 && !g; c -= 4)
            if (pack_lds_bytes(c, p.max_tile_bones, p.ns, &so, &wo, &mo) <= half) g = c;
        if (!g && pack_lds_bytes(4, p.max_tile_bones, p.ns, &so, &wo, &mo) <= 160 * 1024) g = 4;
        if (ov.group > 0) g = std::max(uint32_t(ov.group) / 4 * 4, 4u);
        if (g) {
            g = std::min(g, (ni + 3) / 4 * 4);
            while (g > 4 && uint64_t(p.ntiles) * ((ni + g - 1) / g) < 1536) g -= 4;     // keep 256 CUs x 3 workgroups busy twice over
            dp.group = g;
            lds = pack_lds_bytes(g, p.max_tile_bones, p.ns, &dp.stage_off, &dp.w_off, &dp.mp_off);
            if (lds > 160 * 1024) pack = false;
        } else {
            pack = false;
        }
    }
    if (!pack) {
        dp.group = group;
        lds = deform_lds_bytes(threads, layout, morph, group, p.max_tile_bones, p.ns, &dp.stage_off, &dp.w_off, dp.tile_order != 0);
    }
    if (lds > 160 * 1024)
        return fail(MMDX_ERR_UNSUPPORTED, "tile needs " + std::to_string(lds) + " bytes of LDS (> 160 KiB): "
                                          "too many distinct bones in one vertex tile / too many morph slots");

    if (morph == kMorphFused4) {
        dp.stagger = uint32_t(std::max(ov.stagger, 0));
        dp.slots_per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(uint32_t(160 * 1024 / std::max<size_t>(lds, 1)), pack ? 3u : 2u));
    }
    if (pev) HIP_TRY(hipEventRecord(pev[0], st));
    // One frame of one model into device memory: the latency-ordered kernel (parts of tiles on every CU, direct stores).
    // Outputs in mapped host memory keep the tile kernel: its 16-byte coalesced stores are what crosses PCIe well.
    // Models with fewer tiles than the chip has CUs only (config 2: 6.3 us against the tile kernel's 6.9); a large model fills the
    // chip with whole tiles and is better off with their coalesced stores (config 5: 14.9 us against 15.2).
    const bool frame = one_frame && !out_direct && !out_bounce && (ov.frame_kernel == 2 || (ov.frame_kernel == 1 && p.ntiles < 256));
    if (frame) {
        DeformParams fp = dp;
        fp.morphed = nullptr;                                  // nothing reads a single frame's morphed positions later
        const size_t flds = frame_lds_bytes(morph, p.max_tile_bones, p.ns, &fp.w_off);
        if (flds > 160 * 1024)
            return fail(MMDX_ERR_UNSUPPORTED, "tile needs " + std::to_string(flds) + " bytes of LDS (> 160 KiB)");
        HIP_TRY((fast ? launch_frame_fast : launch_frame)(ov.frame_threads, int(layout), morph, p.f16, fp, p.ntiles, flds, st));
    } else if (pack) {
        HIP_TRY((fast ? launch_pack_fast : launch_pack)(int(layout), p.f16, dp, p.ntiles, lds, st));
    } else {
        HIP_TRY((fast ? launch_deform_fast : launch_deform)(threads, int(layout), morph, p.f16, dp, p.ntiles, lds, st));
    }
    if (pev) {
        HIP_TRY(hipEventRecord(pev[1], st));
        if (m->prof_has_morph.size() <= m->prof_calls) m->prof_has_morph.resize(m->prof_calls + 1);
        m->prof_has_morph[m->prof_calls] = morph != kMorphNone;
        ++m->prof_calls;
    }

    if (!out_dev) {
        if (!out_direct && !out_bounce) {
            HIP_TRY(hipMemcpyAsync(a->out_a, dp.out_a, bytes_a, hipMemcpyDeviceToHost, st));
            if (bytes_b) HIP_TRY(hipMemcpyAsync(a->out_b, dp.out_b, bytes_b, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(wait_stream(st));
        if (out_bounce) {
            std::memcpy(a->out_a, m->bounce, bytes_a);
            if (bytes_b) std::memcpy(a->out_b, static_cast<unsigned char *>(m->bounce) + off_b, bytes_b);
        }
    } else if (!(a->flags & MMDX_PALETTE_ON_DEVICE) ||
               (morph != kMorphNone && !(a->flags & MMDX_WEIGHTS_ON_DEVICE))) {
        // borrowed host inputs must be consumed before we return
        HIP_TRY(wait_stream(st));
    }
    return MMDX_OK;
}

mmdx_status mmdx_deform(mmdx_model_t m, const float *w, const float *palette, float *out_pos,
                        float *out_nrm) {
    mmdx_deform_args a;
    std::memset(&a, 0, sizeof(a));
    a.struct_size = sizeof(a);
    a.n_instances = 1;
    a.out_layout = MMDX_OUT_SOA;
    a.morph_weights = w; a.palettes = palette;
    a.out_a = out_pos; a.out_b = out_nrm;
    a.pos_scale = 1.0f;
    return mmdx_deform_batched(m, &a);
}

mmdx_status mmdx_deform_vertex32(mmdx_model_t m, const float *w, const float *palette,
                                 float pos_scale, void *out_vertices) {
    mmdx_deform_args a;
    std::memset(&a, 0, sizeof(a));
    a.struct_size = sizeof(a);
    a.n_instances = 1;
    a.out_layout = MMDX_OUT_VERTEX32;
    a.morph_weights = w; a.palettes = palette;
    a.out_a = out_vertices;
    a.pos_scale = pos_scale;
    return mmdx_deform_batched(m, &a);
}

mmdx_status mmdx_sync(mmdx_model_t m) {
    if (!m) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL");
    if (m->device < 0) return fail(MMDX_ERR_NO_DEVICE, "host-only model");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(wait_stream(m->stream));
    return MMDX_OK;
}


// ---- VMD morph motion: device-side evaluation ----------------------------------------------------
mmdx_status mmdx_morph_motion_eval(mmdx_morph_motion_t mm, mmdx_model_t model, uint32_t n_instances,
                                   const uint32_t *frames, uint32_t flags, float *out_weights) {
    if (!mm || !frames || !out_weights || !n_instances)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or n_instances == 0");
    if (tl_recording_depth > 0 && (flags & (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE)) != (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "while a graph is being recorded every operand must be in device memory");
    if (mmdx_status rst = recording_thread_check(model)) return rst;
    const MorphMotionHost h = morph_motion_host(mm);
    MorphMotionDevice &d = morph_motion_device(mm);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MMDX_ERR_NO_DEVICE, "no HIP device available (morph tracks are evaluated on the GPU)");
    const int device = model && model->device >= 0 ? model->device : current_device();
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = model && model->device >= 0 ? model->stream : nullptr;
    if (d.device != device) {
        if (tl_recording_depth > 0) return fail(MMDX_ERR_INVALID_ARGUMENT, "first use of this motion on the device: run the sequence once before recording it");
        if (graph_pinned(&d.pin)) return fail(MMDX_ERR_INVALID_ARGUMENT, "this motion's device tables are held by a recorded graph: destroy the graph before moving it to another device");
        morph_motion_release_device(d);
        HIP_TRY(hipMalloc(&d.key_off, (size_t(h.nm) + 1) * 4));
        HIP_TRY(hipMalloc(&d.frames, std::max<size_t>(size_t(h.nkeys) * 4, 16)));
        HIP_TRY(hipMalloc(&d.weights, std::max<size_t>(size_t(h.nkeys) * 4, 16)));
        HIP_TRY(hipMemcpy(d.key_off, h.key_off, (size_t(h.nm) + 1) * 4, hipMemcpyHostToDevice));
        if (h.nkeys) {
            HIP_TRY(hipMemcpy(d.frames, h.frames, size_t(h.nkeys) * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d.weights, h.weights, size_t(h.nkeys) * 4, hipMemcpyHostToDevice));
        }
        d.device = device;
    }
    graph_note_handle(model, &d.pin);
    MorphTrackParams t;
    t.key_off = static_cast<const uint32_t *>(d.key_off);
    t.key_frames = static_cast<const uint32_t *>(d.frames);
    t.key_weights = static_cast<const float *>(d.weights);
    t.nm = h.nm; t.ni = n_instances;
    if (flags & MMDX_FRAMES_ON_DEVICE) {
        t.frames = frames;
    } else {
        if (d.frames_in_bytes < size_t(n_instances) * 4) {
            if (graph_pinned(&d.pin)) return hip_fail(hipErrorIllegalState, "morph motion frame scratch");
            if (d.frames_in) (void)hipFree(d.frames_in);
            d.frames_in = nullptr; d.frames_in_bytes = 0;
            HIP_TRY(hipMalloc(&d.frames_in, size_t(n_instances) * 4));
            d.frames_in_bytes = size_t(n_instances) * 4;
        }
        HIP_TRY(hipMemcpyAsync(d.frames_in, frames, size_t(n_instances) * 4, hipMemcpyHostToDevice, st));
        t.frames = static_cast<const uint32_t *>(d.frames_in);
    }
    const size_t out_bytes = size_t(n_instances) * h.nm * 4;
    if (flags & MMDX_OUT_ON_DEVICE) {
        t.out = out_weights;
    } else {
        if (d.out_bytes < out_bytes) {
            if (graph_pinned(&d.pin)) return hip_fail(hipErrorIllegalState, "morph motion output scratch");
            if (d.out) (void)hipFree(d.out);
            d.out = nullptr; d.out_bytes = 0;
            HIP_TRY(hipMalloc(&d.out, std::max<size_t>(out_bytes, 16)));
            d.out_bytes = std::max<size_t>(out_bytes, 16);
        }
        t.out = static_cast<float *>(d.out);
    }
    HIP_TRY(launch_morph_track_eval(t, st));
    if (!(flags & MMDX_OUT_ON_DEVICE)) {
        if (out_bytes) HIP_TRY(hipMemcpyAsync(out_weights, t.out, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(wait_stream(st));
    } else if (!(flags & MMDX_FRAMES_ON_DEVICE)) {
        HIP_TRY(wait_stream(st));   // borrowed host frames must be consumed before returning
    }
    return MMDX_OK;
}

mmdx_status mmdx_graph_begin(mmdx_model_t m) {
    if (!m || m->device < 0) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL or host-only");
    if (m->capturing) return fail(MMDX_ERR_INVALID_ARGUMENT, "this model is already recording a graph");
    if (m->profile) return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_profile_enable and graph recording exclude each other");
    HIP_TRY(hipSetDevice(m->device));
    if (tl_recording_depth > 0) {
        // a second recording on a thread that is already recording: drain this model's stream by polling (a blocking wait is one
        // of the calls a recording thread should not make)
        hipError_t q;
        while ((q = hipStreamQuery(m->stream)) == hipErrorNotReady) std::this_thread::yield();
        HIP_TRY(q);
    } else {
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    HIP_TRY(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    m->capturing = true;
    m->capture_thread = std::this_thread::get_id();
    m->rec_poisoned = false;
    {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        m->rec_pins.assign(1, &m->pin);
        m->pin.recorders.push_back(m);
    }
    ++tl_recording_depth;
    return MMDX_OK;
}

mmdx_status mmdx_graph_end(mmdx_model_t m, mmdx_graph_t *out) {
    if (!m || !out) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (!m->capturing) return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_graph_end without mmdx_graph_begin");
    // thread-local capture mode: the runtime wants the end on the thread that began, and the per-thread recording
    // state lives there -- refuse before anything is touched, so that the right thread can still end it
    if (m->capture_thread != std::this_thread::get_id())
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_graph_end must be called on the thread that called mmdx_graph_begin");
    m->capturing = false;
    --tl_recording_depth;
    std::vector<GraphPin *> used;
    bool poisoned;
    {   // the recording is over either way: its handles no longer count this model as a recorder
        std::lock_guard<std::mutex> lk(g_graph_mu);
        used.swap(m->rec_pins);
        for (GraphPin *p : used) p->recorders.erase(std::remove(p->recorders.begin(), p->recorders.end(), m), p->recorders.end());
        poisoned = m->rec_poisoned;
    }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(m->stream, &g);
    if (tl_recording_depth == 0) {                      // handles destroyed while this thread was recording, and their blocks
        std::vector<mmdx_model_s *> models;
        models.swap(tl_deferred_models);
        for (mmdx_model_s *dm : models) {
            if (dm->device >= 0) {
                (void)hipSetDevice(dm->device);
                (void)hipStreamSynchronize(dm->stream);
            }
            free_model(dm);
        }
        for (void *p : tl_deferred_free) (void)hipFree(p);
        tl_deferred_free.clear();
        (void)hipSetDevice(m->device);
    }
    if (poisoned) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return fail(MMDX_ERR_INVALID_ARGUMENT, "a skeleton or motion used by this recording was destroyed before mmdx_graph_end: "
                                               "the recording holds freed device addresses and is discarded");
    }
    if (e != hipSuccess || !g) return hip_fail(e != hipSuccess ? e : hipErrorUnknown, "hipStreamEndCapture (a recorded call failed?)");
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate");
    mmdx_graph_s *gr = new (std::nothrow) mmdx_graph_s;
    if (!gr) { (void)hipGraphExecDestroy(exec); return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    gr->exec = exec; gr->stream = m->stream; gr->device = m->device;
    {   // pin every handle whose buffers the recorded calls referred to
        std::lock_guard<std::mutex> lk(g_graph_mu);
        gr->pins = used;
        for (GraphPin *p : gr->pins) {
            p->graphs.push_back(gr);
            p->pins.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    *out = gr;
    return MMDX_OK;
}

mmdx_status mmdx_graph_launch(mmdx_graph_t g) {
    if (!g) return fail(MMDX_ERR_INVALID_ARGUMENT, "graph is NULL");
    if (!g->valid.load(std::memory_order_acquire))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "a model, skeleton or motion this graph was recorded from has been destroyed: "
                                               "the graph holds freed device addresses and cannot be replayed");
    {   // a replay may recompute a model's morphed positions from rates the host never saw: its host-side record of them is void
        std::lock_guard<std::mutex> lk(g_graph_mu);
        for (GraphPin *p : g->pins) p->replayed.store(true, std::memory_order_release);
    }
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(g->exec), static_cast<hipStream_t>(g->stream)));
    return MMDX_OK;
}

void mmdx_graph_destroy(mmdx_graph_t g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        for (GraphPin *p : g->pins) {
            p->graphs.erase(std::remove(p->graphs.begin(), p->graphs.end(), g), p->graphs.end());
            p->pins.fetch_sub(1, std::memory_order_acq_rel);
        }
        g->pins.clear();
    }
    (void)hipSetDevice(g->device);
    (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(g->exec));
    delete g;
}

mmdx_status mmdx_device_malloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(MMDX_ERR_INVALID_ARGUMENT, "ptr is NULL");
    HIP_TRY(hipSetDevice(current_device()));
    HIP_TRY(hipMalloc(ptr, bytes ? bytes : 16));
    return MMDX_OK;
}

mmdx_status mmdx_host_malloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(MMDX_ERR_INVALID_ARGUMENT, "ptr is NULL");
    HIP_TRY(hipSetDevice(current_device()));
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return MMDX_OK;
}

mmdx_status mmdx_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return MMDX_OK;
}

mmdx_status mmdx_device_free(void *ptr) {
    if (ptr) HIP_TRY(hipFree(ptr));
    return MMDX_OK;
}

mmdx_status mmdx_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MMDX_OK;
}

mmdx_status mmdx_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MMDX_OK;
}

mmdx_status mmdx_device_memset(void *dst, int value, size_t bytes) {
    // hipMemset of device memory is asynchronous to the host and runs on the null stream, which the models'
    // non-blocking streams do not wait for: complete it here, so that whatever the caller launches next (on any
    // stream) is ordered after it.
    HIP_TRY(hipMemset(dst, value, bytes));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return MMDX_OK;
}



mmdx_status mmdx_device_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return MMDX_OK;
}




mmdx_status mmdx_crowd_output_alloc(mmdx_model_t m, uint32_t n_instances, int32_t out_layout, uint32_t max_tries,
                                    void **out_a, void **out_b, mmdx_placement_info *info) {
    if (!m || !out_a || !out_b || !n_instances) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or n_instances == 0");
    if (m->device < 0) return fail(MMDX_ERR_NO_DEVICE, "host-only model");
    if (info && info->struct_size != sizeof(mmdx_placement_info))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_placement_info.struct_size mismatch");
    uint32_t bpva, bpvb;
    switch (out_layout) {
    case MMDX_OUT_SOA: bpva = 12; bpvb = 12; break;
    case MMDX_OUT_VERTEX32: bpva = 32; bpvb = 0; break;
    case MMDX_OUT_SOA_POS16: bpva = 6; bpvb = 12; break;
    default: return fail(MMDX_ERR_INVALID_ARGUMENT, "unknown out_layout");
    }
    *out_a = *out_b = nullptr;
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t nv = m->plan.nv;
    const size_t bytes_a = std::max<size_t>(size_t(n_instances) * nv * bpva, 16);
    const size_t bytes_b = bpvb ? std::max<size_t>(size_t(n_instances) * nv * bpvb, 16) : 0;
    // the replay needs every piece 16-byte aligned; otherwise (odd vertex counts) allocate without probing
    const bool can_probe = max_tries > 1 && nv && (size_t(nv) * bpva) % 16 == 0 && (size_t(nv) * bpvb) % 16 == 0;
    struct Cand { void *a = nullptr, *b = nullptr; float gbs = 0.f; };
    std::vector<Cand> parked;
    Cand best;
    float fill_gbs = 0.f;
    uint32_t tries = 0, slow_in_a_row = 0;
    mmdx_status st = MMDX_OK;
    auto release = [](Cand &c) { if (c.a) (void)hipFree(c.a); if (c.b) (void)hipFree(c.b); c.a = c.b = nullptr; };
    for (; tries < std::max(max_tries, 1u); ) {
        Cand c;
        hipError_t e = hipMalloc(&c.a, bytes_a);
        if (e == hipSuccess && bytes_b) e = hipMalloc(&c.b, bytes_b);
        if (e != hipSuccess) {
            release(c);
            (void)hipGetLastError();
            if (best.a) break;                       // out of memory while shopping: keep what we have
            st = hip_fail(e, "hipMalloc of the crowd output arrays");
            break;
        }
        ++tries;
        if (!can_probe) { best = c; break; }
        float ms = 0.f;
        if (fill_gbs == 0.f) {                       // the yardstick: a linear fill of array a
            hipEvent_t e0, e1;
            e = hipEventCreate(&e0);
            if (e == hipSuccess) e = hipEventCreate(&e1);
            if (e == hipSuccess) e = launch_fill(c.a, bytes_a & ~size_t(15), nullptr);
            if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
            for (int i = 0; i < 5 && e == hipSuccess; ++i) e = launch_fill(c.a, bytes_a & ~size_t(15), nullptr);
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            if (e == hipSuccess && ms > 0.f) fill_gbs = float(double(bytes_a) * 5 / (ms * 1e-3) / 1e9);
        }
        if (e == hipSuccess) e = time_store_pattern(c.a, c.b, nv, n_instances, bpva, bpvb, 5, &ms);
        if (e != hipSuccess) { release(c); st = hip_fail(e, "probing a placement of the crowd output arrays"); break; }
        c.gbs = float(double(bytes_a + bytes_b) / (ms * 1e-3) / 1e9);
        if (launch_overrides().placement_log)
            std::fprintf(stderr, "mmdx placement try %u: a=%p b=%p store %.0f GB/s (fill %.0f)\n", tries, c.a, c.b, c.gbs, fill_gbs);
        // measured (tools/archive/probes/shop_probe.py): about one placement in seven is fast either way; freeing a rejected
        // candidate at once needs a few tries less on average than keeping it parked, and no extra memory.  But a process can
        // get STUCK that way -- freed blocks come straight back, 128 tries in a row in the same slow backing (seen once in the
        // round-2 profile runs: 0.279 instead of 0.227 ms per step) -- so after 8 slow tries in a row the rejected candidates
        // are kept until the end (up to 48 pairs, ~60 GB of the 288): every further try then draws fresh physical memory.
        if (c.gbs < 0.92f * fill_gbs) ++slow_in_a_row; else slow_in_a_row = 0;
        const bool park = launch_overrides().placement_park != 0 || (slow_in_a_row >= 8 && parked.size() < 48);
        if (c.gbs > best.gbs) {
            if (best.a) { if (park) parked.push_back(best); else release(best); }
            best = c;
        } else if (park) {
            parked.push_back(c);
        } else {
            release(c);
        }
        if (best.gbs >= 0.92f * fill_gbs) break;     // the fast mode sits at 0.96-0.99 of the fill rate, the slow ones at 0.72-0.8
    }
    const bool freed_any = tries > 1;
    for (Cand &c : parked) release(c);
    if (st != MMDX_OK) { release(best); return st; }
    // The driver wipes freed VRAM in the background (gigabytes per rejected try), which steals HBM bandwidth
    // from whatever runs next: replay the pattern on the chosen arrays until its rate is back where it was
    // measured, so the caller's first launches already see a quiet memory system (bounded: ~0.25 s).
    for (int i = 0; freed_any && can_probe && i < 150; ++i) {
        float ms = 0.f;
        if (time_store_pattern(best.a, best.b, nv, n_instances, bpva, bpvb, 5, &ms) != hipSuccess) break;
        if (double(bytes_a + bytes_b) / (ms * 1e-3) / 1e9 >= 0.98 * best.gbs) break;
    }
    HIP_TRY(hipDeviceSynchronize());
    *out_a = best.a;
    *out_b = best.b;
    if (info) {
        info->tries = tries;
        info->probed = can_probe ? 1u : 0u;
        info->store_GBs = best.gbs;
        info->fill_GBs = fill_gbs;
        // the probe's verdict, for the caller to pass on in mmdx_deform_args.flags (nothing is remembered here)
        info->store_flags = can_probe && fill_gbs > 0.f ? (best.gbs >= 0.92f * fill_gbs ? uint32_t(MMDX_OUT_STORES_CACHED)
                                                                                         : uint32_t(MMDX_OUT_STORES_WRITE_THROUGH))
                                                        : 0u;
    }
    return MMDX_OK;
}

}  // extern "C"
