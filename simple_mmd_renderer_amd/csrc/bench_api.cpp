// bench_api.cpp -- the entry points of include/mmdx_bench.h: HIP-event timers on a model's stream, per-kernel profiling of
// mmdx_deform_batched (the events themselves are recorded by api.cpp's deform call), the streaming copy / fill / store-pattern
// ceilings printed next to the roofline, and the re-read of the launch-shape override environment.  NOT part of the drop-in
// boundary: nothing in the reference corresponds to these (SURVEY.md 8b lists four calls); they serve bench.py, tools/ and tests.
#include "api_internal.hpp"

using namespace mmdx;

// average ms of `iters` store-pattern launches on the default stream (after one warm-up launch)
hipError_t mmdx::time_store_pattern(void *a, void *b, uint32_t nv, uint32_t ni, uint32_t bpva, uint32_t bpvb, int iters,
                              float *avg_ms) {
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return e; }
    e = launch_pattern_fill(a, b, nv, ni, bpva, bpvb, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_pattern_fill(a, b, nv, ni, bpva, bpvb, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / float(iters);
    return e;
}

extern "C" {

mmdx_status mmdx_timer_start(mmdx_model_t m) {
    if (!m || m->device < 0) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL or host-only");
    HIP_TRY(hipEventRecord(m->ev_t0, m->stream));
    return MMDX_OK;
}

mmdx_status mmdx_timer_stop(mmdx_model_t m, float *ms) {
    if (!m || m->device < 0 || !ms) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL or host-only");
    HIP_TRY(hipEventRecord(m->ev_t1, m->stream));
    HIP_TRY(hipEventSynchronize(m->ev_t1));
    HIP_TRY(hipEventElapsedTime(ms, m->ev_t0, m->ev_t1));
    return MMDX_OK;
}

mmdx_status mmdx_profile_enable(mmdx_model_t m, int32_t enabled) {
    if (!m || m->device < 0) return fail(MMDX_ERR_INVALID_ARGUMENT, "model is NULL or host-only");
    m->profile = enabled != 0;
    m->profile_stride = enabled > 1 ? uint32_t(enabled) : 1u;
    m->prof_seen = 0;
    m->prof_calls = 0;
    return MMDX_OK;
}

mmdx_status mmdx_profile_collect(mmdx_model_t m, uint32_t *n_calls, float *skin_ms_total,
                                 float *morph_ms_total) {
    if (!m || m->device < 0 || !n_calls || !skin_ms_total || !morph_ms_total)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or host-only model");
    double skin = 0.0, morph = 0.0;
    for (size_t c = 0; c < m->prof_calls; ++c) {
        hipEvent_t *ev = m->prof_events.data() + 4 * c;
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        skin += ms;
        if (m->prof_has_morph[c]) {
            HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3]));
            morph += ms;
        }
    }
    *n_calls = uint32_t(m->prof_calls);
    *skin_ms_total = float(skin);
    *morph_ms_total = float(morph);
    m->prof_calls = 0;
    return MMDX_OK;
}

void mmdx_debug_reload_env(void) { launch_overrides() = read_launch_overrides(); }

mmdx_status mmdx_debug_last_store_policy(mmdx_model_t model, int32_t *write_through) {
    if (!model || !write_through) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *write_through = model->last_write_through ? 1 : 0;
    return MMDX_OK;
}

mmdx_status mmdx_debug_morph_pass_stats(mmdx_model_t model, uint32_t *walks, uint32_t *device_skips, uint32_t *host_skips) {
    if (!model || !walks || !device_skips || !host_skips) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *walks = *device_skips = 0;
    *host_skips = model->host_skips;
    if (model->device < 0 || !model->seen.ptr) return MMDX_OK;
    HIP_TRY(hipSetDevice(model->device));
    HIP_TRY(hipStreamSynchronize(model->stream));
    uint32_t w[2] = {0, 0};
    HIP_TRY(hipMemcpy(w, static_cast<const uint32_t *>(model->seen.ptr) + kSeenWalks, sizeof(w), hipMemcpyDeviceToHost));
    *walks = w[0];
    *device_skips = w[1];
    return MMDX_OK;
}

#ifndef MMDX_SOURCE_SHA
#define MMDX_SOURCE_SHA "unstamped"
#endif
// (the "mmdx-source-sha:" prefix lets build.py find the stamp in the file without loading it)
const char *mmdx_build_source_sha(void) {
    static const char stamp[] = "mmdx-source-sha:" MMDX_SOURCE_SHA;
    return stamp + sizeof("mmdx-source-sha:") - 1;
}

static mmdx_status bench_stream_op(void *dst, const void *src, size_t bytes, int32_t iters, float *avg_ms) {
    if (!dst || !avg_ms || iters <= 0 || bytes < 16) return fail(MMDX_ERR_INVALID_ARGUMENT, "bad argument");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipError_t e = src ? launch_copy(dst, src, bytes, nullptr) : launch_fill(dst, bytes, nullptr);  // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i)
        e = src ? launch_copy(dst, src, bytes, nullptr) : launch_fill(dst, bytes, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return hip_fail(e, "bench stream op");
    *avg_ms = ms / float(iters);
    return MMDX_OK;
}

mmdx_status mmdx_bench_copy(void *dst, const void *src, size_t bytes, int32_t iters, float *avg_ms) {
    if (!src) return fail(MMDX_ERR_INVALID_ARGUMENT, "src is NULL");
    return bench_stream_op(dst, src, bytes, iters, avg_ms);
}

mmdx_status mmdx_bench_fill(void *dst, size_t bytes, int32_t iters, float *avg_ms) {
    return bench_stream_op(dst, nullptr, bytes, iters, avg_ms);
}

mmdx_status mmdx_bench_store_pattern(void *out_a, void *out_b, uint32_t n_vertices, uint32_t n_instances,
                                     int32_t iters, float *avg_ms) {
    if (!out_a || !out_b || !avg_ms || iters <= 0 || !n_vertices || !n_instances || (n_vertices & 3))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "bad argument (n_vertices must be a multiple of 4)");
    hipError_t e = time_store_pattern(out_a, out_b, n_vertices, n_instances, 12, 12, iters, avg_ms);
    if (e != hipSuccess) return hip_fail(e, "bench store pattern");
    return MMDX_OK;
}

}  // extern "C"
