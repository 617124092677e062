/* mmdx_bench.h -- measurement and A/B entry points of libmmdx.so.
 *
 * NOT part of the drop-in boundary (include/mmdx.h): nothing in the reference corresponds to these.  They exist
 * for bench.py, tools/ and the GPU tests: HIP-event timers on a model's stream, per-kernel profiling of
 * mmdx_deform_batched, the streaming copy / fill / store-pattern ceilings printed next to the roofline
 * (SURVEY.md section 8d), and the re-read of the launch-shape override environment.  Same library, same status codes.
 */
#ifndef MMDX_BENCH_H_INCLUDED
#define MMDX_BENCH_H_INCLUDED

#include "mmdx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- timing on the handle's stream (HIP events; for bench harnesses) ------------------------- */
MMDX_API mmdx_status mmdx_timer_start(mmdx_model_t model);
MMDX_API mmdx_status mmdx_timer_stop(mmdx_model_t model, float *elapsed_ms); /* syncs the stream  */
/* Per-kernel timing with no host synchronisation: while enabled, mmdx_deform_batched records HIP events on
 * the launch stream around its morph kernels and around its skinning kernel -- on every call (enabled == 1)
 * or on every N-th call (enabled == N > 1): the four event records cost the stream about 7 us per call, which a
 * throughput measurement should not pay on every step.  (Timing the skinning kernel alone is not offered: its
 * start event must follow another event record, or it is stamped with the end of the previous KERNEL and the
 * interval then includes the launch gap.)  mmdx_profile_collect waits for the recorded calls, returns how
 * many were timed and the summed milliseconds of the skinning kernel and of the morph pass, and resets
 * the recording. */
MMDX_API mmdx_status mmdx_profile_enable(mmdx_model_t model, int32_t enabled);
MMDX_API mmdx_status mmdx_profile_collect(mmdx_model_t model, uint32_t *n_calls, float *skin_ms_total,
                                          float *morph_ms_total);

/* The launch-shape overrides for A/B runs (environment variables MMDX_GROUP, MMDX_THREADS, MMDX_LDS_TARGET,
 * MMDX_INTERLEAVE; tools/ab.py) are read once per process; this re-reads them.  Not for product use. */
MMDX_API void mmdx_debug_reload_env(void);
/* Which store flavour the kernel of the model's last mmdx_deform_batched call ran with: 0 = cached non-temporal stores, 1 =
 * write-through (mmdx.h, MMDX_OUT_STORES_*; 1 only where the launched kernel has that flavour). */
MMDX_API mmdx_status mmdx_debug_last_store_policy(mmdx_model_t model, int32_t *write_through);
/* What became of the model's shared morph passes so far (mmdx.h, MMDX_MORPH_UNCHANGED): launches that walked the morph table,
 * launches whose device-side comparison found the rates unchanged and skipped the walk, and calls whose host-side comparison
 * skipped the launch altogether.  Waits for the model's stream. */
MMDX_API mmdx_status mmdx_debug_morph_pass_stats(mmdx_model_t model, uint32_t *walks, uint32_t *device_skips,
                                                 uint32_t *host_skips);
/* The revision of the sources this library was built from: SHA-1 over the translation units and headers of build.py's list,
 * in that order, embedded at build time (-DMMDX_SOURCE_SHA).  bench.py and smoke() print it next to the same hash of the files
 * in the tree and refuse to run when they differ: a timed library is provably the committed code. */
MMDX_API const char *mmdx_build_source_sha(void);
/* Device-to-device streaming copy / fill timed with HIP events: the practical HBM ceiling printed
 * next to the roofline.  bytes_moved = 2*bytes for copy, bytes for fill. */
MMDX_API mmdx_status mmdx_bench_copy(void *dst_device, const void *src_device, size_t bytes,
                                     int32_t iterations, float *avg_ms);
MMDX_API mmdx_status mmdx_bench_fill(void *dst_device, size_t bytes, int32_t iterations,
                                     float *avg_ms);
/* Store-only replay of the crowd kernel's SoA output pattern (two arrays of n_instances x
 * n_vertices x 12 bytes, 6 KiB pieces): the write ceiling of THAT pattern on this box. */
MMDX_API mmdx_status mmdx_bench_store_pattern(void *out_a_device, void *out_b_device,
                                              uint32_t n_vertices, uint32_t n_instances,
                                              int32_t iterations, float *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* MMDX_BENCH_H_INCLUDED */
