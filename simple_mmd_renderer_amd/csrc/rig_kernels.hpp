// rig_kernels.hpp -- launchers of rig_kernels.hip and the stream hand-over from api.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include "rig.hpp"

namespace mmdx {
int env_int(const char *name, int dflt);   // api.cpp

hipError_t launch_bone_track_eval(const BoneTrackParams &p, hipStream_t stream);
hipError_t launch_skeleton_fk(const SkeletonParams &p, hipStream_t stream);
// bone tracks -> palette in one launch (parallel-FK skeletons): poses of an instance live in LDS (nb * 32 bytes <= kMotionFkMaxLds)
constexpr size_t kMotionFkMaxLds = 64 * 1024;
hipError_t launch_motion_fk(const BoneTrackParams &t, const SkeletonParams &p, hipStream_t stream);
hipError_t launch_skeleton_ordered(const SerialParams &p, const uint8_t *round_coop /* host, [n_rounds] or nullptr */, hipStream_t stream);
hipError_t launch_bone_morph(const BoneMorphParams &p, hipStream_t stream);
hipError_t launch_physics_override(const PhysicsParams &p, hipStream_t stream);

// api.cpp: the device and stream a motion / rig call runs on -- the model's own when a (device) model
// is given, so that the deform call that follows is ordered after it; else the selected device's
// default stream.  Fails with MMDX_ERR_NO_DEVICE when there is no GPU (no CPU fallback).
mmdx_status resolve_stream(mmdx_model_t model, int *device, hipStream_t *stream);
mmdx_status hip_status(hipError_t e, const char *what);
// api.cpp: the calling thread is between mmdx_graph_begin and mmdx_graph_end (nothing may allocate, copy from the host or wait)
bool graph_recording();
// api.cpp: hipFree, or -- while this thread records a graph, when the runtime refuses frees -- parked until mmdx_graph_end
void device_free_or_defer(void *ptr);
// api.cpp: host-visible completion of the work queued on `stream` (see there).
hipError_t wait_stream(hipStream_t stream);

}  // namespace mmdx
