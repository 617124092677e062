// vmd.hpp -- glue between the VMD host code (vmd.cpp, no HIP) and the HIP side (api.cpp, kernels.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mmdx.h"
#include "graph_pin.hpp"

namespace mmdx {

struct MorphMotionHost {          // keyframes of every model morph, model-morph order
    uint32_t nm;
    const uint32_t *key_off;      // [nm+1]
    const uint32_t *frames;       // [nkeys] ascending inside a morph
    const float *weights;         // [nkeys]
    uint32_t nkeys;
};

struct MorphMotionDevice {        // owned by api.cpp (hipMalloc / hipFree)
    void *key_off = nullptr, *frames = nullptr, *weights = nullptr;
    void *frames_in = nullptr, *out = nullptr;   // per-call scratch for host-pointer callers
    size_t frames_in_bytes = 0, out_bytes = 0;
    int device = -1;
    GraphPin pin;                 // recorded graphs that hold these addresses
};

const MorphMotionHost morph_motion_host(const mmdx_morph_motion_s *m);
MorphMotionDevice &morph_motion_device(mmdx_morph_motion_s *m);
void morph_motion_release_device(MorphMotionDevice &d);   // api.cpp

struct VmdBoneTracks {            // views into a parsed motion, valid until mmdx_vmd_destroy
    const std::vector<std::string> *names;    // UTF-8, track order
    const std::vector<uint32_t> *off;         // [tracks+1] into keys
    const mmdx_vmd_bone_key *keys;            // sorted by frame inside a track
};
VmdBoneTracks vmd_bone_tracks(const mmdx_vmd_s *v);

struct MorphTrackParams {
    const uint32_t *key_off, *key_frames;
    const float *key_weights;
    const uint32_t *frames;       // [ni] frame number of every instance
    float *out;                   // [ni][nm]
    uint32_t nm, ni;
};

}  // namespace mmdx
