// kernels.hpp -- launch interface between the C-ABI host code (api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.hpp"
#include "vmd.hpp"

namespace mmdx {

// Morph handling of one deform launch
enum : int {
    kMorphNone = 0,     // model has no vertex-morph slot
    kMorphShared = 1,   // positions come from the `morphed` buffer written by morph_apply (crowd
                        // with one shared facial state: the morph pass runs once per call)
    kMorphFused1 = 2,   // ONE set of morph rates for the launch, gathered inside the deform kernel: a single-model frame,
                        // or a crowd with a shared facial state (every workgroup repeats its tile's walk)
    kMorphFused4 = 3    // per-instance weights, 4 instances share one pass over a CSR row
};

struct DeformParams {
    // static streams (HBM, uploaded once by mmdx_model_create)
    const TileHdr *tiles;
    const void *spos;            // f32 [NV][3]  |  f16 mode: u16 [NV][4]
    const float *snrm;           // [NV][3]
    const float *suv;            // [NV][2]
    const uint16_t *perm;        // [NV]
    const uint16_t *skin1;
    const uint32_t *skin2_ids;
    const float *skin2_w;
    const uint2 *skin4_ids;
    const float4 *skin4_w;
    const uint32_t *bone_list;
    const uint2 *ell;            // [ntiles*8] {first entry, padded row length} per 64-slot slice
    const void *entries;         // float4 [NE]  |  f16 mode: uint2 [NE]
    // per call
    const float *palettes;       // [NI][NB][16] device
    const float *wslot;          // rows of NS+1 weights (column NS = table padding, always 0):
                                 // kMorphFused1: f32 [NI][NS+1]
                                 // kMorphFused4: float4 [ceil(NI/4)][NS+1] (instance quads)
                                 // morph_apply : f32 [NS+1]
    float *morphed;              // f32 [NV][3] sorted order (kMorphShared)
    // kMorphFused1 with fused_rates != nullptr: slot weights are evaluated inside the kernel from the
    // raw morph rates [NI][NM] (saves the flatten launch of a single-model frame)
    const float *fused_rates;
    const uint32_t *slot_top, *chain_off;
    const float *chain_rate;
    uint32_t nm;
    void *out_a;
    void *out_b;
    uint32_t nv, nb, ns, ni;
    uint32_t group;              // instances per workgroup (multiple of 4 for kMorphFused4)
    uint32_t ntiles, ngroups, rem_per_xcd;  // filled by launch_deform (XCD-aware work mapping)
    uint32_t pal_stride;         // float4 per instance in LDS (= max_tile_bones * 3)
    uint32_t stage_off;          // byte offsets inside dynamic LDS
    uint32_t w_off;
    uint32_t mp_off;             // pack_kernel: the coordinate / position-image regions
    float pos_scale;
    uint32_t out_aligned;        // out_a and out_b are 16-byte aligned
    uint32_t finite_offsets;     // every vertex-morph offset is finite (branch-free morph skip is exact)
    uint32_t interleave;         // crowd modes: instance = g*ngroups + grp instead of grp*group + g
    uint32_t tile_order;         // MMDX_CREATE_TILE_ORDER: outputs in the engine's vertex order (tile-local class sort), stored straight
                                 // from registers -- no LDS image, no per-instance barrier
    uint32_t write_through;      // launch the write-through flavour of the copy-out (api.cpp sets it only where deform_has_write_through())
    uint32_t stagger;            // per-instance-morph kernels: first-round workgroups of residency slot k start k * stagger * 64 cycles late,
                                 // so that the workgroups sharing a CU are not all in their walk (or all in their stores) at once
    uint32_t slots_per_cu;       // ... with this many workgroups resident per CU
    unsigned long long *stamps;  // diagnostic builds of pack_kernel only (PK_STAMPS): per-wave cycle counters; nullptr in the product
    uint32_t *morph_seen;        // kMorphFused1 crowds: the handle's RatesSeen record; a launch that overwrites `morphed` clears its
                                 // valid word (the record no longer describes what `morphed` holds)
};

// Device-side record of the morph rates the `morphed` buffer of a handle was last computed from (shared morph pass of a crowd):
// morph_apply_kernel compares the call's rates with it, bit for bit, and skips its walk when they are equal -- the automatic form of
// MMDX_MORPH_UNCHANGED for rates that live in device memory, where the host cannot look.  u32 words:
enum : uint32_t { kSeenValid = 0, kSeenTicket = 1, kSeenWalks = 2, kSeenSkips = 3, kSeenRates = 4 };   // then NM rate bit patterns

struct FlattenParams {
    const float *rates;          // [NIw][NM] device
    const uint32_t *slot_top, *chain_off;
    const float *chain_rate;
    float *out;
    uint32_t nm, ns, niw;
    uint32_t quad;               // 1: write float4 [ceil(NIw/4)][NS] instance quads (pad lanes = 0)
    uint32_t *seen;              // morph_apply with the flatten fused in: the handle's RatesSeen record (nullptr: always walk)
};

// Bytes of dynamic LDS the deform kernel needs for (layout, morph mode, group).
size_t deform_lds_bytes(int threads, int layout, int morph, uint32_t group, uint32_t max_tile_bones, uint32_t ns,
                        uint32_t *stage_off, uint32_t *w_off, bool tile_order = false);

// Does this launch shape exist in the write-through store flavour?  (Measured to pay on the SoA f32 crowd kernels only: 256 threads,
// no morphs or shared morphs, original vertex order.)  api.cpp asks before it sets DeformParams::write_through, pick() asserts it.
constexpr bool deform_has_write_through(int threads, int layout, int morph, bool f16, bool tile_order) {
    return threads == 256 && layout == 0 /* MMDX_OUT_SOA */ && !f16 && !tile_order && (morph == kMorphNone || morph == kMorphShared);
}

hipError_t launch_deform(int threads, int layout, int morph, bool f16, const DeformParams &p,
                         uint32_t ntiles, size_t lds_bytes, hipStream_t stream);
// Per-instance morph weights, packs of 4 instances, 512 threads (kernels.hip pack_kernel): SoA f32 and f16-position layouts, original
// vertex order.  p.group a multiple of 4; LDS offsets from pack_lds_bytes.
size_t pack_lds_bytes(uint32_t group, uint32_t max_tile_bones, uint32_t ns, uint32_t *stage_off, uint32_t *w_off, uint32_t *mp_off);
hipError_t launch_pack(int layout, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes, hipStream_t stream);
hipError_t launch_pack_fast(int layout, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes, hipStream_t stream);
// One frame of one model (ni == 1, kMorphNone / kMorphFused1): latency-ordered kernel, a workgroup = `threads` (128 / 256) sorted
// slots of a tile, direct stores.  LDS: the tile's palette, then the slot weights at *w_off.
size_t frame_lds_bytes(int morph, uint32_t max_tile_bones, uint32_t ns, uint32_t *w_off);
hipError_t launch_frame(int threads, int layout, int morph, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes,
                        hipStream_t stream);
// `fused` != nullptr: evaluate the slot weights inside the kernel (ns <= kMaxFusedSlots), no flatten launch
hipError_t launch_morph_apply(bool f16, const DeformParams &p, const FlattenParams *fused,
                              hipStream_t stream);
constexpr uint32_t kMaxFusedSlots = 8192;
hipError_t launch_pattern_fill(void *a, void *b, uint32_t nv, uint32_t ni, uint32_t bpva, uint32_t bpvb,
                               hipStream_t stream);
hipError_t launch_flatten(const FlattenParams &p, hipStream_t stream);
hipError_t launch_morph_track_eval(const MorphTrackParams &t, hipStream_t stream);
hipError_t launch_copy(void *dst, const void *src, size_t bytes, hipStream_t stream);
hipError_t launch_fill(void *dst, size_t bytes, hipStream_t stream);
hipError_t prepare_kernels();  // raise the dynamic-LDS limit of every deform variant (once)
// kernels_fast.hip: the same kernels with multiply-add contraction allowed (MMDX_CREATE_FAST_MATH models)
hipError_t launch_deform_fast(int threads, int layout, int morph, bool f16, const DeformParams &p,
                              uint32_t ntiles, size_t lds_bytes, hipStream_t stream);
hipError_t launch_frame_fast(int threads, int layout, int morph, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes,
                             hipStream_t stream);
hipError_t launch_morph_apply_fast(bool f16, const DeformParams &p, const FlattenParams *fused, hipStream_t stream);
hipError_t prepare_kernels_fast();

}  // namespace mmdx
