// rig.hpp -- bone tracks and skeleton: host-side compilation (rig.cpp, no HIP) and the parameter
// blocks of the device kernels (rig_kernels.hip), driven by the C ABI in rig_api.cpp.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mmdx.h"

namespace mmdx {

constexpr uint32_t kLinearCurve = 0xFFFFFFFFu;
constexpr uint32_t kCurveSamples = 32;          // Bezier<float, 32>, L/util/math.inl:446
constexpr uint32_t kIdentityParent = 0xFFFFFFFFu;

// ---- bone tracks bound to a model's bones ------------------------------------------------------
struct BoneMotionHost {
    uint32_t nb = 0, n_mapped = 0;
    std::vector<uint32_t> key_off;              // [nb+1]
    std::vector<uint32_t> key_frame;            // [K] ascending inside a bone
    std::vector<float> key_tr;                  // [K][4]  translation xyz, 0
    std::vector<float> key_rot;                 // [K][4]  quaternion xyzw
    std::vector<uint32_t> key_curve;            // [K][4]  table id for x, y, z, rotation; kLinearCurve = identity
    std::vector<float> lut;                     // [n_curves][32] presampled curves
};

// The reference's presampled interpolation curve for control points (x0,y0,x1,y1) given in 1/127
// units.  Returns false for a linear curve (no table needed).
bool presample_curve(int8_t x0, int8_t y0, int8_t x1, int8_t y1, float out[kCurveSamples]);

void build_bone_motion(const std::vector<std::string> &track_names, const std::vector<uint32_t> &track_off,
                       const mmdx_vmd_bone_key *keys, uint32_t n_bones, const char *const *bone_names,
                       BoneMotionHost &out);

struct BoneTrackParams {
    const uint32_t *key_off, *key_frame;
    const float *key_tr, *key_rot;              // float4 per key
    const uint32_t *key_curve;                  // uint4 per key
    const float *lut;
    const uint32_t *frames;                     // [ni]
    float *out;                                 // [ni][nb][8]: t.xyz, 0, q.xyzw
    uint32_t nb, ni;
};

// ---- skeleton: local poses -> skinning palette --------------------------------------------------
enum : uint32_t { kBoneAppendRot = 1u, kBoneAppendTr = 2u, kBoneIsIkLink = 4u, kBoneHasIk = 8u };
enum : uint32_t { kFixNone = 0, kFixX = 1, kFixY = 2, kFixZ = 3, kFixAll = 4 };
enum : uint32_t { kOrderZXY = 0, kOrderXYZ = 1, kOrderYZX = 2 };

struct BoneRec {                                // ordered solver: everything static about one bone (48 B)
    float local_offset[3];
    int32_t parent;                             // -1 = none
    float neg_rest[3];
    int32_t append_parent;                      // valid iff bits has kBoneAppendRot/Tr
    float append_ratio;
    uint32_t bits;
    uint32_t ik;                                // index into iks when kBoneHasIk
    uint32_t pad;
};
struct IkRec {
    uint32_t target, loop, link0, nlinks;       // loop already clamped to <= 256
    float angle_limit;
    uint32_t fast;                              // plain chain topology: solved on an LDS window
    int32_t outside_parent;                     // fast: parent bone of the root-most link, -1 = none
    uint32_t nested;                            // a link or the target is itself an IK bone: its solve runs inside this one
};
constexpr uint32_t kMaxIkDepth = 3;             // IK solves nested inside one another (the reference recurses without bound)
constexpr uint32_t kMaxFastLinks = 6;           // window = (links + target + outside parent) x 31 floats per lane
struct LinkRec {                                // the per-link constants the Poser ctor derives (48 B)
    uint32_t bone, limited, order, fix;
    float lo[4], hi[4];                         // min / max per component, [3] unused
};

struct BoneMorphApp {                           // one application of a bone-morph entry, reference order (48 B)
    uint32_t bone, top;                         // target bone; top-level morph whose rate starts the chain
    uint32_t chain_off, chain_len;              // group sub-rates multiplied on the way down
    float tr[3];
    float pad;
    float rot[4];
};

// Schedule of the ordered solver: the evaluation sequence cut into rounds of events (bone evaluations, an
// IK bone's including its CCD solve) that touch disjoint state, so the slots of one instance run a round
// concurrently and the result is the serial sequence's.  Chains solved on the LDS window sit in the first
// `windows` slots of their round.
struct RoundRec {
    uint32_t first, count;                      // events[first .. first + count), count <= kSolveSlots
};
constexpr uint32_t kSolveSlots = 16;            // events of one instance in flight
constexpr uint32_t kSolveInstances = 16;        // instances per workgroup (kSolveSlots x kSolveInstances threads)
constexpr size_t kSolveLdsBudget = 72 * 1024;   // LDS windows of one workgroup

struct SkeletonPlan {
    uint32_t nb = 0, n_pre = 0, n_post = 0, max_chain = 0;
    uint32_t nm = 0;                            // morphs of the model (row length of the rates)
    std::vector<BoneMorphApp> apps;             // bone-morph applications in UpdateMorphTransform order
    std::vector<float> app_chain;               // group sub-rates
    uint32_t n_ik = 0, n_links = 0, n_append = 0, fast_slots = 0, n_fast = 0;
    bool serial = false;                        // IK or append bones present: the ordered solver, not parallel FK
    std::vector<uint32_t> order;                // evaluation sequence: pre-physics sorted, then post-physics sorted
    // parallel FK
    std::vector<float> local_offset;            // [nb][4] rest position relative to the parent (or absolute)
    std::vector<float> neg_rest;                // [nb][4] -rest position: translation row of the global offset
    std::vector<uint32_t> chain_off;            // [nb+1]
    std::vector<uint32_t> chain;                // per bone: kIdentityParent? then ancestors root-first, the bone last
    // ordered solver
    std::vector<BoneRec> bones;
    std::vector<IkRec> iks;
    std::vector<LinkRec> links;
    std::vector<uint32_t> events;               // bone ids, round after round
    std::vector<RoundRec> rounds;               // pre-physics rounds, then post-physics rounds
    std::vector<uint8_t> round_coop;            // per round: its number of events if every one is an IK bone whose chain is a window
                                                // chain (IkRec::fast) -- the round can run on the 16-lanes-per-solve kernel
                                                // (rig_kernels.hip ik_coop_kernel) -- else 0
    uint32_t n_rounds_pre = 0, windows = 0;     // windows: LDS chain windows per instance
    bool nested_ik = false;                     // some IK chain holds an IK bone among its links / as its target
};

// Bones whose state evaluating `bone` reads / writes in the ordered solver (UpdateBoneTransform incl. the IK
// solve): what the round schedule is derived from, exposed for the schedule checker in tests/.
void solve_event_sets(const SkeletonPlan &plan, uint32_t bone, std::vector<uint32_t> &reads, std::vector<uint32_t> &writes);

// status: MMDX_OK, or the error code with its text in `err`.
mmdx_status build_skeleton(const mmdx_skeleton_desc &d, SkeletonPlan &out, std::string &err);

constexpr uint32_t kMorphStateFloats = 7;       // morph_translation_ xyz, morph_rotation_ ijke

struct BoneMorphParams {
    const BoneMorphApp *apps;
    const float *chain;
    const float *rates;                         // [ni or 1][nm]
    float *out;                                 // [nb][7][ni], instance fastest
    uint32_t napps, nb, ni, nm, shared;
};

struct SkeletonParams {
    const float *morph;                         // [nb][7][ni] or nullptr (no bone morphs)
    const float *poses;                         // [ni][nb][8]
    float *out;                                 // [ni][nb][16] row-major, row-vector convention
    const float *local_offset, *neg_rest;       // float4 per bone
    const uint32_t *chain_off, *chain;
    uint32_t nb, ni;
};

constexpr uint32_t kSerialStateFloats = 4 + 4 + 4 + 3 + 16;   // total_rot, ik_rot, pre_ik_rot, total_tr, local

struct SerialParams {
    const float *morph;                         // [nb][7][ni] or nullptr (no bone morphs)
    const float *poses;                         // [ni][nb][8]
    float *out;                                 // [ni][nb][16]
    float *state;                               // [nb][kSerialStateFloats][ni] scratch, instance fastest
    const uint32_t *order;
    const BoneRec *bones;
    const IkRec *iks;
    const LinkRec *links;
    const uint32_t *events;
    const RoundRec *rounds;
    uint32_t nb, ni, n_pre;
    uint32_t n_rounds_pre, n_rounds;
    uint32_t fast_slots;                        // LDS window size in bones (0: no fast chain)
    uint32_t windows;                           // LDS windows per instance
    uint32_t passes;                            // bit 0: reset + pre-physics list, bit 1: post-physics list
    uint32_t nested;                            // some IK chain holds an IK bone: the kernel variant with nested solves
    // one launch may run a SEGMENT of the schedule (the rounds in between go to ik_coop_kernel): rounds [seg_r0, seg_r1) of the
    // lists in `passes`; seg_flags bit 0: PrePhysicsPosing's reset first, bit 1 / 2: the palette rows of the pre- / post-physics list
    uint32_t seg_r0, seg_r1, seg_flags;
};

// The physics reactor's writes between the two lists (mmdx_skeleton_solve_post): Synchronize, then Fix.
struct PhysicsParams {
    const uint32_t *bone;                       // [k]
    const uint8_t *strict;                      // [k]
    const float *skinning;                      // [ni][k][16]
    float *out;                                 // [ni][nb][16] palettes (rows of the listed bones are rewritten)
    float *state;                               // the ordered solver's scratch
    const BoneRec *bones;
    uint32_t k, nb, ni;
};

}  // namespace mmdx
