// rig_api.cpp -- C ABI of the bone-track and skeleton entry points (include/mmdx.h) over the HIP runtime.
// Static tables (keys, curve tables, chains) are uploaded on first use per device; per-call operands may
// live on the host (copied through handle-owned scratch) or in HBM.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/mmdx.h"
#include "error.hpp"
#include "graph_pin.hpp"
#include "rig.hpp"
#include "rig_kernels.hpp"
#include "vmd.hpp"

using namespace mmdx;

namespace {

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return hip_status(e_, #expr);      \
    } while (0)

struct Buf {
    void *ptr = nullptr;
    size_t bytes = 0;
    const GraphPin *pin = nullptr;           // the owning handle's: a buffer a recorded graph holds may not move
    hipError_t ensure(size_t need) {
        need = std::max<size_t>(need, 16);
        if (need <= bytes) return hipSuccess;
        if (graph_recording()) return hipErrorStreamCaptureUnsupported;   // run the sequence once un-captured first
        if (graph_pinned(pin)) return hipErrorIllegalState;                 // (hip_status turns both into a clear message)
        release();
        hipError_t e = hipMalloc(&ptr, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    template <typename T>
    hipError_t upload(const std::vector<T> &v) {
        hipError_t e = ensure(v.size() * sizeof(T));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() {
        device_free_or_defer(ptr);
        ptr = nullptr; bytes = 0;
    }
};

}  // namespace

struct mmdx_bone_motion_s {
    BoneMotionHost host;
    int device = -1;
    Buf key_off, key_frame, key_tr, key_rot, key_curve, lut, frames_in, out;
    GraphPin pin;                                         // recorded graphs that hold these buffers' addresses
    mmdx_bone_motion_s() {
        for (Buf *b : {&key_off, &key_frame, &key_tr, &key_rot, &key_curve, &lut, &frames_in, &out}) b->pin = &pin;
    }
};

struct mmdx_skeleton_s {
    SkeletonPlan plan;
    int device = -1;
    Buf local_offset, neg_rest, chain_off, chain, poses_in, out;
    Buf order, bones, iks, links, events, rounds, state;  // ordered solver
    Buf apps, app_chain, rates_in, morph_state;           // bone morphs
    Buf over_bone, over_strict, over_skin;                // physics seam: the reactor's writes of one frame
    uint32_t pre_instances = 0;                           // instances of the last mmdx_skeleton_solve_pre (0: none pending)
    const float *pre_poses = nullptr;                     // the poses that call solved (device address)
    const float *pre_morph = nullptr;
    GraphPin pin;                                         // recorded graphs that hold these buffers' addresses
    mmdx_skeleton_s() {
        for (Buf *b : {&local_offset, &neg_rest, &chain_off, &chain, &poses_in, &out, &order, &bones, &iks, &links, &events, &rounds,
                       &state, &apps, &app_chain, &rates_in, &morph_state, &over_bone, &over_strict, &over_skin})
            b->pin = &pin;
    }
    void release_all() {
        for (Buf *b : {&local_offset, &neg_rest, &chain_off, &chain, &poses_in, &out, &order, &bones, &iks, &links, &events, &rounds,
                       &state,
                       &apps, &app_chain, &rates_in, &morph_state, &over_bone, &over_strict, &over_skin})
            b->release();
    }
};

// static tables of a motion / a skeleton -> the device they are about to run on (once per device)
static mmdx_status motion_to_device(mmdx_bone_motion_t m, int device) {
    if (m->device == device) return MMDX_OK;
    if (graph_pinned(&m->pin)) return hip_status(hipErrorIllegalState, "moving a bone motion to another device");
    const BoneMotionHost &h = m->host;
    for (Buf *b : {&m->key_off, &m->key_frame, &m->key_tr, &m->key_rot, &m->key_curve, &m->lut, &m->frames_in, &m->out})
        b->release();
    HIP_TRY(m->key_off.upload(h.key_off));
    HIP_TRY(m->key_frame.upload(h.key_frame));
    HIP_TRY(m->key_tr.upload(h.key_tr));
    HIP_TRY(m->key_rot.upload(h.key_rot));
    HIP_TRY(m->key_curve.upload(h.key_curve));
    HIP_TRY(m->lut.upload(h.lut));
    m->device = device;
    return MMDX_OK;
}
static BoneTrackParams motion_params(mmdx_bone_motion_t m, uint32_t n_instances) {
    BoneTrackParams p;
    p.key_off = static_cast<const uint32_t *>(m->key_off.ptr);
    p.key_frame = static_cast<const uint32_t *>(m->key_frame.ptr);
    p.key_tr = static_cast<const float *>(m->key_tr.ptr);
    p.key_rot = static_cast<const float *>(m->key_rot.ptr);
    p.key_curve = static_cast<const uint32_t *>(m->key_curve.ptr);
    p.lut = static_cast<const float *>(m->lut.ptr);
    p.frames = nullptr; p.out = nullptr;
    p.nb = m->host.nb; p.ni = n_instances;
    return p;
}
static mmdx_status skeleton_to_device(mmdx_skeleton_t s, int device) {
    if (s->device == device) return MMDX_OK;
    if (graph_pinned(&s->pin)) return hip_status(hipErrorIllegalState, "moving a skeleton to another device");
    const SkeletonPlan &pl = s->plan;
    s->release_all();
    HIP_TRY(s->apps.upload(pl.apps));
    HIP_TRY(s->app_chain.upload(pl.app_chain));
    if (pl.serial) {
        HIP_TRY(s->order.upload(pl.order));
        HIP_TRY(s->bones.upload(pl.bones));
        HIP_TRY(s->iks.upload(pl.iks));
        HIP_TRY(s->links.upload(pl.links));
        HIP_TRY(s->events.upload(pl.events));
        HIP_TRY(s->rounds.upload(pl.rounds));
    } else {
        HIP_TRY(s->local_offset.upload(pl.local_offset));
        HIP_TRY(s->neg_rest.upload(pl.neg_rest));
        HIP_TRY(s->chain_off.upload(pl.chain_off));
        HIP_TRY(s->chain.upload(pl.chain));
    }
    s->device = device;
    return MMDX_OK;
}

static mmdx_status skeleton_solve(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances, const float *poses,
                                  const float *morph_weights, uint32_t flags, float *out_palettes, uint32_t passes,
                                  const mmdx_physics_overrides *ov);

extern "C" {

mmdx_status mmdx_vmd_bind_bones(mmdx_vmd_t vmd, uint32_t n_bones, const char *const *bone_names,
                                mmdx_bone_motion_t *out) {
    if (!vmd || !out || (n_bones && !bone_names)) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    try {
        std::unique_ptr<mmdx_bone_motion_s> m(new mmdx_bone_motion_s);
        const VmdBoneTracks t = vmd_bone_tracks(vmd);
        build_bone_motion(*t.names, *t.off, t.keys, n_bones, bone_names, m->host);
        *out = m.release();
    } catch (const std::bad_alloc &) {
        return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    return MMDX_OK;
}

mmdx_status mmdx_bone_motion_get_info(mmdx_bone_motion_t m, uint32_t *n_bones, uint32_t *n_mapped,
                                      uint32_t *n_keys, uint32_t *n_curves) {
    if (!m) return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_bones) *n_bones = m->host.nb;
    if (n_mapped) *n_mapped = m->host.n_mapped;
    if (n_keys) *n_keys = uint32_t(m->host.key_frame.size());
    if (n_curves) *n_curves = uint32_t(m->host.lut.size() / kCurveSamples);
    return MMDX_OK;
}

mmdx_status mmdx_bone_motion_eval(mmdx_bone_motion_t m, mmdx_model_t model, uint32_t n_instances,
                                  const uint32_t *frames, uint32_t flags, float *out_poses) {
    if (!m || !frames || !out_poses || !n_instances)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or n_instances == 0");
    int device;
    hipStream_t st;
    if (mmdx_status s = resolve_stream(model, &device, &st)) return s;
    const BoneMotionHost &h = m->host;
    if (graph_recording() && ((flags & (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE)) != (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE) ||
                              m->device != device))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "while a graph is being recorded every operand must be in device memory and the "
                                               "motion must have run on this device before");
    if (mmdx_status r = motion_to_device(m, device)) return r;
    graph_note_handle(model, &m->pin);
    BoneTrackParams p = motion_params(m, n_instances);
    if (flags & MMDX_FRAMES_ON_DEVICE) {
        p.frames = frames;
    } else {
        HIP_TRY(m->frames_in.ensure(size_t(n_instances) * 4));
        HIP_TRY(hipMemcpyAsync(m->frames_in.ptr, frames, size_t(n_instances) * 4, hipMemcpyHostToDevice, st));
        p.frames = static_cast<const uint32_t *>(m->frames_in.ptr);
    }
    const size_t out_bytes = size_t(n_instances) * h.nb * MMDX_POSE_FLOATS * sizeof(float);
    if (flags & MMDX_OUT_ON_DEVICE) {
        p.out = out_poses;
    } else {
        HIP_TRY(m->out.ensure(out_bytes));
        p.out = static_cast<float *>(m->out.ptr);
    }
    HIP_TRY(launch_bone_track_eval(p, st));
    if (!(flags & MMDX_OUT_ON_DEVICE)) {
        if (out_bytes) HIP_TRY(hipMemcpyAsync(out_poses, p.out, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(wait_stream(st));
    } else if (!(flags & MMDX_FRAMES_ON_DEVICE)) {
        HIP_TRY(wait_stream(st));   // borrowed host frames must be consumed before returning
    }
    return MMDX_OK;
}

void mmdx_bone_motion_destroy(mmdx_bone_motion_t m) {
    if (!m) return;
    graph_drop_handle(&m->pin);
    if (m->device >= 0) (void)hipSetDevice(m->device);
    for (Buf *b : {&m->key_off, &m->key_frame, &m->key_tr, &m->key_rot, &m->key_curve, &m->lut, &m->frames_in,
                   &m->out})
        b->release();
    delete m;
}

mmdx_status mmdx_skeleton_create(const mmdx_skeleton_desc *desc, mmdx_skeleton_t *out) {
    if (!desc || !out || desc->struct_size != sizeof(mmdx_skeleton_desc))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or mmdx_skeleton_desc.struct_size mismatch");
    *out = nullptr;
    if (desc->create_flags & ~uint32_t(MMDX_SKELETON_PHYSICS_SEAM))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "unknown bits in mmdx_skeleton_desc.create_flags");
    try {
        std::unique_ptr<mmdx_skeleton_s> s(new mmdx_skeleton_s);
        std::string err;
        if (mmdx_status st = build_skeleton(*desc, s->plan, err)) return fail(st, "skeleton: " + err);
        *out = s.release();
    } catch (const std::bad_alloc &) {
        return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    return MMDX_OK;
}

mmdx_status mmdx_skeleton_get_info(mmdx_skeleton_t s, mmdx_skeleton_info *info) {
    if (!s || !info || info->struct_size != sizeof(mmdx_skeleton_info))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or mmdx_skeleton_info.struct_size mismatch");
    info->n_bones = s->plan.nb;
    info->n_pre_physics = s->plan.n_pre;
    info->n_post_physics = s->plan.n_post;
    info->max_chain = s->plan.max_chain;
    info->solver = s->plan.serial ? MMDX_SOLVER_SERIAL : MMDX_SOLVER_PARALLEL_FK;
    info->n_ik_bones = s->plan.n_ik;
    info->n_ik_links = s->plan.n_links;
    info->n_append_bones = s->plan.n_append;
    info->n_bone_morph_entries = uint32_t(s->plan.apps.size());
    info->n_solve_rounds = uint32_t(s->plan.rounds.size());
    info->n_ik_rounds_16_lanes = 0;
    for (uint8_t c : s->plan.round_coop) info->n_ik_rounds_16_lanes += (c && !s->plan.nested_ik) ? 1u : 0u;
    return MMDX_OK;
}

mmdx_status mmdx_skeleton_solve_motion(mmdx_skeleton_t s, mmdx_bone_motion_t m, mmdx_model_t model, uint32_t n_instances,
                                       const uint32_t *frames, uint32_t flags, float *out_palettes) {
    if (!s || !m || !frames || !out_palettes || !n_instances)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or n_instances == 0");
    if (m->host.nb != s->plan.nb)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "the motion was bound to " + std::to_string(m->host.nb) + " bones, the skeleton has " +
                                               std::to_string(s->plan.nb));
    int device;
    hipStream_t st;
    if (mmdx_status r = resolve_stream(model, &device, &st)) return r;
    const SkeletonPlan &pl = s->plan;
    const size_t pose_bytes = size_t(n_instances) * pl.nb * MMDX_POSE_FLOATS * sizeof(float);
    if (pl.serial || size_t(pl.nb) * 32 > kMotionFkMaxLds) {
        // append bones / IK (the ordered solver) or a skeleton too large for the LDS pose table: the two launches, the poses in the
        // motion's scratch buffer
        if (graph_recording() && m->out.bytes < pose_bytes)
            return fail(MMDX_ERR_INVALID_ARGUMENT, "run the call once before recording: it sizes the motion's pose buffer");
        if (mmdx_status r = motion_to_device(m, device)) return r;
        HIP_TRY(m->out.ensure(pose_bytes));
        if (mmdx_status r = mmdx_bone_motion_eval(m, model, n_instances, frames, (flags & MMDX_FRAMES_ON_DEVICE) | MMDX_OUT_ON_DEVICE,
                                                  static_cast<float *>(m->out.ptr)))
            return r;
        return mmdx_skeleton_solve(s, model, n_instances, static_cast<const float *>(m->out.ptr),
                                   MMDX_POSES_ON_DEVICE | (flags & MMDX_OUT_ON_DEVICE), out_palettes);
    }
    if (graph_recording() && ((flags & (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE)) != (MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE) ||
                              m->device != device || s->device != device))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "while a graph is being recorded every operand must be in device memory and the motion "
                                               "and the skeleton must have run on this device before");
    if (mmdx_status r = motion_to_device(m, device)) return r;
    if (mmdx_status r = skeleton_to_device(s, device)) return r;
    graph_note_handle(model, &m->pin);
    graph_note_handle(model, &s->pin);
    BoneTrackParams tp = motion_params(m, n_instances);
    if (flags & MMDX_FRAMES_ON_DEVICE) {
        tp.frames = frames;
    } else {
        HIP_TRY(m->frames_in.ensure(size_t(n_instances) * 4));
        HIP_TRY(hipMemcpyAsync(m->frames_in.ptr, frames, size_t(n_instances) * 4, hipMemcpyHostToDevice, st));
        tp.frames = static_cast<const uint32_t *>(m->frames_in.ptr);
    }
    const size_t out_bytes = size_t(n_instances) * pl.nb * 16 * sizeof(float);
    SkeletonParams fp;
    fp.morph = nullptr;
    fp.poses = nullptr;
    if (flags & MMDX_OUT_ON_DEVICE) {
        fp.out = out_palettes;
    } else {
        HIP_TRY(s->out.ensure(out_bytes));
        fp.out = static_cast<float *>(s->out.ptr);
    }
    fp.local_offset = static_cast<const float *>(s->local_offset.ptr);
    fp.neg_rest = static_cast<const float *>(s->neg_rest.ptr);
    fp.chain_off = static_cast<const uint32_t *>(s->chain_off.ptr);
    fp.chain = static_cast<const uint32_t *>(s->chain.ptr);
    fp.nb = pl.nb; fp.ni = n_instances;
    HIP_TRY(launch_motion_fk(tp, fp, st));
    if (!(flags & MMDX_OUT_ON_DEVICE)) {
        HIP_TRY(hipMemcpyAsync(out_palettes, fp.out, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(wait_stream(st));
    } else if (!(flags & MMDX_FRAMES_ON_DEVICE)) {
        HIP_TRY(wait_stream(st));                    // the borrowed host frame numbers must be consumed before returning
    }
    return MMDX_OK;
}

mmdx_status mmdx_skeleton_solve(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances, const float *poses,
                                uint32_t flags, float *out_palettes) {
    return mmdx_skeleton_solve_morphed(s, model, n_instances, poses, nullptr, flags, out_palettes);
}

mmdx_status mmdx_skeleton_solve_morphed(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances,
                                        const float *poses, const float *morph_weights, uint32_t flags,
                                        float *out_palettes) {
    return skeleton_solve(s, model, n_instances, poses, morph_weights, flags, out_palettes, 3u, nullptr);
}

mmdx_status mmdx_skeleton_solve_pre(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances, const float *poses,
                                    const float *morph_weights, uint32_t flags, float *out_palettes) {
    if (s && !s->plan.serial)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_skeleton_solve_pre needs a skeleton created with MMDX_SKELETON_PHYSICS_SEAM");
    return skeleton_solve(s, model, n_instances, poses, morph_weights, flags, out_palettes, 1u, nullptr);
}

mmdx_status mmdx_skeleton_solve_post(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances,
                                     const mmdx_physics_overrides *ov, uint32_t flags, float *out_palettes) {
    if (s && !s->plan.serial)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_skeleton_solve_post needs a skeleton created with MMDX_SKELETON_PHYSICS_SEAM");
    if (s && (!s->pre_instances || s->pre_instances != n_instances))
        return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_skeleton_solve_post without a matching mmdx_skeleton_solve_pre");
    if (ov) {
        if (ov->struct_size != sizeof(mmdx_physics_overrides))
            return fail(MMDX_ERR_INVALID_ARGUMENT, "mmdx_physics_overrides.struct_size mismatch");
        if (ov->n_bones && (!ov->bone || !ov->skinning)) return fail(MMDX_ERR_INVALID_ARGUMENT, "overrides: bone / skinning is NULL");
        for (uint32_t k = 0; s && k < ov->n_bones; ++k)
            if (ov->bone[k] < 0 || uint32_t(ov->bone[k]) >= s->plan.nb)
                return fail(MMDX_ERR_BAD_INDEX, "overrides: bone " + std::to_string(ov->bone[k]) + " out of range");
    }
    return skeleton_solve(s, model, n_instances, nullptr, nullptr, flags, out_palettes, 2u, ov);
}

}  // extern "C"

// passes: bit 0 = reset + bone morphs + pre-physics list, bit 1 = (physics overrides +) post-physics list
static mmdx_status skeleton_solve(mmdx_skeleton_t s, mmdx_model_t model, uint32_t n_instances, const float *poses,
                                  const float *morph_weights, uint32_t flags, float *out_palettes, uint32_t passes,
                                  const mmdx_physics_overrides *ov) {
    if (!s || (!poses && (passes & 1u)) || !out_palettes || !n_instances)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or n_instances == 0");
    int device;
    hipStream_t st;
    if (mmdx_status r = resolve_stream(model, &device, &st)) return r;
    const SkeletonPlan &pl = s->plan;
    if (graph_recording()) {
        const uint32_t need = MMDX_OUT_ON_DEVICE | ((passes & 1u) ? uint32_t(MMDX_POSES_ON_DEVICE) : 0u) |
                              ((passes & 1u) && morph_weights && !pl.apps.empty() ? uint32_t(MMDX_WEIGHTS_ON_DEVICE) : 0u);
        if ((flags & need) != need || s->device != device || (ov && ov->n_bones))
            return fail(MMDX_ERR_INVALID_ARGUMENT, "while a graph is being recorded every operand must be in device memory, the "
                                                   "skeleton must have run on this device before, and physics overrides "
                                                   "(host lists) are not recordable");
    }
    if (mmdx_status r = skeleton_to_device(s, device)) return r;
    graph_note_handle(model, &s->pin);
    struct { const float *poses; float *out; } p;
    const size_t in_bytes = size_t(n_instances) * pl.nb * MMDX_POSE_FLOATS * sizeof(float);
    const size_t out_bytes = size_t(n_instances) * pl.nb * 16 * sizeof(float);
    if (!(passes & 1u)) {
        p.poses = s->pre_poses;                      // the post-physics list re-reads the poses the pre step solved:
                                                     // they must still be there (device poses are the caller's to keep)
    } else if (flags & MMDX_POSES_ON_DEVICE) {
        p.poses = poses;
    } else {
        HIP_TRY(s->poses_in.ensure(in_bytes));
        if (in_bytes) HIP_TRY(hipMemcpyAsync(s->poses_in.ptr, poses, in_bytes, hipMemcpyHostToDevice, st));
        p.poses = static_cast<const float *>(s->poses_in.ptr);
    }
    if (flags & MMDX_OUT_ON_DEVICE) {
        p.out = out_palettes;
    } else {
        HIP_TRY(s->out.ensure(out_bytes));
        p.out = static_cast<float *>(s->out.ptr);
    }
    // bone morphs first: per-bone morph_translation_ / morph_rotation_ of every instance
    const float *morph_state = (passes & 1u) ? nullptr : s->pre_morph;
    bool borrowed_rates = false;
    if ((passes & 1u) && morph_weights && !pl.apps.empty()) {
        const bool shared = (flags & MMDX_WEIGHTS_SHARED) != 0;
        const size_t rate_bytes = size_t(shared ? 1 : n_instances) * pl.nm * sizeof(float);
        BoneMorphParams mp;
        if (flags & MMDX_WEIGHTS_ON_DEVICE) {
            mp.rates = morph_weights;
        } else {
            HIP_TRY(s->rates_in.ensure(rate_bytes));
            HIP_TRY(hipMemcpyAsync(s->rates_in.ptr, morph_weights, rate_bytes, hipMemcpyHostToDevice, st));
            mp.rates = static_cast<const float *>(s->rates_in.ptr);
            borrowed_rates = true;
        }
        HIP_TRY(s->morph_state.ensure(size_t(n_instances) * pl.nb * kMorphStateFloats * sizeof(float)));
        mp.apps = static_cast<const BoneMorphApp *>(s->apps.ptr);
        mp.chain = static_cast<const float *>(s->app_chain.ptr);
        mp.out = static_cast<float *>(s->morph_state.ptr);
        mp.napps = uint32_t(pl.apps.size()); mp.nb = pl.nb; mp.ni = n_instances; mp.nm = pl.nm; mp.shared = shared ? 1u : 0u;
        HIP_TRY(launch_bone_morph(mp, st));
        morph_state = mp.out;
    }
    if (pl.serial) {
        HIP_TRY(s->state.ensure(size_t(n_instances) * pl.nb * kSerialStateFloats * sizeof(float)));
        SerialParams sp;
        sp.morph = morph_state;
        sp.poses = p.poses; sp.out = p.out;
        sp.state = static_cast<float *>(s->state.ptr);
        sp.order = static_cast<const uint32_t *>(s->order.ptr);
        sp.bones = static_cast<const BoneRec *>(s->bones.ptr);
        sp.iks = static_cast<const IkRec *>(s->iks.ptr);
        sp.links = static_cast<const LinkRec *>(s->links.ptr);
        sp.events = static_cast<const uint32_t *>(s->events.ptr);
        sp.rounds = static_cast<const RoundRec *>(s->rounds.ptr);
        sp.nb = pl.nb; sp.ni = n_instances; sp.n_pre = pl.n_pre;
        sp.n_rounds_pre = pl.n_rounds_pre; sp.n_rounds = uint32_t(pl.rounds.size());
        sp.fast_slots = pl.fast_slots;
        sp.windows = pl.windows;
        sp.passes = passes;
        sp.nested = pl.nested_ik ? 1u : 0u;
        bool borrowed_over = false;
        if ((passes & 2u) && ov && ov->n_bones) {     // the reactor's writes, between the two lists
            PhysicsParams pp;
            std::vector<uint32_t> ob(ov->bone, ov->bone + ov->n_bones);
            std::vector<uint8_t> os(ov->n_bones, 0);
            for (uint32_t k = 0; ov->strict && k < ov->n_bones; ++k) os[k] = ov->strict[k] ? 1 : 0;
            HIP_TRY(s->over_bone.ensure(ob.size() * 4));
            HIP_TRY(s->over_strict.ensure(os.size()));
            HIP_TRY(hipMemcpyAsync(s->over_bone.ptr, ob.data(), ob.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(s->over_strict.ptr, os.data(), os.size(), hipMemcpyHostToDevice, st));
            const size_t xf_bytes = size_t(n_instances) * ov->n_bones * 64;
            if (flags & MMDX_OVERRIDES_ON_DEVICE) {
                pp.skinning = ov->skinning;
            } else {
                HIP_TRY(s->over_skin.ensure(xf_bytes));
                HIP_TRY(hipMemcpyAsync(s->over_skin.ptr, ov->skinning, xf_bytes, hipMemcpyHostToDevice, st));
                pp.skinning = static_cast<const float *>(s->over_skin.ptr);
            }
            HIP_TRY(hipStreamSynchronize(st));        // ob / os are locals: the copies above must have read them
            borrowed_over = false;
            pp.bone = static_cast<const uint32_t *>(s->over_bone.ptr);
            pp.strict = static_cast<const uint8_t *>(s->over_strict.ptr);
            pp.out = p.out; pp.state = sp.state; pp.bones = sp.bones;
            pp.k = ov->n_bones; pp.nb = pl.nb; pp.ni = n_instances;
            HIP_TRY(launch_physics_override(pp, st));
        }
        (void)borrowed_over;
        HIP_TRY(launch_skeleton_ordered(sp, pl.round_coop.empty() ? nullptr : pl.round_coop.data(), st));
        if (passes == 1u) { s->pre_instances = n_instances; s->pre_poses = p.poses; s->pre_morph = morph_state; }
        else s->pre_instances = 0;
    } else {
        SkeletonParams fp;
        fp.morph = morph_state;
        fp.poses = p.poses; fp.out = p.out;
        fp.local_offset = static_cast<const float *>(s->local_offset.ptr);
        fp.neg_rest = static_cast<const float *>(s->neg_rest.ptr);
        fp.chain_off = static_cast<const uint32_t *>(s->chain_off.ptr);
        fp.chain = static_cast<const uint32_t *>(s->chain.ptr);
        fp.nb = pl.nb; fp.ni = n_instances;
        HIP_TRY(launch_skeleton_fk(fp, st));
    }
    if (!(flags & MMDX_OUT_ON_DEVICE)) {
        if (out_bytes) HIP_TRY(hipMemcpyAsync(out_palettes, p.out, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(wait_stream(st));
    } else if (!(flags & MMDX_POSES_ON_DEVICE) || borrowed_rates) {
        HIP_TRY(wait_stream(st));   // borrowed host poses / rates must be consumed before returning
    }
    return MMDX_OK;
}

extern "C" void mmdx_skeleton_destroy(mmdx_skeleton_t s) {
    if (!s) return;
    graph_drop_handle(&s->pin);
    if (s->device >= 0) (void)hipSetDevice(s->device);
    s->release_all();
    delete s;
}
