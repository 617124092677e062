// mmdx_poser.hpp -- C++ host mirror of the hot-path slice of the reference's mmd::Poser, over the C
// ABI in include/mmdx.h.  Header-only; link with libmmdx.so.  No HIP headers needed.
//
// Mirrors (reference file:line, L/ = 3rd_party/libmmd/include/mmd/):
//   mmd::Poser::pose_image, SetMorphPose, ResetPosing (rates only), Deform    L/motion/poser.inl:17-43
//   the palette hook PhysicsReactor::GetPoserBoneImage(...).skinning_matrix_   L/motion/physics.inl:32-40
//   the viewer's struct Vertex + UpdateDeformedVertices()                      main.cpp:50-54, :821-863
//   SetBonePose, PrePhysicsPosing / PostPhysicsPosing (bone solve -> palette)  L/motion/poser_impl.inl:362-394, :466-469
//   mmd::MotionPlayer(motion, poser)::SeekFrame                                L/motion/poser_impl.inl:522-548
//   the loaders' entry points (PmxReader / PmdReader / VmdReader ::Read*)      through Poser::FromFile, Motion
// Same names and argument meaning; errors surface as mmdx::Error (the reference's loaders throw
// mmd::exception, its Deform has no error path at all).
#pragma once

#include <cstdint>
#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mmdx.h"

namespace mmdx {

struct Error : std::runtime_error {
    mmdx_status status;
    Error(mmdx_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(mmdx_status s) {
    if (s != MMDX_OK) throw Error(s, mmdx_last_error_string());
}

struct Vector3f { float x, y, z; };                       // = mmd::Vector3f (12 bytes, packed)
struct Vertex { float pos[3]; float normal[3]; float uv[2]; };  // = main.cpp:50-54
static_assert(sizeof(Vector3f) == 12 && sizeof(Vertex) == 32, "layout");

// Flat model description the application fills once from its loader (mmd::Model getters).
struct ModelData {
    std::vector<float> positions, normals, uvs;           // [NV][3], [NV][3], [NV][2]
    std::vector<int32_t> skin_type, bone_ids;             // [NV], [NV][4]
    std::vector<float> bone_weights;                      // [NV][4]
    std::vector<int32_t> bone_parent;                     // [NB]
    std::vector<int32_t> morph_type;                      // [NM]
    std::vector<uint32_t> morph_offset, morph_index;      // [NM+1], [E]
    std::vector<float> morph_value;                       // [E][3]
    uint32_t n_vertices = 0, n_bones = 0, n_morphs = 0;
};

// Page-locked storage for the buffers handed to the deform calls, e.g.
//   std::vector<Vertex, mmdx::PinnedAllocator<Vertex>> vertices;
// (the viewer's `vertices` array of main.cpp:735): outputs in such memory are written by the kernel directly.
template <class T>
struct PinnedAllocator {
    using value_type = T;
    PinnedAllocator() = default;
    template <class U>
    PinnedAllocator(const PinnedAllocator<U> &) {}
    T *allocate(std::size_t n) {
        void *p = nullptr;
        if (mmdx_host_malloc(&p, n * sizeof(T)) != MMDX_OK) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, std::size_t) noexcept { (void)mmdx_host_free(p); }
    template <class U>
    bool operator==(const PinnedAllocator<U> &) const { return true; }
    template <class U>
    bool operator!=(const PinnedAllocator<U> &) const { return false; }
};

class Poser {
public:
    struct PoseImage {                                    // = mmd::Poser::PoseImage
        std::vector<Vector3f> coordinates;
        std::vector<Vector3f> normals;
    } pose_image;

    // `normalize` = run Model::Normalize's retagging, as both of the reference's readers do at load.  `extra_flags`: the engine's
    // opt-ins -- MMDX_CREATE_FAST_MATH (results within the stated tolerance instead of bit-identical), MMDX_CREATE_TILE_ORDER
    // (pose_image in the engine's vertex order: remap the index buffer once through VertexOrder()).
    explicit Poser(const ModelData &m, bool normalize = true, uint32_t extra_flags = 0)
        : nv_(m.n_vertices), nb_(m.n_bones), nm_(m.n_morphs) {
        mmdx_model_desc d;
        std::memset(&d, 0, sizeof(d));
        d.struct_size = sizeof(d);
        d.flags = (normalize ? MMDX_CREATE_NORMALIZE : 0) | extra_flags;
        d.n_vertices = nv_; d.n_bones = nb_; d.n_morphs = nm_;
        d.positions = m.positions.data(); d.normals = m.normals.data();
        d.uvs = m.uvs.empty() ? nullptr : m.uvs.data();
        d.skin_type = m.skin_type.data(); d.bone_ids = m.bone_ids.data();
        d.bone_weights = m.bone_weights.data();
        d.bone_parent = m.bone_parent.empty() ? nullptr : m.bone_parent.data();
        d.morph_type = m.morph_type.data(); d.morph_offset = m.morph_offset.data();
        d.morph_index = m.morph_index.data(); d.morph_value = m.morph_value.data();
        check(mmdx_model_create(&d, &model_));
        pose_image.coordinates.resize(nv_);
        pose_image.normals.resize(nv_);
        morph_rates_.assign(nm_, 0.0f);
        palette_.assign(size_t(nb_) * 16, 0.0f);
        for (uint32_t b = 0; b < nb_; ++b)
            for (int k = 0; k < 4; ++k) palette_[size_t(b) * 16 + k * 5] = 1.0f;
        Deform();  // the reference's constructor ends with ResetPosing(); Deform() (poser_impl.inl:126-127)
    }
    // From a .pmx / .pmd file through the bundled loaders: the GPU model AND the rig (bone solve on the
    // device, so PrePhysicsPosing() works without libmmd).  Names are kept for MotionPlayer.
    static Poser *FromFile(const std::string &path) {
        mmdx_pmx_t pmx = nullptr;
        const bool pmd = path.size() > 4 && (path.substr(path.size() - 4) == ".pmd" || path.substr(path.size() - 4) == ".PMD");
        check(pmd ? mmdx_pmd_load_file(path.c_str(), &pmx) : mmdx_pmx_load_file(path.c_str(), &pmx));
        Poser *p = nullptr;
        try {
            p = new Poser(pmx);
        } catch (...) {
            mmdx_pmx_destroy(pmx);
            throw;
        }
        mmdx_pmx_destroy(pmx);
        return p;
    }
    ~Poser() {
        if (skeleton_) mmdx_skeleton_destroy(skeleton_);
        mmdx_model_destroy(model_);
    }
    Poser(const Poser &) = delete;
    Poser &operator=(const Poser &) = delete;

    void ResetPosing() {
        std::fill(morph_rates_.begin(), morph_rates_.end(), 0.0f);
        for (uint32_t b = 0; b < uint32_t(bone_poses_.size() / 8); ++b) {
            float *q = bone_poses_.data() + size_t(b) * 8;
            q[0] = q[1] = q[2] = q[3] = q[4] = q[5] = q[6] = 0.0f; q[7] = 1.0f;
        }
        if (skeleton_) { PrePhysicsPosing(); PostPhysicsPosing(); }       // as the reference does (:138-139)
    }
    void SetMorphPose(size_t index, float weight) { morph_rates_.at(index) = weight; }
    // = Poser::SetBonePose(index, Motion::BonePose(translation, rotation)); quaternion x, y, z, w
    void SetBonePose(size_t index, const float translation[3], const float rotation[4]) {
        float *q = bone_poses_.data() + index * 8;
        if (index * 8 + 8 > bone_poses_.size()) throw Error(MMDX_ERR_BAD_INDEX, "SetBonePose: bone index out of range");
        std::memcpy(q, translation, 12); q[3] = 0.0f; std::memcpy(q + 4, rotation, 16);
    }
    // Bone solve on the device (bone morphs, append bones, IK incl. nested IK): local poses -> skinning matrices, in the
    // reference's two steps (main.cpp:1801-1810).  Only for posers that own a rig (FromFile).  PrePhysicsPosing() runs
    // the pre-physics bone list; SkinningMatrix(b) of those bones is then what a reactor's kinematic bodies read
    // (PoserMotionState::Reset).  A reactor hands the transforms of the bodies it moved to SetPhysicsTransforms() --
    // what PoserMotionState::Synchronize() writes, with `strict` marking the bodies Fix() applies to
    // (mmd-bullet_impl.inl:34-56) -- and PostPhysicsPosing() applies them and runs the post-physics list.
    void PrePhysicsPosing() {
        if (!skeleton_) throw Error(MMDX_ERR_UNSUPPORTED, "this Poser has no rig: fill SkinningMatrix() yourself");
        check(mmdx_skeleton_solve_pre(skeleton_, model_, 1, bone_poses_.data(), nm_ ? morph_rates_.data() : nullptr,
                                      MMDX_WEIGHTS_SHARED, palette_.data()));
        physics_bones_.clear(); physics_strict_.clear(); physics_skinning_.clear();
        post_pending_ = true;       // the rows of the post-physics bones are unspecified until PostPhysicsPosing()
    }
    void SetPhysicsTransforms(const std::vector<int32_t> &bones, const std::vector<uint8_t> &strict,
                              const float *skinning /*[bones.size()][16]*/) {
        physics_bones_ = bones;
        physics_strict_ = strict;
        physics_strict_.resize(bones.size(), 0);
        physics_skinning_.assign(skinning, skinning + bones.size() * 16);
    }
    void PostPhysicsPosing() {
        if (!skeleton_) throw Error(MMDX_ERR_UNSUPPORTED, "this Poser has no rig: fill SkinningMatrix() yourself");
        mmdx_physics_overrides ov;
        ov.struct_size = sizeof(ov);
        ov.n_bones = uint32_t(physics_bones_.size());
        ov.bone = physics_bones_.data(); ov.strict = physics_strict_.data(); ov.skinning = physics_skinning_.data();
        check(mmdx_skeleton_solve_post(skeleton_, model_, 1, ov.n_bones ? &ov : nullptr, 0, palette_.data()));
        post_pending_ = false;
    }

    // original_to_engine[file vertex] = its position in pose_image of an MMDX_CREATE_TILE_ORDER poser (what the viewer's index
    // buffer is remapped through, main.cpp:781-787); the identity-free inverse comes back in engine_to_original if asked for.
    std::vector<uint32_t> VertexOrder(std::vector<uint32_t> *engine_to_original = nullptr) const {
        std::vector<uint32_t> o2e(nv_);
        if (engine_to_original) engine_to_original->resize(nv_);
        check(mmdx_model_get_vertex_order(model_, engine_to_original ? engine_to_original->data() : nullptr, o2e.data()));
        return o2e;
    }

    const std::vector<std::string> &bone_names() const { return bone_names_; }
    const std::vector<std::string> &morph_names() const { return morph_names_; }
    std::vector<float> &morph_rates() { return morph_rates_; }
    std::vector<float> &bone_poses() { return bone_poses_; }      // [NB][8]: t.xyz, 0, q.xyzw

    // What a PhysicsReactor-derived tap reads out of the reference Poser after PostPhysicsPosing():
    // float[16], row-vector convention, translation in elements 12..14.
    // (Between PrePhysicsPosing() and PostPhysicsPosing() only the pre-physics bones' matrices are valid -- what a reactor's
    // kinematic bodies read; the frame's palette is final after PostPhysicsPosing(), as in the reference, main.cpp:1810.)
    float *SkinningMatrix(size_t bone) { return palette_.data() + bone * 16; }
    void SetSkinningMatrices(const float *palette /*[NB][16]*/) {
        std::memcpy(palette_.data(), palette, palette_.size() * sizeof(float));
        post_pending_ = false;      // the caller supplied the whole palette
    }

    void Deform() {
        require_final_palette();
        check(mmdx_deform(model_, morph_rates_.data(), palette_.data(),
                          reinterpret_cast<float *>(pose_image.coordinates.data()),
                          reinterpret_cast<float *>(pose_image.normals.data())));
    }

    // Deform + repack in ONE pass on the GPU: replaces main.cpp:1821 and :1824 together.  With a vector on
    // PinnedAllocator (below) the kernel stores the vertices straight into it; a plain vector goes through the
    // library's bounce buffer (one extra CPU copy).
    template <class Alloc>
    void UpdateDeformedVertices(std::vector<Vertex, Alloc> &vertices, float mmd_to_meter = 0.1f) {
        vertices.resize(nv_);
        require_final_palette();
        check(mmdx_deform_vertex32(model_, morph_rates_.data(), palette_.data(), mmd_to_meter,
                                   vertices.data()));
    }

    mmdx_model_t handle() const { return model_; }
    uint32_t vertex_count() const { return nv_; }
    uint32_t bone_count() const { return nb_; }
    uint32_t morph_count() const { return nm_; }

private:
    // Until round 2 PrePhysicsPosing() solved both bone lists; since the physics seam it solves the pre-physics list only.  A
    // host that still calls PrePhysicsPosing() and then deforms would skin with unspecified rows for the post-physics bones:
    // refuse loudly instead.
    void require_final_palette() const {
        if (post_pending_)
            throw Error(MMDX_ERR_INVALID_ARGUMENT, "Deform() after PrePhysicsPosing() without PostPhysicsPosing(): the post-physics "
                                                   "bones' skinning matrices are not solved yet (main.cpp:1801-1810 calls both)");
    }
    bool post_pending_ = false;

    explicit Poser(mmdx_pmx_t pmx) {
        mmdx_model_desc d;
        check(mmdx_pmx_get_model_desc(pmx, &d));
        nv_ = d.n_vertices; nb_ = d.n_bones; nm_ = d.n_morphs;
        check(mmdx_model_create(&d, &model_));
        mmdx_skeleton_desc sd;
        check(mmdx_pmx_get_skeleton_desc(pmx, &sd));
        sd.create_flags |= MMDX_SKELETON_PHYSICS_SEAM;      // PrePhysicsPosing | a reactor's writes | PostPhysicsPosing
        const mmdx_status st = mmdx_skeleton_create(&sd, &skeleton_);
        if (st != MMDX_OK) { mmdx_model_destroy(model_); throw Error(st, mmdx_last_error_string()); }
        char buf[1024];
        for (uint32_t b = 0; b < nb_; ++b) { check(mmdx_pmx_get_name(pmx, MMDX_PMX_NAME_BONE, b, buf, sizeof(buf))); bone_names_.push_back(buf); }
        for (uint32_t m = 0; m < nm_; ++m) { check(mmdx_pmx_get_name(pmx, MMDX_PMX_NAME_MORPH, m, buf, sizeof(buf))); morph_names_.push_back(buf); }
        pose_image.coordinates.resize(nv_);
        pose_image.normals.resize(nv_);
        morph_rates_.assign(nm_, 0.0f);
        palette_.assign(size_t(nb_) * 16, 0.0f);
        bone_poses_.assign(size_t(nb_) * 8, 0.0f);
        ResetPosing();
        Deform();
    }

    uint32_t nv_ = 0, nb_ = 0, nm_ = 0;
    mmdx_model_t model_ = nullptr;
    mmdx_skeleton_t skeleton_ = nullptr;
    std::vector<float> morph_rates_, palette_, bone_poses_;
    std::vector<int32_t> physics_bones_;                  // what a reactor moved this frame (SetPhysicsTransforms)
    std::vector<uint8_t> physics_strict_;
    std::vector<float> physics_skinning_;
    std::vector<std::string> bone_names_, morph_names_;
};

// = mmd::Motion filled by VmdReader::ReadMotion
class Motion {
public:
    explicit Motion(const std::string &vmd_path) { check(mmdx_vmd_load_file(vmd_path.c_str(), &vmd_)); }
    ~Motion() { mmdx_vmd_destroy(vmd_); }
    Motion(const Motion &) = delete;
    Motion &operator=(const Motion &) = delete;
    mmdx_vmd_t handle() const { return vmd_; }
    uint32_t GetLength() const {
        mmdx_vmd_info info;
        info.struct_size = sizeof(info);
        check(mmdx_vmd_get_info(vmd_, &info));
        return info.max_frame;
    }

private:
    mmdx_vmd_t vmd_ = nullptr;
};

// = mmd::MotionPlayer: associates the motion's tracks with the poser's bones and morphs by name at
// construction; SeekFrame evaluates every track at `frame` (on the device) and hands the results to the
// poser through SetMorphPose / SetBonePose, like the reference's loop over its name maps.
class MotionPlayer {
public:
    MotionPlayer(const Motion &motion, Poser &poser) : poser_(poser) {
        std::vector<const char *> bn, mn;
        for (const std::string &s : poser.bone_names()) bn.push_back(s.c_str());
        for (const std::string &s : poser.morph_names()) mn.push_back(s.c_str());
        check(mmdx_vmd_bind_bones(motion.handle(), uint32_t(bn.size()), bn.data(), &bones_));
        const mmdx_status st = mmdx_vmd_bind_morphs(motion.handle(), uint32_t(mn.size()), mn.data(), &morphs_);
        if (st != MMDX_OK) { mmdx_bone_motion_destroy(bones_); throw Error(st, mmdx_last_error_string()); }
        uint32_t nb = 0, mapped = 0, keys = 0, curves = 0;
        check(mmdx_bone_motion_get_info(bones_, &nb, &mapped, &keys, &curves));
        mapped_bones_ = mapped;
    }
    ~MotionPlayer() {
        mmdx_bone_motion_destroy(bones_);
        mmdx_morph_motion_destroy(morphs_);
    }
    MotionPlayer(const MotionPlayer &) = delete;
    MotionPlayer &operator=(const MotionPlayer &) = delete;

    void SeekFrame(size_t frame) {
        const uint32_t f = uint32_t(frame);
        if (poser_.morph_count())
            check(mmdx_morph_motion_eval(morphs_, poser_.handle(), 1, &f, 0, poser_.morph_rates().data()));
        if (poser_.bone_count())
            check(mmdx_bone_motion_eval(bones_, poser_.handle(), 1, &f, 0, poser_.bone_poses().data()));
    }
    uint32_t mapped_bones() const { return mapped_bones_; }

private:
    Poser &poser_;
    mmdx_bone_motion_t bones_ = nullptr;
    mmdx_morph_motion_t morphs_ = nullptr;
    uint32_t mapped_bones_ = 0;
};

}  // namespace mmdx
