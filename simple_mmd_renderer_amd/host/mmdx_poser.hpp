// mmdx_poser.hpp -- C++ host mirror of the hot-path slice of the reference's mmd::Poser, over the C
// ABI in include/mmdx.h.  Header-only; link with libmmdx.so.  No HIP headers needed.
//
// Mirrors (reference file:line, L/ = 3rd_party/libmmd/include/mmd/):
//   mmd::Poser::pose_image, SetMorphPose, ResetPosing (rates only), Deform    L/motion/poser.inl:17-43
//   the palette hook PhysicsReactor::GetPoserBoneImage(...).skinning_matrix_   L/motion/physics.inl:32-40
//   the viewer's struct Vertex + UpdateDeformedVertices()                      main.cpp:50-54, :821-863
// Same names and argument meaning; errors surface as mmdx::Error (the reference's loaders throw
// mmd::exception, its Deform has no error path at all).
#pragma once

#include <cstdint>
#include <cstring>
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

class Poser {
public:
    struct PoseImage {                                    // = mmd::Poser::PoseImage
        std::vector<Vector3f> coordinates;
        std::vector<Vector3f> normals;
    } pose_image;

    // `normalize` = run Model::Normalize's retagging, as both of the reference's readers do at load.
    explicit Poser(const ModelData &m, bool normalize = true)
        : nv_(m.n_vertices), nb_(m.n_bones), nm_(m.n_morphs) {
        mmdx_model_desc d;
        std::memset(&d, 0, sizeof(d));
        d.struct_size = sizeof(d);
        d.flags = normalize ? MMDX_CREATE_NORMALIZE : 0;
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
    ~Poser() { mmdx_model_destroy(model_); }
    Poser(const Poser &) = delete;
    Poser &operator=(const Poser &) = delete;

    void ResetPosing() { std::fill(morph_rates_.begin(), morph_rates_.end(), 0.0f); }
    void SetMorphPose(size_t index, float weight) { morph_rates_.at(index) = weight; }

    // What a PhysicsReactor-derived tap reads out of the reference Poser after PostPhysicsPosing():
    // float[16], row-vector convention, translation in elements 12..14.
    float *SkinningMatrix(size_t bone) { return palette_.data() + bone * 16; }
    void SetSkinningMatrices(const float *palette /*[NB][16]*/) {
        std::memcpy(palette_.data(), palette, palette_.size() * sizeof(float));
    }

    void Deform() {
        check(mmdx_deform(model_, morph_rates_.data(), palette_.data(),
                          reinterpret_cast<float *>(pose_image.coordinates.data()),
                          reinterpret_cast<float *>(pose_image.normals.data())));
    }

    // Deform + repack in ONE pass on the GPU: replaces main.cpp:1821 and :1824 together.
    void UpdateDeformedVertices(std::vector<Vertex> &vertices, float mmd_to_meter = 0.1f) {
        vertices.resize(nv_);
        check(mmdx_deform_vertex32(model_, morph_rates_.data(), palette_.data(), mmd_to_meter,
                                   vertices.data()));
    }

    mmdx_model_t handle() const { return model_; }
    uint32_t vertex_count() const { return nv_; }

private:
    uint32_t nv_, nb_, nm_;
    mmdx_model_t model_ = nullptr;
    std::vector<float> morph_rates_, palette_;
};

}  // namespace mmdx
