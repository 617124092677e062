// libmmd_glue.hpp -- the REFERENCE-SIDE binding a maintainer of simple_mmd_renderer adds to put mmdx under the viewer
// (INTEGRATION.md section 1): everything that touches libmmd's own types lives here, everything behind it is the C ABI of
// include/mmdx.h.  Header-only C++ (the reference's language); needs libmmd's headers on the include path
// (3rd_party/libmmd/include) and nothing of HIP.  Include it AFTER "mmd/mmd.hxx", as main.cpp would (main.cpp:22).
//
//   mmdx::glue::CreateMmdxModel(model, flags, &handle)   mmd::Model (as PmxReader / PmdReader left it) -> mmdx_model_t.
//       Replaces nothing in the viewer; called once after the Poser is constructed (main.cpp:650-699, :663).  Reads the
//       model through its public accessors only: Model::GetVertex / SkinningOperator (L/model/model.inl:21-165),
//       Model::GetBone (:204-281), Model::GetMorph / MorphData (:334-517).
//   mmdx::glue::PaletteTap::Read(poser, nb, out)         the finished bone palette, through the reference's own plug-in door:
//       PhysicsReactor::GetPoserBoneImage (L/motion/physics.inl:32-40, friend of Poser: L/motion/poser.inl:15) -- the one
//       mmd-bullet uses to WRITE the same field (mmd-bullet_impl.inl:34-56).  Call after PostPhysicsPosing() (main.cpp:1810).
//   mmdx::glue::MorphRateMirror                          the morph half of MotionPlayer::SeekFrame (L/motion/poser_impl.inl:521-542)
//       into a plain float array: Poser::morph_rates_ is private and has no accessor, so the viewer mirrors the one loop that
//       fills it.  Seek(frame, rates) == ResetPosing()'s zeroing (:131-133) + SetMorphPose for every registered morph.
//   mmdx::glue::DeformFrame(...)                         the two replaced lines, main.cpp:1821 + :1824, as one call.
//
// Compiled and run by tests/test_libmmd_glue.py against the real libmmd (build container) and, through
// oracle/_ref/libmmd_ref.so, on the GPU box (glue -> mmdx_deform_vertex32 vs libmmd's golden vertices).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mmdx.h"

namespace mmdx {
namespace glue {

// The flat arrays mmdx_model_desc points into; kept alive by the caller for the duration of mmdx_model_create only.
struct FlatModel {
    std::vector<float> positions, normals, uvs, bone_weights, morph_value;
    std::vector<int32_t> skin_type, bone_ids, bone_parent, morph_type;
    std::vector<uint32_t> morph_offset, morph_index;

    mmdx_model_desc Desc(uint32_t flags) const {
        mmdx_model_desc d;
        std::memset(&d, 0, sizeof(d));
        d.struct_size = sizeof(d);
        d.flags = flags;
        d.n_vertices = uint32_t(skin_type.size());
        d.n_bones = uint32_t(bone_parent.size());
        d.n_morphs = uint32_t(morph_type.size());
        d.positions = positions.data(); d.normals = normals.data(); d.uvs = uvs.data();
        d.skin_type = skin_type.data(); d.bone_ids = bone_ids.data(); d.bone_weights = bone_weights.data();
        d.bone_parent = bone_parent.data();
        d.morph_type = morph_type.data(); d.morph_offset = morph_offset.data();
        d.morph_index = morph_index.data(); d.morph_value = morph_value.data();
        return d;
    }
};

// mmd::Model -> flat arrays.  Indices keep libmmd's meaning: a bone id it zero-extended from "-1" (255 / 65535,
// L/util/dwarf_impl.inl:90-95) arrives as that number and mmdx_model_create treats it as the reference's arithmetic does
// (irrelevant where the weight makes it so, an error otherwise); size_t(-1) parents become -1.
inline void Flatten(const mmd::Model &model, FlatModel &f) {
    const size_t nv = model.GetVertexNum(), nb = model.GetBoneNum(), nm = model.GetMorphNum();
    f.positions.resize(nv * 3); f.normals.resize(nv * 3); f.uvs.resize(nv * 2);
    f.bone_weights.assign(nv * 4, 0.f);
    f.skin_type.resize(nv); f.bone_ids.assign(nv * 4, 0);
    f.bone_parent.resize(nb); f.morph_type.resize(nm);
    f.morph_offset.assign(nm + 1, 0); f.morph_index.clear(); f.morph_value.clear();
    typedef mmd::Model::SkinningOperator Op;
    for (size_t i = 0; i < nv; ++i) {
        // const: Vertex<cref>'s NON-const GetSkinningOperator() does not compile (it returns a mutable reference to a const
        // member, L/model/model_vertex_impl.inl:125-129; a template, so only an instantiation shows it) -- Poser::Deform reads
        // through a const proxy too (L/motion/poser_impl.inl:405-406)
        const mmd::Model::Vertex<mmd::cref> v = model.GetVertex(i);
        std::memcpy(&f.positions[3 * i], v.GetCoordinate().v, 12);
        std::memcpy(&f.normals[3 * i], v.GetNormal().v, 12);
        std::memcpy(&f.uvs[2 * i], v.GetUVCoordinate().v, 8);
        const Op &op = v.GetSkinningOperator();
        f.skin_type[i] = int32_t(op.GetSkinningType());
        int32_t *id = &f.bone_ids[4 * i];
        float *w = &f.bone_weights[4 * i];
        switch (op.GetSkinningType()) {
        case Op::SKINNING_BDEF1:
            id[0] = int32_t(op.GetBDEF1().GetBoneID());
            break;
        case Op::SKINNING_BDEF4:
            for (int k = 0; k < 4; ++k) {
                id[k] = int32_t(op.GetBDEF4().GetBoneID(k));
                w[k] = op.GetBDEF4().GetBoneWeight(k);
            }
            break;
        case Op::SKINNING_SDEF:
            id[0] = int32_t(op.GetSDEF().GetBoneID(0)); id[1] = int32_t(op.GetSDEF().GetBoneID(1));
            w[0] = op.GetSDEF().GetBoneWeight();
            break;
        default:   // BDEF2 and every unknown tag: Poser::Deform's `default:` reads the BDEF2 member (poser_impl.inl:417-426)
            id[0] = int32_t(op.GetBDEF2().GetBoneID(0)); id[1] = int32_t(op.GetBDEF2().GetBoneID(1));
            w[0] = op.GetBDEF2().GetBoneWeight();
            break;
        }
    }
    for (size_t b = 0; b < nb; ++b) {
        const size_t p = model.GetBone(b).GetParentIndex();
        f.bone_parent[b] = p < nb ? int32_t(p) : -1;
    }
    typedef mmd::Model::Morph Morph;
    for (size_t m = 0; m < nm; ++m) {
        const Morph &morph = model.GetMorph(m);
        f.morph_type[m] = int32_t(morph.GetType());
        for (size_t j = 0; j < morph.GetMorphDataNum(); ++j) {
            const Morph::MorphData &d = morph.GetMorphData(j);
            if (morph.GetType() == Morph::MORPH_TYPE_VERTEX) {
                const mmd::Vector3f &o = d.GetVertexMorph().GetOffset();
                f.morph_index.push_back(uint32_t(d.GetVertexMorph().GetVertexIndex()));
                for (int c = 0; c < 3; ++c) f.morph_value.push_back(o.v[c]);      // (libmmd's vectors are packed structs: no pointers into them)
            } else if (morph.GetType() == Morph::MORPH_TYPE_GROUP) {
                f.morph_index.push_back(uint32_t(d.GetGroupMorph().GetMorphIndex()));
                f.morph_value.push_back(d.GetGroupMorph().GetMorphRate());
                f.morph_value.push_back(0.f); f.morph_value.push_back(0.f);
            } else {   // bone / uv / material morphs: not on the deformation path (poser_impl.inl:347-358); entries keep their slot
                f.morph_index.push_back(0);
                f.morph_value.insert(f.morph_value.end(), 3, 0.f);
            }
        }
        f.morph_offset[m + 1] = uint32_t(f.morph_index.size());
    }
}

// flags: 0 for a model that came out of PmxReader / PmdReader (both end with model.Normalize(): the tags are final);
// MMDX_CREATE_NORMALIZE for one built by hand; MMDX_CREATE_HOST_ONLY to validate without a GPU.
inline mmdx_status CreateMmdxModel(const mmd::Model &model, uint32_t flags, mmdx_model_t *out) {
    FlatModel f;
    Flatten(model, f);
    const mmdx_model_desc d = f.Desc(flags);
    return mmdx_model_create(&d, out);
}

// The ten pure virtuals of the reference's only plug-in interface, stubbed; Read() is the point.
struct PaletteTap : mmd::PhysicsReactor {
    void AddPoser(mmd::Poser &) override {}
    void RemovePoser(mmd::Poser &) override {}
    void Reset() override {}
    void React(float) override {}
    void SetGravityStrength(float) override {}
    void SetGravityDirection(const mmd::Vector3f &) override {}
    float GetGravityStrength() const override { return 0.f; }
    mmd::Vector3f GetGravityDirection() const override { return mmd::Vector3f(); }
    void SetFloor(bool) override {}
    bool IsHasFloor() const override { return false; }
    static void Read(mmd::Poser &poser, size_t n_bones, float *out /*[n_bones][16]*/) {
        for (size_t b = 0; b < n_bones; ++b) std::memcpy(out + 16 * b, GetPoserBoneImage(poser, b).skinning_matrix_.v, 64);
    }
};

// Built next to the viewer's MotionPlayer (main.cpp: wherever `motion_player` is created) from the same motion and model.
class MorphRateMirror {
public:
    MorphRateMirror(const mmd::Motion &motion, const mmd::Model &model) : motion_(motion), n_morphs_(model.GetMorphNum()) {
        for (size_t i = 0; i < n_morphs_; ++i) {
            const std::wstring &name = model.GetMorph(i).GetName();
            if (motion_.IsMorphRegistered(name)) map_.push_back(std::make_pair(name, i));
        }
    }
    size_t size() const { return n_morphs_; }
    // rates[n_morphs]: what Poser::morph_rates_ holds after ResetPosing(); MotionPlayer::SeekFrame(frame)
    void Seek(size_t frame, float *rates) const {
        for (size_t i = 0; i < n_morphs_; ++i) rates[i] = 0.f;
        for (size_t k = 0; k < map_.size(); ++k) rates[map_[k].second] = motion_.GetMorphPose(map_[k].first, frame).GetWeight();
    }

private:
    const mmd::Motion &motion_;
    size_t n_morphs_;
    std::vector<std::pair<std::wstring, size_t> > map_;
};

// The viewer's two replaced lines -- poser->Deform(); UpdateDeformedVertices(); (main.cpp:1821, :1824) -- as one call:
// reads the finished palette through the tap and fills the viewer's `struct Vertex` array (32 bytes per vertex,
// main.cpp:50-54) with positions * pos_scale, normals and uvs.  `palette_scratch` has n_bones * 16 floats.
inline mmdx_status DeformFrame(mmdx_model_t model, mmd::Poser &poser, size_t n_bones, const float *morph_rates,
                               float *palette_scratch, float pos_scale, void *out_vertices) {
    PaletteTap::Read(poser, n_bones, palette_scratch);
    return mmdx_deform_vertex32(model, morph_rates, palette_scratch, pos_scale, out_vertices);
}

}  // namespace glue
}  // namespace mmdx
