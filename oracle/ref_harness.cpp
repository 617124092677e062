// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C driver around the REAL reference implementation: the vendored header-only libmmd that
// CU-Production/simple_mmd_renderer evaluates on the CPU every frame.  libmmd is #included BY PATH
// from /root/reference (no reference source is copied into this repo); the result is built into
// oracle/_ref/libmmd_ref.so by oracle/Makefile.  It is used to
//   (1) validate the from-scratch C restatement (oracle/mmdx_oracle.c),
//   (2) generate the golden vectors committed under tests/golden/ (oracle/gen_golden.py),
//   (3) optionally serve as bench.py's cpu_baseline (kind "reference") when the prebuilt .so is present.
//
// What is driven (reference file:line):
//   * Model builder API            L/model/model.inl:204-281, :669-689 (NewBone/NewVertex/NewMorph ...)
//   * Model::Normalize             L/model/model_impl.inl:406-452
//   * Poser ctor / posing          L/motion/poser_impl.inl:16-140, :362-394
//   * Poser::Deform                L/motion/poser_impl.inl:396-461
//   * palette tap / inject         L/motion/physics.inl:32-40 (protected GetPoserBoneImage, exactly
//                                  how include/mmd-bullet writes skinning_matrix_)
//   * 32-byte repack               main.cpp:50-54, :838-859 (app code; needs sokol, cannot be compiled
//                                  here, so the 10 lines of arithmetic are re-expressed below)
// (L/ = 3rd_party/libmmd/include/mmd/)

// Include order matters for ONE reference function: Bezier::interpolate calls an unqualified `abs`
// (L/util/math_impl.inl:1417).  The viewer's translation unit sees <math.h>/<stdlib.h> (through sokol
// and imgui, main.cpp:10-20) before mmd.hxx (main.cpp:22), so ::abs(float) is the overload it binds;
// with mmd.hxx first g++ binds ::abs(int) and every non-linear curve collapses to a constant.  The
// oracle follows the application.
#include <math.h>
#include <stdlib.h>

#include <mmd/mmd.hxx>

// the reference-side binding a maintainer would add to the viewer, compiled here against the real
// libmmd so that the GPU box can run it end to end; only its libmmd-facing half is used (nothing of libmmdx is linked)
#include "../simple_mmd_renderer_amd/host/libmmd_glue.hpp"

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// The reference's only plugin interface; deriving from it is the legal way to reach
// Poser::bone_images_[i].skinning_matrix_.
class PaletteTap : public mmd::PhysicsReactor {
public:
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

    static float *Matrix(mmd::Poser &poser, size_t i) {
        return GetPoserBoneImage(poser, i).skinning_matrix_.v;
    }
    // What BulletPhysicsReactor::React does to a poser once the world has been stepped
    // (mmd-bullet_impl.inl:312-326): PoserMotionState::Synchronize of every body physics moved (:34-40, the body's
    // transform becomes the bone's skinning matrix), then PoserMotionState::Fix of every strict one (:42-56).
    // Here the bodies' transforms are the CALLER's input (random, also non-rigid ones Bullet never produces) and the
    // two small member functions are re-expressed with libmmd's OWN matrix operators (operator*, Inverse, the vector
    // add) on libmmd's OWN BoneImage.  The REAL reactor -- mmd::BulletPhysicsReactor over the vendored Bullet, which
    // does build here with plain g++ -- runs in oracle/ref_bullet_harness.cpp and pins the same seam with its own
    // Synchronize / Fix / React (tests/test_bullet_reactor.py, tests/golden/rig_bullet_expect.npz).
    static void Synchronize(mmd::Poser &poser, size_t bone, const float *skinning) {
        std::memcpy(GetPoserBoneImage(poser, bone).skinning_matrix_.v, skinning, 64);
    }
    static void Fix(mmd::Poser &poser, size_t bone) {
        BoneImageReference t = GetPoserBoneImage(poser, bone);
        mmd::Matrix4f parent_local;
        t.local_matrix_ = t.global_offset_matrix_inv_ * t.skinning_matrix_;
        if (t.has_parent_) {
            parent_local = GetPoserBoneImage(poser, t.parent_).local_matrix_;
            t.local_matrix_ = t.local_matrix_ * parent_local.Inverse();
        }
        t.local_matrix_.r.v[3].downgrade.vector3d = t.total_translation_ + t.local_offset_;
        if (t.has_parent_) t.local_matrix_ = t.local_matrix_ * parent_local;
        t.skinning_matrix_ = t.global_offset_matrix_ * t.local_matrix_;
    }
};

struct Ref {
    mmd::Model model;
    mmd::Poser *poser = nullptr;
    ~Ref() { delete poser; }
};

inline mmd::Vector3f V3(const float *p) {
    mmd::Vector3f v;
    v.v[0] = p[0]; v.v[1] = p[1]; v.v[2] = p[2];
    return v;
}

}  // namespace

extern "C" {

// Flat model description (same meaning as include/mmdx.h's mmdx_model_desc):
//   skin_type[NV]          raw SkinningType value (0 BDEF1, 1 BDEF2, 2 BDEF4, 3 SDEF, other = "unknown")
//   bone_ids[NV][4]        int64, -1 allowed for "none" (becomes size_t(-1))
//   bone_weights[NV][4]    BDEF2/SDEF use [0]
//   sdef[NV][9] or NULL    C, R0, R1
//   bone_pos[NB][3], bone_parent[NB] (int64, -1 = root)
//   morph_type[NM], morph_off[NM+1], morph_index[E], morph_value[E][3]
//       vertex morph: index = vertex, value = offset
//       group  morph: index = morph,  value[0] = rate
//       bone   morph: index = bone,   value = translation (rotation = identity)
//       uv/ext-uv/material: index = vertex/material, value ignored beyond being stored where it fits
void *mmdref_create(uint32_t nv, uint32_t nb, uint32_t nm,
                    const float *positions, const float *normals, const float *uvs,
                    const int32_t *skin_type, const int64_t *bone_ids, const float *bone_weights,
                    const float *sdef,
                    const float *bone_pos, const int64_t *bone_parent,
                    const int32_t *morph_type, const uint32_t *morph_off,
                    const uint32_t *morph_index, const float *morph_value,
                    int normalize) {
    Ref *r = new Ref;
    mmd::Model &m = r->model;
    m.SetExtraUVNumber(0);
    for (uint32_t b = 0; b < nb; ++b) {
        mmd::Model::Bone &bone = m.NewBone();
        bone.SetName(L"b" + std::to_wstring(b));
        bone.SetPosition(V3(bone_pos + 3 * b));
        bone.SetParentIndex(bone_parent[b] < 0 ? size_t(-1) : size_t(bone_parent[b]));
        bone.SetTransformLevel(0);
        bone.SetHasIK(false);
        bone.SetAppendRotate(false);
        bone.SetAppendTranslate(false);
        bone.SetPostPhysics(false);
    }
    for (uint32_t i = 0; i < nv; ++i) {
        mmd::Model::Vertex<mmd::ref> v = m.NewVertex();
        v.SetCoordinate(V3(positions + 3 * i));
        v.SetNormal(V3(normals + 3 * i));
        if (uvs) {
            mmd::Vector2f uv;
            uv.v[0] = uvs[2 * i]; uv.v[1] = uvs[2 * i + 1];
            v.SetUVCoordinate(uv);
        }
        mmd::Model::SkinningOperator &op = v.GetSkinningOperator();
        std::memset(&op, 0, sizeof(op));
        const int64_t *id = bone_ids + 4 * i;
        const float *w = bone_weights + 4 * i;
        op.SetSkinningType(mmd::Model::SkinningOperator::SkinningType(skin_type[i]));
        switch (skin_type[i]) {
        case 0:
            op.GetBDEF1().SetBoneID(size_t(id[0]));
            break;
        case 2:
            for (int k = 0; k < 4; ++k) {
                op.GetBDEF4().SetBoneID(k, size_t(id[k]));
                op.GetBDEF4().SetBoneWeight(k, w[k]);
            }
            break;
        case 3:
            op.GetSDEF().SetBoneID(0, size_t(id[0]));
            op.GetSDEF().SetBoneID(1, size_t(id[1]));
            op.GetSDEF().SetBoneWeight(w[0]);
            if (sdef) {
                op.GetSDEF().SetC(V3(sdef + 9 * i));
                op.GetSDEF().SetR0(V3(sdef + 9 * i + 3));
                op.GetSDEF().SetR1(V3(sdef + 9 * i + 6));
            }
            break;
        default:  // BDEF2 and every unknown tag
            op.GetBDEF2().SetBoneID(0, size_t(id[0]));
            op.GetBDEF2().SetBoneID(1, size_t(id[1]));
            op.GetBDEF2().SetBoneWeight(w[0]);
            break;
        }
    }
    for (uint32_t k = 0; k < nm; ++k) {
        mmd::Model::Morph &morph = m.NewMorph();
        morph.SetName(L"m" + std::to_wstring(k));
        morph.SetType(mmd::Model::Morph::MorphType(morph_type[k]));
        for (uint32_t e = morph_off[k]; e < morph_off[k + 1]; ++e) {
            mmd::Model::Morph::MorphData &d = morph.NewMorphData();
            std::memset(&d, 0, sizeof(d));
            switch (morph_type[k]) {
            case 0:
                d.GetGroupMorph().SetMorphIndex(morph_index[e]);
                d.GetGroupMorph().SetMorphRate(morph_value[3 * e]);
                break;
            case 1:
                d.GetVertexMorph().SetVertexIndex(morph_index[e]);
                d.GetVertexMorph().SetOffset(V3(morph_value + 3 * e));
                break;
            case 2: {
                d.GetBoneMorph().SetBoneIndex(morph_index[e]);
                d.GetBoneMorph().SetTranslation(V3(morph_value + 3 * e));
                mmd::Vector4f q;
                q.q = mmd::Quaternionf::Identity();
                d.GetBoneMorph().SetRotation(q);
                break;
            }
            case 8:
                d.GetMaterialMorph().SetMaterialIndex(morph_index[e]);
                break;
            default: {  // UV / extra UV
                d.GetUVMorph().SetVertexIndex(morph_index[e]);
                mmd::Vector4f o;
                o.v[0] = morph_value[3 * e]; o.v[1] = morph_value[3 * e + 1];
                o.v[2] = morph_value[3 * e + 2]; o.v[3] = 0.f;
                d.GetUVMorph().SetOffset(o);
                break;
            }
            }
        }
    }
    if (normalize) m.Normalize();
    r->poser = new mmd::Poser(m);
    return r;
}

// The reference's own loader: FileReader + PmxReader::ReadModel (which ends with model.Normalize()),
// then Poser -- used to pin this repo's PMX parser by write -> read round trips.
// Returns NULL (and the message through mmdref_last_error) when libmmd throws.
static std::string g_ref_err;
const char *mmdref_last_error(void) { return g_ref_err.c_str(); }

void *mmdref_create_from_pmx(const char *path) {
    Ref *r = new Ref;
    try {
        std::string p(path);
        mmd::FileReader file(std::wstring(p.begin(), p.end()));
        mmd::PmxReader(file).ReadModel(r->model);
        r->poser = new mmd::Poser(r->model);
    } catch (const std::exception &e) {
        g_ref_err = e.what();
        delete r;
        return nullptr;
    }
    return r;
}

// The same through libmmd's PmdReader (the older format).
void *mmdref_create_from_pmd(const char *path) {
    Ref *r = new Ref;
    try {
        std::string p(path);
        mmd::FileReader file(std::wstring(p.begin(), p.end()));
        mmd::PmdReader(file).ReadModel(r->model);
        r->poser = new mmd::Poser(r->model);
    } catch (const std::exception &e) {
        g_ref_err = e.what();
        delete r;
        return nullptr;
    }
    return r;
}

void mmdref_get_counts(void *h, uint32_t *nv, uint32_t *nb, uint32_t *nm, uint32_t *ntri) {
    Ref *r = static_cast<Ref *>(h);
    *nv = uint32_t(r->model.GetVertexNum()); *nb = uint32_t(r->model.GetBoneNum());
    *nm = uint32_t(r->model.GetMorphNum()); *ntri = uint32_t(r->model.GetTriangleNum());
}

// Parse time of the reference's loader alone (seconds), for the loader's CPU baseline.
double mmdref_time_pmx_load(const char *path, int repeats) {
    std::string p(path);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < repeats; ++i) {
        mmd::Model model;
        mmd::FileReader file(std::wstring(p.begin(), p.end()));
        mmd::PmxReader(file).ReadModel(model);
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---- VMD: libmmd's own VmdReader + Motion --------------------------------------------------------
void *mmdref_motion_load(const char *path) {
    mmd::Motion *m = new mmd::Motion;
    try {
        std::string p(path);
        mmd::FileReader file(std::wstring(p.begin(), p.end()));
        mmd::VmdReader(file).ReadMotion(*m);
    } catch (const std::exception &e) {
        g_ref_err = e.what();
        delete m;
        return nullptr;
    }
    return m;
}
void mmdref_motion_destroy(void *h) { delete static_cast<mmd::Motion *>(h); }

// Motion::GetMorphPose for the track stored under the given Shift-JIS name (converted exactly as the
// reader converted it, byte-order mark and all); NaN if no such track.
float mmdref_motion_morph_weight(void *h, const char *sjis_name, uint32_t frame) {
    mmd::Motion *m = static_cast<mmd::Motion *>(h);
    const std::wstring key = mmd::ShiftJISToUTF16String(std::string(sjis_name));
    if (!m->IsMorphRegistered(key)) return std::nanf("");
    return m->GetMorphPose(key, size_t(frame)).GetWeight();
}

// MotionPlayer's name association (poser_impl.inl:522-537), counted.
uint32_t mmdref_motion_count_registered_morphs(void *motion, void *ref) {
    mmd::Motion *m = static_cast<mmd::Motion *>(motion);
    Ref *r = static_cast<Ref *>(ref);
    uint32_t n = 0;
    for (size_t i = 0; i < r->model.GetMorphNum(); ++i)
        if (m->IsMorphRegistered(r->model.GetMorph(i).GetName())) ++n;
    return n;
}

// Motion::GetBonePose (motion_impl.inl:255-319) for the track stored under the given Shift-JIS name:
// out = translation xyz, 0, rotation xyzw.  Returns 0 if no such track.
int mmdref_motion_bone_pose(void *h, const char *sjis_name, uint32_t frame, float *out) {
    mmd::Motion *m = static_cast<mmd::Motion *>(h);
    const std::wstring key = mmd::ShiftJISToUTF16String(std::string(sjis_name));
    if (!m->IsBoneRegistered(key)) return 0;
    const mmd::Motion::BonePose pose = m->GetBonePose(key, size_t(frame));
    for (int k = 0; k < 3; ++k) out[k] = pose.GetTranslation().v[k];
    out[3] = 0.f;
    for (int k = 0; k < 4; ++k) out[4 + k] = pose.GetRotation().v[k];
    return 1;
}

// A bones-only model (no vertices, no morphs) for the bone solve: Poser ctor ordering
// (poser_impl.inl:99-109), UpdateBoneTransform / UpdateBoneSkinningMatrix (:142-166, :320-326).
// flags = PMX bone flag word (0x20 IK, 0x100/0x200 append rotate/translate, 0x1000 post-physics);
// append_* may be NULL when no bone appends, ik_* when no bone has IK.
void *mmdref_create_skeleton(uint32_t nb, const float *bone_pos, const int64_t *bone_parent,
                             const int32_t *level, const uint16_t *flags,
                             const int64_t *append_parent, const float *append_ratio,
                             const int64_t *ik_target, const int32_t *ik_loop, const float *ik_angle,
                             const uint32_t *ik_link_off, const int64_t *ik_link_bone,
                             const uint8_t *ik_link_limited, const float *ik_link_lo, const float *ik_link_hi,
                             uint32_t nm, const int32_t *morph_type, const uint32_t *morph_off,
                             const uint32_t *morph_index, const float *morph_value, const float *morph_rotation) {
    Ref *r = new Ref;
    mmd::Model &m = r->model;
    m.SetExtraUVNumber(0);
    for (uint32_t b = 0; b < nb; ++b) {
        mmd::Model::Bone &bone = m.NewBone();
        const uint16_t f = flags ? flags[b] : 0;
        bone.SetName(L"b" + std::to_wstring(b));
        bone.SetPosition(V3(bone_pos + 3 * b));
        bone.SetParentIndex(bone_parent[b] < 0 ? size_t(-1) : size_t(bone_parent[b]));
        bone.SetTransformLevel(size_t(level ? level[b] : 0));
        bone.SetHasIK((f & 0x0020) != 0);
        if (f & 0x0020) {                                  // as PmxReader fills it, pmx_reader_impl.inl:249-263
            bone.SetIKTargetIndex(size_t(ik_target[b]));
            bone.SetCCDIterateLimit(size_t(ik_loop[b]));
            bone.SetCCDAngleLimit(ik_angle[b]);
            for (uint32_t l = ik_link_off[b]; l < ik_link_off[b + 1]; ++l) {
                mmd::Model::Bone::IKLink &link = bone.NewIKLink();
                link.SetLinkIndex(size_t(ik_link_bone[l]));
                link.SetHasLimit(ik_link_limited[l] != 0);
                if (ik_link_limited[l]) {
                    link.SetLoLimit(V3(ik_link_lo + 3 * l));
                    link.SetHiLimit(V3(ik_link_hi + 3 * l));
                }
            }
        }
        bone.SetAppendRotate((f & 0x0100) != 0);
        bone.SetAppendTranslate((f & 0x0200) != 0);
        if (f & 0x0300) {
            bone.SetAppendIndex(append_parent[b] < 0 ? size_t(-1) : size_t(append_parent[b]));
            bone.SetAppendRatio(append_ratio[b]);
        }
        bone.SetPostPhysics((f & 0x1000) != 0);
    }
    // group (0) and bone (2) morphs; every other type gets an empty vertex morph so that indices keep their meaning
    for (uint32_t k = 0; k < nm; ++k) {
        mmd::Model::Morph &morph = m.NewMorph();
        morph.SetName(L"m" + std::to_wstring(k));
        const bool used = morph_type[k] == 0 || morph_type[k] == 2;
        morph.SetType(used ? mmd::Model::Morph::MorphType(morph_type[k]) : mmd::Model::Morph::MORPH_TYPE_VERTEX);
        for (uint32_t e = morph_off[k]; used && e < morph_off[k + 1]; ++e) {
            mmd::Model::Morph::MorphData &md = morph.NewMorphData();
            if (morph_type[k] == 0) {
                md.GetGroupMorph().SetMorphIndex(size_t(morph_index[e]));
                md.GetGroupMorph().SetMorphRate(morph_value[3 * e]);
            } else {
                md.GetBoneMorph().SetBoneIndex(size_t(morph_index[e]));
                md.GetBoneMorph().SetTranslation(V3(morph_value + 3 * e));
                mmd::Vector4f rot;
                for (int c = 0; c < 4; ++c) rot.v[c] = morph_rotation ? morph_rotation[4 * e + c] : (c == 3 ? 1.f : 0.f);
                md.GetBoneMorph().SetRotation(rot);
            }
        }
    }
    r->poser = new mmd::Poser(r->model);
    return r;
}

// CPU baseline of the palette producer: per instance MotionPlayer::SeekFrame's bone half
// (GetBonePose + SetBonePose per mapped bone, poser_impl.inl:543-547) + Pre/PostPhysicsPosing.
// Bone b of the skeleton is looked up under the Shift-JIS name "b<b>".  Returns seconds.
double mmdref_time_motion_solve(void *motion, void *ref, uint32_t instances, const uint32_t *frames) {
    mmd::Motion *m = static_cast<mmd::Motion *>(motion);
    Ref *r = static_cast<Ref *>(ref);
    const size_t nb = r->model.GetBoneNum();
    std::vector<std::pair<std::wstring, size_t>> map;
    for (size_t b = 0; b < nb; ++b) {
        const std::wstring key = mmd::ShiftJISToUTF16String("b" + std::to_string(b));
        if (m->IsBoneRegistered(key)) map.push_back(std::make_pair(key, b));
    }
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < instances; ++i) {
        for (const auto &kv : map) r->poser->SetBonePose(kv.second, m->GetBonePose(kv.first, size_t(frames[i])));
        r->poser->PrePhysicsPosing();
        r->poser->PostPhysicsPosing();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void mmdref_destroy(void *h) { delete static_cast<Ref *>(h); }

// host/libmmd_glue.hpp: mmd::Model -> the flat arrays + descriptor mmdx_model_create takes (the caller creates the model
// and frees the arrays afterwards), and the palette through the glue's PhysicsReactor-derived tap.
void *mmdref_glue_flatten(void *h, mmdx_model_desc *desc, uint32_t flags) {
    mmdx::glue::FlatModel *f = new mmdx::glue::FlatModel;
    mmdx::glue::Flatten(static_cast<Ref *>(h)->model, *f);
    *desc = f->Desc(flags);
    return f;
}
void mmdref_glue_free(void *flat) { delete static_cast<mmdx::glue::FlatModel *>(flat); }
void mmdref_glue_read_palette(void *h, float *out) {
    Ref *r = static_cast<Ref *>(h);
    mmdx::glue::PaletteTap::Read(*r->poser, r->model.GetBoneNum(), out);
}

// Skin tags after the optional Normalize() -- lets tests check the load-time retagging.
void mmdref_get_skin(void *h, int32_t *type_out, int64_t *ids_out, float *w_out) {
    Ref *r = static_cast<Ref *>(h);
    size_t nv = r->model.GetVertexNum();
    for (size_t i = 0; i < nv; ++i) {
        mmd::Model::Vertex<mmd::ref> vertex = r->model.GetVertex(i);
        const mmd::Model::SkinningOperator &op = vertex.GetSkinningOperator();
        int t = int(op.GetSkinningType());
        type_out[i] = t;
        for (int k = 0; k < 4; ++k) { ids_out[4 * i + k] = -1; w_out[4 * i + k] = 0.f; }
        if (t == 0) {
            ids_out[4 * i] = int64_t(op.GetBDEF1().GetBoneID());
        } else if (t == 2) {
            for (int k = 0; k < 4; ++k) {
                ids_out[4 * i + k] = int64_t(op.GetBDEF4().GetBoneID(k));
                w_out[4 * i + k] = op.GetBDEF4().GetBoneWeight(k);
            }
        } else {
            ids_out[4 * i] = int64_t(op.GetBDEF2().GetBoneID(0));
            ids_out[4 * i + 1] = int64_t(op.GetBDEF2().GetBoneID(1));
            w_out[4 * i] = op.GetBDEF2().GetBoneWeight();
        }
    }
}

void mmdref_reset_posing(void *h) { static_cast<Ref *>(h)->poser->ResetPosing(); }

void mmdref_set_bone_pose(void *h, uint32_t i, const float *t, const float *q) {
    mmd::Vector4f rot;
    rot.v[0] = q[0]; rot.v[1] = q[1]; rot.v[2] = q[2]; rot.v[3] = q[3];
    static_cast<Ref *>(h)->poser->SetBonePose(size_t(i), mmd::Motion::BonePose(V3(t), rot));
}

void mmdref_set_morph(void *h, uint32_t i, float w) {
    static_cast<Ref *>(h)->poser->SetMorphPose(size_t(i), mmd::Motion::MorphPose(w));
}

// PrePhysicsPosing (morph accumulate + bone solve + palette) then PostPhysicsPosing.
void mmdref_pose(void *h) {
    Ref *r = static_cast<Ref *>(h);
    r->poser->PrePhysicsPosing();
    r->poser->PostPhysicsPosing();
}

// The viewer's frame with a physics reactor in it (main.cpp:1801-1810): PrePhysicsPosing, React (see
// PaletteTap::Synchronize / Fix: `skinning` [n][16] are the transforms physics produced for bones[0..n), strict[k]
// != 0 marks the bodies Fix() applies to), PostPhysicsPosing.  pre_palette (may be NULL) receives the skinning
// matrices as they stand after PrePhysicsPosing -- what the reactor's kinematic bodies read.
void mmdref_pose_physics(void *h, uint32_t n, const int64_t *bones, const uint8_t *strict, const float *skinning,
                         float *pre_palette) {
    Ref *r = static_cast<Ref *>(h);
    r->poser->PrePhysicsPosing();
    if (pre_palette) {
        const size_t nb = r->model.GetBoneNum();
        for (size_t b = 0; b < nb; ++b) std::memcpy(pre_palette + 16 * b, PaletteTap::Matrix(*r->poser, b), 64);
    }
    for (uint32_t k = 0; k < n; ++k) PaletteTap::Synchronize(*r->poser, size_t(bones[k]), skinning + 16 * size_t(k));
    for (uint32_t k = 0; k < n; ++k)
        if (strict && strict[k]) PaletteTap::Fix(*r->poser, size_t(bones[k]));
    r->poser->PostPhysicsPosing();
}

// Matrix4f::Inverse() alone (L/util/math_impl.inl:822-897), for pinning the restatements of it.
void mmdref_matrix_inverse(const float *in, float *out) {
    mmd::Matrix4f m;
    std::memcpy(m.v, in, 64);
    const mmd::Matrix4f r = m.Inverse();
    std::memcpy(out, r.v, 64);
}

void mmdref_get_palette(void *h, float *out) {
    Ref *r = static_cast<Ref *>(h);
    size_t nb = r->model.GetBoneNum();
    for (size_t b = 0; b < nb; ++b) std::memcpy(out + 16 * b, PaletteTap::Matrix(*r->poser, b), 64);
}

void mmdref_set_palette(void *h, const float *in) {
    Ref *r = static_cast<Ref *>(h);
    size_t nb = r->model.GetBoneNum();
    for (size_t b = 0; b < nb; ++b) std::memcpy(PaletteTap::Matrix(*r->poser, b), in + 16 * b, 64);
}

void mmdref_deform(void *h) { static_cast<Ref *>(h)->poser->Deform(); }

void mmdref_get_pose_image(void *h, float *pos, float *nrm) {
    Ref *r = static_cast<Ref *>(h);
    size_t nv = r->model.GetVertexNum();
    std::memcpy(pos, r->poser->pose_image.coordinates.data(), nv * 12);
    std::memcpy(nrm, r->poser->pose_image.normals.data(), nv * 12);
}

// main.cpp:838-859 arithmetic: pos * 0.1f per component, normal and uv copied.
void mmdref_repack32(void *h, float pos_scale, float *out) {
    Ref *r = static_cast<Ref *>(h);
    size_t nv = r->model.GetVertexNum();
    for (size_t i = 0; i < nv; ++i) {
        mmd::Model::Vertex<mmd::ref> vertex = r->model.GetVertex(i);
        mmd::Vector2f uv = vertex.GetUVCoordinate();
        const mmd::Vector3f &p = r->poser->pose_image.coordinates[i];
        const mmd::Vector3f &n = r->poser->pose_image.normals[i];
        float *o = out + 8 * i;
        o[0] = p.p.x * pos_scale; o[1] = p.p.y * pos_scale; o[2] = p.p.z * pos_scale;
        o[3] = n.p.x; o[4] = n.p.y; o[5] = n.p.z;
        o[6] = uv.v[0]; o[7] = uv.v[1];
    }
}

// ---- CPU-baseline timing helpers (seconds, single thread = how the reference runs) ----------

// Whole reference frame as frame() sequences it (main.cpp:1786-1825), minus motion seek/physics:
// ResetPosing -> SetMorphPose x NM -> PrePhysicsPosing -> PostPhysicsPosing -> [palette inject]
// -> Deform -> 32-B repack.  `palettes` = frames x NB x 16 floats or NULL (keep solved palette).
double mmdref_time_frames(void *h, uint32_t frames, const float *rates /*[frames][NM]*/,
                          const float *palettes, float *scratch32 /*[NV][8]*/) {
    Ref *r = static_cast<Ref *>(h);
    size_t nm = r->model.GetMorphNum(), nb = r->model.GetBoneNum();
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < frames; ++f) {
        r->poser->ResetPosing();
        for (size_t k = 0; k < nm; ++k)
            r->poser->SetMorphPose(k, mmd::Motion::MorphPose(rates[f * nm + k]));
        r->poser->PrePhysicsPosing();
        r->poser->PostPhysicsPosing();
        if (palettes) mmdref_set_palette(h, palettes + size_t(f) * nb * 16);
        r->poser->Deform();
        mmdref_repack32(h, 0.1f, scratch32);
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Crowd step: one shared morph pass, then per instance {inject palette; Deform()}.
double mmdref_time_crowd(void *h, uint32_t instances, const float *rates /*[NM]*/,
                         const float *palettes /*[instances][NB][16]*/) {
    Ref *r = static_cast<Ref *>(h);
    size_t nm = r->model.GetMorphNum(), nb = r->model.GetBoneNum();
    auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < nm; ++k) r->poser->SetMorphPose(k, mmd::Motion::MorphPose(rates[k]));
    r->poser->PrePhysicsPosing();
    r->poser->PostPhysicsPosing();
    for (uint32_t i = 0; i < instances; ++i) {
        mmdref_set_palette(h, palettes + size_t(i) * nb * 16);
        r->poser->Deform();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
