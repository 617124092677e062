// ref_bullet_harness.cpp -- TEST INFRASTRUCTURE (checker), never part of the product.
//
// The REAL physics reactor of the reference, compiled here: libmmd's mmd::BulletPhysicsReactor
// (3rd_party/libmmd/include/mmd-bullet/mmd-bullet.hxx + mmd-bullet_impl.inl) on top of the vendored Bullet
// (3rd_party/bullet/src, its three unity files btLinearMathAll.cpp / btBulletCollisionAll.cpp / btBulletDynamicsAll.cpp),
// all #included / compiled BY PATH from /root/reference by oracle/Makefile with plain g++ -- nothing is copied, no stand-in
// header or library is involved.  It pins the physics seam of the bone solve: the viewer's frame
//     ResetPosing -> SetBonePose... -> PrePhysicsPosing -> React(1/30) -> PostPhysicsPosing -> Deform      (main.cpp:1786-1821)
// is run for a synthetic model with rigid bodies and 6-DOF spring constraints built through the mmd::Model builder API
// (Model::NewRigidBody / NewConstraint, L/model/model.inl:519-647), and per frame this harness hands out
//   * the palette as PrePhysicsPosing left it (what the reactor's kinematic bodies read),
//   * the transform every body had when React's Synchronize ran, as the skinning matrix Synchronize writes
//     (PoserMotionState::Synchronize, mmd-bullet_impl.inl:34-40: transform_ * body_transform_inv_ -> getOpenGLMatrix),
//   * the final palette after Fix (:42-56) and PostPhysicsPosing, and pose_image after Deform.
// oracle/gen_golden_bullet.py turns that into tests/golden/rig_bullet_expect.npz; the engine's
// mmdx_skeleton_solve_pre / _post (+ mmdx_deform) and the C restatement (oracle/mmdx_oracle.c physics_fix) must
// reproduce palettes and vertices bit for bit from the same poses and body transforms.
//
// The reactor keeps its motion states private; this file reads them (`#define private public` around the one libmmd
// header) to evaluate Synchronize's own expression on Bullet's own values AFTER the unmodified React() has run.
// React(), Synchronize(), Fix() and the whole of Bullet run as the reference compiled them.
// Include order as in oracle/ref_harness.cpp and for the same reason: Matrix4f::Inverse (L/util/math_impl.inl:844-861, called by
// PoserMotionState::Fix) and Bezier::interpolate call an UNQUALIFIED `abs`; the viewer's translation unit has seen <math.h> /
// <stdlib.h> (through sokol and imgui, main.cpp:10-20) before mmd.hxx (main.cpp:22), so ::abs(float) is what binds.  With
// mmd.hxx first g++ binds ::abs(int), every row scale of a rotation truncates to 0 and Inverse() returns the zero matrix --
// the first build of this harness did exactly that and every strict body's bone collapsed.  The oracle follows the application.
#include <math.h>
#include <stdlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <mmd/mmd.hxx>

#include <btBulletDynamicsCommon.h>
#define private public
#include <mmd-bullet/mmd-bullet.hxx>
#undef private

namespace {

struct BulletRef {
    mmd::Model model;
    mmd::Poser *poser = nullptr;
    mmd::BulletPhysicsReactor *reactor = nullptr;
    ~BulletRef() {
        if (reactor && poser) reactor->RemovePoser(*poser);
        delete reactor;
        delete poser;
    }
};

// the legal way to a Poser's bone images (physics.inl:32-40)
class Tap : public mmd::PhysicsReactor {
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
    static const float *Matrix(mmd::Poser &poser, size_t i) { return GetPoserBoneImage(poser, i).skinning_matrix_.v; }
};

inline mmd::Vector3f V3(const float *p) {
    mmd::Vector3f v;
    v.v[0] = p[0]; v.v[1] = p[1]; v.v[2] = p[2];
    return v;
}

}  // namespace

extern "C" {

// Bones as in mmdref_create_skeleton (FK only: flags carries 0x1000 = post-physics), vertices as in mmdref_create (no morphs),
// rigid bodies / constraints field for field as PmxReader fills them (pmx_reader_impl.inl:384-440).
void *mmdbt_create(uint32_t nb, const float *bone_pos, const int64_t *bone_parent, const int32_t *level, const uint16_t *flags,
                   uint32_t nv, const float *positions, const float *normals, const int32_t *skin_type, const int64_t *bone_ids,
                   const float *bone_weights,
                   uint32_t nrb, const int64_t *rb_bone, const uint8_t *rb_group, const uint16_t *rb_mask, const uint8_t *rb_shape,
                   const float *rb_dims, const float *rb_pos, const float *rb_rot, const float *rb_mass, const float *rb_tdamp,
                   const float *rb_rdamp, const float *rb_rest, const float *rb_fric, const uint8_t *rb_type,
                   uint32_t nc, const int64_t *c_body, const float *c_pos, const float *c_rot, const float *c_plo,
                   const float *c_phi, const float *c_rlo, const float *c_rhi, const float *c_st, const float *c_sr) {
    BulletRef *r = new BulletRef;
    mmd::Model &m = r->model;
    m.SetExtraUVNumber(0);
    for (uint32_t b = 0; b < nb; ++b) {
        mmd::Model::Bone &bone = m.NewBone();
        bone.SetName(L"b" + std::to_wstring(b));
        bone.SetPosition(V3(bone_pos + 3 * b));
        bone.SetParentIndex(bone_parent[b] < 0 ? size_t(-1) : size_t(bone_parent[b]));
        bone.SetTransformLevel(size_t(level ? level[b] : 0));
        bone.SetHasIK(false);
        bone.SetAppendRotate(false);
        bone.SetAppendTranslate(false);
        bone.SetPostPhysics(flags && (flags[b] & 0x1000) != 0);
    }
    for (uint32_t i = 0; i < nv; ++i) {
        mmd::Model::Vertex<mmd::ref> v = m.NewVertex();
        v.SetCoordinate(V3(positions + 3 * i));
        v.SetNormal(V3(normals + 3 * i));
        mmd::Model::SkinningOperator &op = v.GetSkinningOperator();
        std::memset(&op, 0, sizeof(op));
        const int64_t *id = bone_ids + 4 * i;
        const float *w = bone_weights + 4 * i;
        op.SetSkinningType(mmd::Model::SkinningOperator::SkinningType(skin_type[i]));
        if (skin_type[i] == 0) {
            op.GetBDEF1().SetBoneID(size_t(id[0]));
        } else if (skin_type[i] == 2) {
            for (int k = 0; k < 4; ++k) {
                op.GetBDEF4().SetBoneID(k, size_t(id[k]));
                op.GetBDEF4().SetBoneWeight(k, w[k]);
            }
        } else {
            op.GetBDEF2().SetBoneID(0, size_t(id[0]));
            op.GetBDEF2().SetBoneID(1, size_t(id[1]));
            op.GetBDEF2().SetBoneWeight(w[0]);
        }
    }
    for (uint32_t i = 0; i < nrb; ++i) {
        mmd::Model::RigidBody &rb = m.NewRigidBody();
        rb.SetName(L"rb" + std::to_wstring(i));
        rb.SetAssociatedBoneIndex(size_t(rb_bone[i]));
        rb.SetCollisionGroup(rb_group[i]);
        rb.GetCollisionMask() = std::bitset<16>(rb_mask[i]);
        rb.SetShape(mmd::Model::RigidBody::RigidBodyShape(rb_shape[i]));
        rb.SetDimensions(V3(rb_dims + 3 * i));
        rb.SetPosition(V3(rb_pos + 3 * i));
        rb.SetRotation(V3(rb_rot + 3 * i));
        rb.SetMass(rb_mass[i]);
        rb.SetTranslateDamp(rb_tdamp[i]);
        rb.SetRotateDamp(rb_rdamp[i]);
        rb.SetRestitution(rb_rest[i]);
        rb.SetFriction(rb_fric[i]);
        rb.SetType(mmd::Model::RigidBody::RigidBodyType(rb_type[i]));
    }
    for (uint32_t i = 0; i < nc; ++i) {
        mmd::Model::Constraint &c = m.NewConstraint();
        c.SetName(L"c" + std::to_wstring(i));
        c.SetAssociatedRigidBodyIndex(0, size_t(c_body[2 * i]));
        c.SetAssociatedRigidBodyIndex(1, size_t(c_body[2 * i + 1]));
        c.SetPosition(V3(c_pos + 3 * i));
        c.SetRotation(V3(c_rot + 3 * i));
        c.SetPositionLowLimit(V3(c_plo + 3 * i));
        c.SetPositionHighLimit(V3(c_phi + 3 * i));
        c.SetRotationLowLimit(V3(c_rlo + 3 * i));
        c.SetRotationHighLimit(V3(c_rhi + 3 * i));
        c.SetSpringTranslate(V3(c_st + 3 * i));
        c.SetSpringRotate(V3(c_sr + 3 * i));
    }
    m.Normalize();
    r->poser = new mmd::Poser(m);
    r->reactor = new mmd::BulletPhysicsReactor();     // main.cpp:666-668
    r->reactor->AddPoser(*r->poser);
    return r;
}

void mmdbt_destroy(void *h) { delete static_cast<BulletRef *>(h); }

// what kind of body each one is, as the reactor classified it (PoserMotionState ctor, mmd-bullet_impl.inl:8-17)
void mmdbt_body_info(void *h, uint8_t *passive, uint8_t *strict, uint8_t *ghost) {
    BulletRef *r = static_cast<BulletRef *>(h);
    const std::vector<mmd::BulletPhysicsReactor::PoserMotionState *> &ms = r->reactor->motion_states_[r->poser];
    for (size_t i = 0; i < ms.size(); ++i) {
        passive[i] = ms[i]->passive_; strict[i] = ms[i]->strict_; ghost[i] = ms[i]->ghost_;
    }
}

// One frame of the viewer (main.cpp:1786-1821).  poses [nb][8] = translation xyz, 0, rotation xyzw.
//   palette_pre [nb][16]  skinning matrices after PrePhysicsPosing
//   body_xf [nrb][16]     per body: transform_ * body_transform_inv_ as an OpenGL matrix, taken after React()
//                         (= what Synchronize wrote for the bodies it applies to; Fix does not touch transform_)
//   palette [nb][16]      after PostPhysicsPosing;  pos / nrm [nv][3] = pose_image after Deform
void mmdbt_frame(void *h, const float *poses, float step, float *palette_pre, float *body_xf, float *palette, float *pos,
                 float *nrm) {
    BulletRef *r = static_cast<BulletRef *>(h);
    mmd::Poser &p = *r->poser;
    const size_t nb = r->model.GetBoneNum(), nv = r->model.GetVertexNum();
    p.ResetPosing();
    for (size_t b = 0; b < nb; ++b) {
        mmd::Vector4f rot;
        for (int c = 0; c < 4; ++c) rot.v[c] = poses[8 * b + 4 + c];
        p.SetBonePose(b, mmd::Motion::BonePose(V3(poses + 8 * b), rot));
    }
    p.PrePhysicsPosing();
    for (size_t b = 0; b < nb; ++b) std::memcpy(palette_pre + 16 * b, Tap::Matrix(p, b), 64);
    r->reactor->React(step);
    const std::vector<mmd::BulletPhysicsReactor::PoserMotionState *> &ms = r->reactor->motion_states_[r->poser];
    for (size_t i = 0; i < ms.size(); ++i) {
        btTransform t = ms[i]->transform_ * ms[i]->body_transform_inv_;
        t.getOpenGLMatrix(body_xf + 16 * i);
    }
    p.PostPhysicsPosing();
    for (size_t b = 0; b < nb; ++b) std::memcpy(palette + 16 * b, Tap::Matrix(p, b), 64);
    p.Deform();
    std::memcpy(pos, p.pose_image.coordinates.data(), nv * 12);
    std::memcpy(nrm, p.pose_image.normals.data(), nv * 12);
}

}  // extern "C"
