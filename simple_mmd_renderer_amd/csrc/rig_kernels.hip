// rig_kernels.hip -- per-frame producers of the bone palette, gfx950.
//   bone_track_eval_kernel : VMD bone tracks -> local poses, one thread per (instance, bone)
//   skeleton_fk_kernel     : local poses -> float[16] skinning palettes, one thread per (instance, bone)
// Both follow the reference's float operation order exactly (file built with -ffp-contract=off), so the
// palettes are bit-identical to libmmd's and the deform kernel downstream stays bit-exact end to end.
#include <hip/hip_runtime.h>

#include "rig.hpp"
#include "rig_kernels.hpp"

#include <atomic>

#include <algorithm>
#include <type_traits>

namespace mmdx {
namespace {

constexpr uint32_t kRigThreads = 256;

// Bezier<float,32>::operator[] (L/util/math_impl.inl:1379-1392): linear lookup in the presampled table.
__device__ __forceinline__ float curve_at(const float *lut, uint32_t id, float x) {
    if (id == kLinearCurve) return x;
    const float *t = lut + size_t(id) * kCurveSamples;
    x = x * float(kCurveSamples - 1);
    const uint32_t ix = uint32_t(x);
    const float r = x - float(ix);
    if (ix < kCurveSamples - 1) return (1.0f - r) * t[ix] + r * t[ix + 1];
    return t[kCurveSamples - 1];
}

// Motion::GetBonePose(name, frame), L/motion/motion_impl.inl:255-319: the local pose (translation, rotation) of one bone at one frame.
__device__ __forceinline__ void eval_bone_pose(const BoneTrackParams &p, uint32_t bone, uint32_t frame, float4 &t_out, float4 &q_out) {
    const uint32_t b = p.key_off[bone], e = p.key_off[bone + 1];
    const float4 *tr = reinterpret_cast<const float4 *>(p.key_tr);
    const float4 *rot = reinterpret_cast<const float4 *>(p.key_rot);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 1.f);   // Poser::ResetPosing
    if (e > b) {
        if (p.key_frame[b] >= frame) {
            t = tr[b]; q = rot[b];
        } else if (p.key_frame[e - 1] <= frame) {
            t = tr[e - 1]; q = rot[e - 1];
        } else {
            uint32_t lo = b, hi = e - 1;                   // key_frame[lo] <= frame < key_frame[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if (p.key_frame[mid] > frame) hi = mid; else lo = mid;
            }
            const uint32_t lf = p.key_frame[lo], rf = p.key_frame[hi];
            if (lf == frame) {
                t = tr[lo]; q = rot[lo];
            } else {
                const float bary = float(frame - lf) / float(rf - lf);
                const uint4 cv = reinterpret_cast<const uint4 *>(p.key_curve)[lo];   // the LEFT key's curves
                const float4 lt = tr[lo], rt = tr[hi], lq = rot[lo], rq = rot[hi];
                float lam = curve_at(p.lut, cv.x, bary);
                t.x = lt.x * (1.0f - lam) + rt.x * lam;
                lam = curve_at(p.lut, cv.y, bary);
                t.y = lt.y * (1.0f - lam) + rt.y * lam;
                lam = curve_at(p.lut, cv.z, bary);
                t.z = lt.z * (1.0f - lam) + rt.z * lam;
                lam = curve_at(p.lut, cv.w, bary);
                // NLerp(l, r)[lam], L/util/math_impl.inl:1260-1282
                if (lam < 1e-7f) {
                    q = lq;
                } else if (lam > 1.0f - 1e-7f) {
                    q = rq;
                } else {
                    const float dot = lq.x * rq.x + lq.y * rq.y + lq.z * rq.z + lq.w * rq.w;
                    const float a = 1.0f - lam;
                    float4 v;
                    if (dot < 0.0f) {
                        v = make_float4(a * lq.x - lam * rq.x, a * lq.y - lam * rq.y, a * lq.z - lam * rq.z,
                                        a * lq.w - lam * rq.w);
                    } else {
                        v = make_float4(a * lq.x + lam * rq.x, a * lq.y + lam * rq.y, a * lq.z + lam * rq.z,
                                        a * lq.w + lam * rq.w);
                    }
                    // Vector4D::Normalize: 1 / float(sqrt(double(sum))), L/util/math_impl.inl:717-728, math.inl:27-29
                    const float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    const float n = 1.0f / float(sqrt(double(s)));
                    q = make_float4(v.x * n, v.y * n, v.z * n, v.w * n);
                }
            }
        }
    }
    t_out = t;
    q_out = q;
}

__global__ __launch_bounds__(kRigThreads) void bone_track_eval_kernel(const BoneTrackParams p) {
    const size_t idx = size_t(blockIdx.x) * kRigThreads + threadIdx.x;
    if (idx >= size_t(p.ni) * p.nb) return;
    const uint32_t i = uint32_t(idx / p.nb), bone = uint32_t(idx - size_t(i) * p.nb);
    float4 t, q;
    eval_bone_pose(p, bone, p.frames[i], t, q);
    float4 *out = reinterpret_cast<float4 *>(p.out) + idx * 2;
    out[0] = t;
    out[1] = q;
}

struct Mat4 {
    float m[4][4];
};

// local_matrix_ of one bone before the parent product (Poser::UpdateBoneTransform,
// L/motion/poser_impl.inl:142-162, with no bone morph: morph_rotation_ = identity, morph_translation_ = 0).
__device__ __forceinline__ Mat4 local_matrix(const float4 t, const float4 r, const float4 off, const float mtx,
                                             const float mty, const float mtz, const float mi, const float mj,
                                             const float mk, const float me) {
    // total_rotation_ = morph_rotation_ * rotation_   (Quaternion::operator*, L/util/math_impl.inl:510-517)
    const float i = (me * r.x + mi * r.w + mj * r.z) - mk * r.y;
    const float j = (me * r.y + mj * r.w + mk * r.x) - mi * r.z;
    const float k = (me * r.z + mi * r.y + mk * r.w) - mj * r.x;
    const float e = me * r.w - (mi * r.x + mj * r.y + mk * r.z);
    // total_translation_ = morph_translation_ + translation_
    const float tx = mtx + t.x, ty = mty + t.y, tz = mtz + t.z;
    // Quaternion::ToRotateMatrix, L/util/math_impl.inl:540-563
    const float ii = i * i, jj = j * j, kk = k * k, ij = i * j, jk = j * k, ki = i * k, ie = i * e, je = j * e,
                ke = k * e;
    Mat4 L;
    L.m[0][0] = 1.0f - 2.0f * (jj + kk); L.m[0][1] = 2.0f * (ij + ke); L.m[0][2] = 2.0f * (ki - je); L.m[0][3] = 0.f;
    L.m[1][0] = 2.0f * (ij - ke); L.m[1][1] = 1.0f - 2.0f * (kk + ii); L.m[1][2] = 2.0f * (jk + ie); L.m[1][3] = 0.f;
    L.m[2][0] = 2.0f * (ki + je); L.m[2][1] = 2.0f * (jk - ie); L.m[2][2] = 1.0f - 2.0f * (ii + jj); L.m[2][3] = 0.f;
    L.m[3][0] = tx + off.x; L.m[3][1] = ty + off.y; L.m[3][2] = tz + off.z; L.m[3][3] = 1.f;
    return L;
}

// Matrix4x4::operator*, L/util/math_impl.inl:984-1003: every element is a left-to-right 4-term sum.
__device__ __forceinline__ Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x)
            r.m[y][x] = a.m[y][0] * b.m[0][x] + a.m[y][1] * b.m[1][x] + a.m[y][2] * b.m[2][x] + a.m[y][3] * b.m[3][x];
    return r;
}

// One thread per (instance, bone): rebuild the bone's local matrix from the root of its parent chain
// down (local(c) * local(parent), the association the reference's in-order sweep produces), then
// skinning = global_offset * local (L/motion/poser_impl.inl:320-326).  Chains are short (rig depth),
// poses and chain lists are L2-resident, and nothing synchronises: 1024 x 300 bones is one wave per CU.
template <class PoseAt>
__device__ __forceinline__ void fk_bone(const SkeletonParams &p, PoseAt pose_at, uint32_t i, uint32_t bone, float4 *out) {
    const float4 *off = reinterpret_cast<const float4 *>(p.local_offset);
    uint32_t c0 = p.chain_off[bone];
    const uint32_t c1 = p.chain_off[bone + 1];
    Mat4 M;
    bool have = false;
    if (p.chain[c0] == kIdentityParent) {      // the parent is evaluated later: still the identity
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) M.m[y][x] = x == y ? 1.f : 0.f;
        have = true;
        ++c0;
    }
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t b = p.chain[c];
        float mt[3] = {0.f, 0.f, 0.f}, mq[4] = {0.f, 0.f, 0.f, 1.f};      // no bone morph: zero / identity
        if (p.morph) {
            const float *ms = p.morph + size_t(b) * kMorphStateFloats * p.ni + i;
            mt[0] = ms[0]; mt[1] = ms[size_t(p.ni)]; mt[2] = ms[2 * size_t(p.ni)];
            mq[0] = ms[3 * size_t(p.ni)]; mq[1] = ms[4 * size_t(p.ni)]; mq[2] = ms[5 * size_t(p.ni)]; mq[3] = ms[6 * size_t(p.ni)];
        }
        const Mat4 L = local_matrix(pose_at(2 * b), pose_at(2 * b + 1), off[b], mt[0], mt[1], mt[2], mq[0], mq[1], mq[2], mq[3]);
        M = have ? mul(L, M) : L;
        have = true;
    }
    const float4 g = reinterpret_cast<const float4 *>(p.neg_rest)[bone];
    Mat4 G;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) G.m[y][x] = x == y ? 1.f : 0.f;
    G.m[3][0] = g.x; G.m[3][1] = g.y; G.m[3][2] = g.z;
    const Mat4 S = mul(G, M);
#pragma unroll
    for (int y = 0; y < 4; ++y) out[y] = make_float4(S.m[y][0], S.m[y][1], S.m[y][2], S.m[y][3]);
}

__global__ __launch_bounds__(kRigThreads) void skeleton_fk_kernel(const SkeletonParams p) {
    const size_t idx = size_t(blockIdx.x) * kRigThreads + threadIdx.x;
    if (idx >= size_t(p.ni) * p.nb) return;
    const uint32_t i = uint32_t(idx / p.nb), bone = uint32_t(idx - size_t(i) * p.nb);
    const float4 *pose = reinterpret_cast<const float4 *>(p.poses) + size_t(i) * p.nb * 2;
    fk_bone(p, [&](uint32_t k) { return pose[k]; }, i, bone, reinterpret_cast<float4 *>(p.out) + idx * 4);
}

// Bone tracks -> palette in ONE launch (mmdx_skeleton_solve_motion on a parallel-FK skeleton): a workgroup is one instance; its
// threads evaluate the bones' local poses at the instance's frame into LDS (eval_bone_pose: what bone_track_eval_kernel writes to
// HBM), one barrier, then every thread rebuilds its bones' matrices from those poses (fk_bone: what skeleton_fk_kernel does from
// HBM).  The same two functions, so the same bits; the [NI][NB][8] pose array never leaves the chip (it is still written out when
// the caller asks for it: t.out != nullptr).
__global__ __launch_bounds__(1024) void motion_fk_kernel(const BoneTrackParams t, const SkeletonParams p) {
    extern __shared__ float4 pose_lds[];                 // [nb][2]
    const uint32_t i = blockIdx.x, frame = t.frames[i];
    for (uint32_t b = threadIdx.x; b < p.nb; b += blockDim.x) {   // one bone per thread up to 1 024 bones: one latency chain, not several
        float4 tr, q;
        eval_bone_pose(t, b, frame, tr, q);
        pose_lds[2 * b] = tr;
        pose_lds[2 * b + 1] = q;
        if (t.out) {
            float4 *o = reinterpret_cast<float4 *>(t.out) + (size_t(i) * p.nb + b) * 2;
            o[0] = tr; o[1] = q;
        }
    }
    __syncthreads();
    for (uint32_t bone = threadIdx.x; bone < p.nb; bone += blockDim.x)
        fk_bone(p, [&](uint32_t k) { return pose_lds[k]; }, i, bone, reinterpret_cast<float4 *>(p.out) + (size_t(i) * p.nb + bone) * 4);
}

// ---- ordered solver: append (inherit) bones + CCD-IK -----------------------------------------------
// The reference's evaluation sequence (Poser::UpdateBoneTransform, L/motion/poser_impl.inl:142-310) with the
// per-bone state in HBM scratch laid out [bone][field][instance] (lanes = consecutive instances).  The float
// operation order is the reference's; sqrt/sin/cos/asin/acos/atan2 go through double and back like
// L/util/math.inl:27-45.  skeleton_ordered_kernel below runs the sequence; bone_morph_kernel (one thread
// per instance) prepares the bone morphs' contribution.
constexpr uint32_t kBoneMorphThreads = 64;

struct Quat {
    float i, j, k, e;
};
enum : uint32_t { kStTotalRot = 0, kStIkRot = 4, kStPreIkRot = 8, kStTotalTr = 12, kStLocal = 15 };

// floats of one lane's LDS window: fast_slots cells, padded to an odd count so the lanes of a wave fall into
// different banks
__host__ __device__ constexpr uint32_t window_floats(uint32_t fast_slots) { return (fast_slots * kSerialStateFloats) | 1u; }

template <typename Ptr, bool kCellMajor>
struct StateT {
    Ptr base;         // already offset to this lane (global scratch: + instance; LDS window: the lane's own cells)
    size_t stride;    // HBM scratch: elements between consecutive (index, field) cells = instances
    __device__ __forceinline__ auto &at(uint32_t idx, uint32_t f) const {
        if constexpr (kCellMajor) return base[idx * kSerialStateFloats + f];   // field = immediate offset
        else return base[(size_t(idx) * kSerialStateFloats + f) * stride];
    }
    __device__ __forceinline__ Quat quat(uint32_t idx, uint32_t f) const {
        return {at(idx, f), at(idx, f + 1), at(idx, f + 2), at(idx, f + 3)};
    }
    __device__ __forceinline__ void set_quat(uint32_t idx, uint32_t f, const Quat q) const {
        at(idx, f) = q.i; at(idx, f + 1) = q.j; at(idx, f + 2) = q.k; at(idx, f + 3) = q.e;
    }
    __device__ __forceinline__ Mat4 local(uint32_t idx) const {
        Mat4 m;
#pragma unroll
        for (int k = 0; k < 16; ++k) m.m[k / 4][k % 4] = at(idx, kStLocal + k);
        return m;
    }
    __device__ __forceinline__ void set_local(uint32_t idx, const Mat4 &m) const {
#pragma unroll
        for (int k = 0; k < 16; ++k) at(idx, kStLocal + k) = m.m[k / 4][k % 4];
    }
};
using State = StateT<float *, false>;                                       // per-bone state in HBM scratch
using ChainState = StateT<__attribute__((address_space(3))) float *, true>;  // one IK chain's bones in LDS, a lane's
                                                                             // cells contiguous

// float(sqrt(double(x))) is the correctly rounded f32 square root (rounding twice is innocuous for sqrt when the
// wide format has >= 2 x 24 + 2 bits), which is what sqrtf compiles to; sincos shares its argument reduction and
// polynomials with sin and cos.  Both substitutions checked over all 2^32 floats: tools/archive/probes/libm_probe.hip.
#if defined(IK_FAKE_LIBM) && !defined(MMDX_DIAGNOSTIC_BUILD)
#error "IK_FAKE_LIBM is a timing diagnostic (wrong results): add MMDX_DIAGNOSTIC_BUILD to MMDX_BUILD_DEFS"
#endif
#ifdef IK_FAKE_LIBM     // timing diagnostic only (wrong results): what the solver costs without its double-precision libm calls
__device__ __forceinline__ float d_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ void d_sincos(float x, float *s, float *c) { *s = x - x * x * x * 0.16f; *c = 1.0f - x * x * 0.5f; }
__device__ __forceinline__ float d_sin(float x) { return x - x * x * x * 0.16f; }
__device__ __forceinline__ float d_asin(float x) { return x + x * x * x * 0.16f; }
__device__ __forceinline__ float d_acos(float x) { return 1.5707964f - (x + x * x * x * 0.16f); }
__device__ __forceinline__ float d_atan2(float y, float x) { return y / (fabsf(x) + fabsf(y) + 1e-9f); }
#else
__device__ __forceinline__ float d_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ void d_sincos(float x, float *s, float *c) {
    double ds, dc;
    sincos(double(x), &ds, &dc);
    *s = float(ds);
    *c = float(dc);
}
__device__ __forceinline__ float d_sin(float x) { return float(sin(double(x))); }
__device__ __forceinline__ float d_asin(float x) { return float(asin(double(x))); }
__device__ __forceinline__ float d_acos(float x) { return float(acos(double(x))); }
__device__ __forceinline__ float d_atan2(float y, float x) { return float(atan2(double(y), double(x))); }
#endif

__device__ __forceinline__ Quat q_identity() { return {0.f, 0.f, 0.f, 1.f}; }
__device__ __forceinline__ Quat q_mul(const Quat a, const Quat q) {   // L/util/math_impl.inl:510-517
    Quat r;
    r.i = (a.e * q.i + a.i * q.e + a.j * q.k) - a.k * q.j;
    r.j = (a.e * q.j + a.j * q.e + a.k * q.i) - a.i * q.k;
    r.k = (a.e * q.k + a.i * q.j + a.k * q.e) - a.j * q.i;
    r.e = a.e * q.e - (a.i * q.i + a.j * q.j + a.k * q.k);
    return r;
}
__device__ __forceinline__ Quat q_scale(const Quat a, float s) { return {a.i * s, a.j * s, a.k * s, a.e * s}; }
__device__ __forceinline__ Quat q_inverse(const Quat a) {             // :474-477
    const float n = 1.0f / (a.i * a.i + a.j * a.j + a.k * a.k + a.e * a.e);
    return q_scale({-a.i, -a.j, -a.k, a.e}, n);
}
__device__ Quat q_slerp_from_identity(const Quat b, float l) {         // SLerp(Identity, b)[l], :1312-1337
    const Quat a = q_identity();
    float comega = a.e * b.e + a.i * b.i + a.j * b.j + a.k * b.k;
    const bool flip = comega < 0.0f;
    if (flip) comega = -comega;
    const float omega = d_acos(comega);
    if (omega > 1e-7f) {
        const float rs = 1.0f / d_sin(omega);
        const float p = d_sin((1.0f - l) * omega) * rs;
        l = d_sin(l * omega) * rs;
        if (flip) l = -l;
        const Quat x = q_scale(a, p), y = q_scale(b, l);
        return {x.i + y.i, x.j + y.j, x.k + y.k, x.e + y.e};
    }
    return a;
}
__device__ __forceinline__ Mat4 q_to_matrix(const Quat q) {           // :540-563
    const float ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k, ij = q.i * q.j, jk = q.j * q.k, ki = q.i * q.k;
    const float ie = q.i * q.e, je = q.j * q.e, ke = q.k * q.e;
    Mat4 L;
    L.m[0][0] = 1.0f - 2.0f * (jj + kk); L.m[0][1] = 2.0f * (ij + ke); L.m[0][2] = 2.0f * (ki - je); L.m[0][3] = 0.f;
    L.m[1][0] = 2.0f * (ij - ke); L.m[1][1] = 1.0f - 2.0f * (kk + ii); L.m[1][2] = 2.0f * (jk + ie); L.m[1][3] = 0.f;
    L.m[2][0] = 2.0f * (ki + je); L.m[2][1] = 2.0f * (jk - ie); L.m[2][2] = 1.0f - 2.0f * (ii + jj); L.m[2][3] = 0.f;
    L.m[3][0] = 0.f; L.m[3][1] = 0.f; L.m[3][2] = 0.f; L.m[3][3] = 1.f;
    return L;
}
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v_normalize(const V3 v) {                // :390-400
    const float n = 1.0f / d_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x * n, v.y * n, v.z * n};
}
__device__ __forceinline__ float v_dot(const V3 a, const V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ Quat axis_to_quat(const V3 axis, float angle) {             // :1047-1058
    const float norm = d_sqrt(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
    if (norm < 1e-7f) return q_identity();
    angle *= 0.5f;
    float sn, cs;
    d_sincos(angle, &sn, &cs);
    const float s = sn / norm;
    return {s * axis.x, s * axis.y, s * axis.z, cs};
}

// The three Euler orders share one shape -- one asin and two atan2 of products of the quaternion's components --
// so the operands are selected per order and each function is called once: lanes of a wave that solve chains
// with different orders then do not take turns through three copies of the double-precision calls.
__device__ __forceinline__ void quat_to_euler(uint32_t order, const Quat q, float *r) {   // :1059-1071, :1110-1135
    const float ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k;
    const float ei = q.e * q.i, ej = q.e * q.j, ek = q.e * q.k;
    const float ij = q.i * q.j, ik = q.i * q.k, jk = q.j * q.k;
    const bool zxy = order == kOrderZXY, xyz = order == kOrderXYZ;
    // component 0: ZXY asin(2(ei + jk)); XYZ atan2(2(ei - jk), 1 - 2(ii + jj)); YZX atan2(2(ei - jk), 1 - 2(ii + kk))
    // component 1: ZXY atan2(2(ej - ik), 1 - 2(ii + jj)); XYZ asin(2(ej + ik)); YZX atan2(2(ej - ik), 1 - 2(jj + kk))
    // component 2: ZXY atan2(2(ek - ij), 1 - 2(ii + kk)); XYZ atan2(2(ek - ij), 1 - 2(jj + kk)); YZX asin(2(ek + ij))
    const float s_arg = zxy ? 2.0f * (ei + jk) : xyz ? 2.0f * (ej + ik) : 2.0f * (ek + ij);
    const float a_y = zxy ? 2.0f * (ej - ik) : 2.0f * (ei - jk);               // first atan2: component 1 (ZXY) or 0
    const float a_x = zxy ? 1 - 2.0f * (ii + jj) : xyz ? 1 - 2.0f * (ii + jj) : 1 - 2.0f * (ii + kk);
    const float b_y = (zxy || xyz) ? 2.0f * (ek - ij) : 2.0f * (ej - ik);       // second atan2: component 2 or 1 (YZX)
    const float b_x = zxy ? 1 - 2.0f * (ii + kk) : 1 - 2.0f * (jj + kk);
    const float s = d_asin(s_arg), a = d_atan2(a_y, a_x), b = d_atan2(b_y, b_x);
    r[0] = zxy ? s : a;
    r[1] = zxy ? a : xyz ? s : b;
    r[2] = zxy ? b : xyz ? b : s;
}
__device__ __forceinline__ Quat euler_to_quat(uint32_t order, const float *r) {           // :1137-1149, :1176-1201
    float cx, sx, cy, sy, cz, sz;
    d_sincos(r[0] * 0.5f, &sx, &cx);
    d_sincos(r[1] * 0.5f, &sy, &cy);
    d_sincos(r[2] * 0.5f, &sz, &cz);
    // every component is (product of three) +- (product of three): the orders differ in the signs only
    const float ia = sx * cy * cz, ib = cx * sy * sz, ja = cx * sy * cz, jb = sx * cy * sz, ka = cx * cy * sz, kb = sx * sy * cz;
    Quat q;
    q.e = cx * cy * cz - sx * sy * sz;
    q.i = order == kOrderZXY ? ia - ib : ia + ib;
    q.j = order == kOrderXYZ ? ja - jb : ja + jb;
    q.k = order == kOrderYZX ? ka - kb : ka + kb;
    return q;
}
__device__ __forceinline__ void limit_euler(float *e, const float *lo, const float *hi, bool ikt) {   // poser_impl.inl:178-194
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (e[i] < lo[i]) {
            const float tf = 2 * lo[i] - e[i];
            e[i] = (tf <= hi[i] && ikt) ? tf : lo[i];
        }
        if (e[i] > hi[i]) {
            const float tf = 2 * hi[i] - e[i];
            e[i] = (tf >= lo[i] && ikt) ? tf : hi[i];
        }
    }
}

// local_matrix_ from the total rotation / translation (the values just stored in the state, passed along so
// they are not read back), then the parent product.  `self` / `parent` index the state (bone ids in HBM
// scratch, chain slots in LDS); parent < 0 = none.
template <class S>
__device__ void place_at(const S &st, const V3 local_offset, const Quat total, const V3 tr, uint32_t self, int32_t parent) {
    Mat4 L = q_to_matrix(total);
    L.m[3][0] = tr.x + local_offset.x;
    L.m[3][1] = tr.y + local_offset.y;
    L.m[3][2] = tr.z + local_offset.z;
    if (parent >= 0) L = mul(L, st.local(uint32_t(parent)));
    st.set_local(self, L);
}

// UpdateBoneTransform up to (not including) the IK solve, poser_impl.inl:142-166.  `ap` = state index of
// the append parent (read only when the bone appends).
struct MorphXf {                                  // morph_translation_, morph_rotation_ of one bone
    float tx, ty, tz;
    Quat q;
};
__device__ __forceinline__ MorphXf morph_of(const SerialParams &p, uint32_t bone, uint32_t inst) {
    if (!p.morph) return {0.f, 0.f, 0.f, q_identity()};
    const float *ms = p.morph + size_t(bone) * kMorphStateFloats * p.ni + inst;
    const size_t n = p.ni;
    return {ms[0], ms[n], ms[2 * n], {ms[3 * n], ms[4 * n], ms[5 * n], ms[6 * n]}};
}

template <class S>
__device__ void transform_at(const S &st, const BoneRec &rec, const MorphXf mx, const float4 t, const float4 r,
                             uint32_t self, int32_t parent, uint32_t ap) {
    Quat total = q_mul(mx.q, {r.x, r.y, r.z, r.w});
    float tx = mx.tx + t.x, ty = mx.ty + t.y, tz = mx.tz + t.z;
    if (rec.bits & (kBoneAppendRot | kBoneAppendTr)) {
        // the reference assigns total_rotation_ / total_translation_ before reading the append parent's,
        // which matters when a bone names itself: keep that order
        st.set_quat(self, kStTotalRot, total);
        st.at(self, kStTotalTr + 0) = tx; st.at(self, kStTotalTr + 1) = ty; st.at(self, kStTotalTr + 2) = tz;
        if (rec.bits & kBoneAppendRot) {
            total = q_mul(total, q_slerp_from_identity(st.quat(ap, kStTotalRot), rec.append_ratio));
            st.set_quat(self, kStTotalRot, total);
        }
        if (rec.bits & kBoneAppendTr) {
            tx = tx + rec.append_ratio * st.at(ap, kStTotalTr + 0);
            ty = ty + rec.append_ratio * st.at(ap, kStTotalTr + 1);
            tz = tz + rec.append_ratio * st.at(ap, kStTotalTr + 2);
        }
    }
    if (rec.bits & kBoneIsIkLink) {
        st.set_quat(self, kStPreIkRot, total);
        total = q_mul(st.quat(self, kStIkRot), total);
    }
    st.set_quat(self, kStTotalRot, total);
    st.at(self, kStTotalTr + 0) = tx; st.at(self, kStTotalTr + 1) = ty; st.at(self, kStTotalTr + 2) = tz;
    place_at(st, V3{rec.local_offset[0], rec.local_offset[1], rec.local_offset[2]}, total, V3{tx, ty, tz}, self, parent);
}

__device__ __forceinline__ void transform_bone(const State &st, const SerialParams &p, const float4 *pose, uint32_t inst,
                                               uint32_t b) {
    const BoneRec rec = p.bones[b];
    transform_at(st, rec, morph_of(p, b, inst), pose[2 * size_t(b)], pose[2 * size_t(b) + 1], b, rec.parent,
                 uint32_t(rec.append_parent));
}

// What the CCD loop needs to know about link j of its chain: where it and its parent live in the state, its
// rest offset and its limits.  On the HBM state these come from the rig tables (indices = bone ids); on the
// LDS window the chain's constants were copied next to it (index j = link j, see solve_ik).
struct LinkInfo {
    uint32_t limited, order, fix;
    float lo[3], hi[3];
};
constexpr uint32_t kLinkConstFloats = 10;       // local offset xyz, limited | order << 8 | fix << 16, lo xyz, hi xyz

struct TableChain {
    const SerialParams &p;
    const LinkRec *links;
    __device__ __forceinline__ uint32_t idx(uint32_t j) const { return links[j].bone; }
    __device__ __forceinline__ int32_t par(uint32_t j) const { return p.bones[links[j].bone].parent; }
    __device__ __forceinline__ V3 offset(uint32_t j) const {
        const float *o = p.bones[links[j].bone].local_offset;
        return {o[0], o[1], o[2]};
    }
    __device__ __forceinline__ LinkInfo link(uint32_t j) const {
        const LinkRec lk = links[j];
        return {lk.limited, lk.order, lk.fix, {lk.lo[0], lk.lo[1], lk.lo[2]}, {lk.hi[0], lk.hi[1], lk.hi[2]}};
    }
};
struct WindowChain {
    const __attribute__((address_space(3))) float *consts;     // [link][kLinkConstFloats], shared by the window's lanes
    uint32_t n;
    int32_t outside;                                           // window slot of the root-most link's parent, -1 = none
    __device__ __forceinline__ uint32_t idx(uint32_t j) const { return j; }
    __device__ __forceinline__ int32_t par(uint32_t j) const { return j + 1 < n ? int32_t(j + 1) : outside; }
    __device__ __forceinline__ V3 offset(uint32_t j) const {
        const auto *c = consts + j * kLinkConstFloats;
        return {c[0], c[1], c[2]};
    }
    __device__ __forceinline__ LinkInfo link(uint32_t j) const {
        const auto *c = consts + j * kLinkConstFloats;
        const uint32_t w = __float_as_uint(c[3]);
        return {w & 0xFFu, (w >> 8) & 0xFFu, w >> 16, {c[4], c[5], c[6]}, {c[7], c[8], c[9]}};
    }
};

// The CCD loop of UpdateBoneTransform, poser_impl.inl:196-309, over a state `st` in which the chain `ch` says
// where link j and its parent live, the target at tidx, its parent at tpar (< 0 = none).
// Nested IK (`nested` != 0, HBM state only): a link or the target that is itself an IK bone gets its own solve right
// after it has been transformed, exactly where UpdateBoneTransform would recurse -- for the links once, when the outer
// solve places them, for the target every time the loop re-places it.  DEPTH counts the solves on the stack; rig.cpp
// rejects rigs that would exceed kMaxIkDepth.
template <int DEPTH>
__device__ __noinline__ void solve_ik_nested(const State &st, const SerialParams &p, const float4 *pose, uint32_t inst, uint32_t b);

template <class S, class Chain, int DEPTH = 0>
__device__ void ccd(const S &st, const SerialParams &p, const float4 *pose, uint32_t inst, const IkRec &ik,
                    const LinkRec *links, const V3 ik_pos, const Chain &ch, uint32_t tidx, int32_t tpar) {
    // DEPTH 0: the rig has no nested IK at all -- no call site, no stack, the register allocation of the plain solver
    constexpr bool kCanNest = std::is_same<S, State>::value && DEPTH >= 1 && DEPTH < int(kMaxIkDepth);
    auto inner = [&](uint32_t bone, uint32_t bits) {
        if constexpr (kCanNest) {
            if (ik.nested && (bits & kBoneHasIk)) solve_ik_nested<DEPTH + 1>(st, p, pose, inst, bone);
        }
    };
    const BoneRec trec = p.bones[ik.target];
    const float4 tt = pose[2 * size_t(ik.target)], tr = pose[2 * size_t(ik.target) + 1];
    const MorphXf tmx = morph_of(p, ik.target, inst);
    for (uint32_t i = 0; i < ik.nlinks; ++i) st.set_quat(ch.idx(i), kStIkRot, q_identity());
    for (uint32_t i = 0; i < ik.nlinks; ++i) {
        const uint32_t j = ik.nlinks - i - 1, lb = links[j].bone;
        const BoneRec rec = p.bones[lb];
        transform_at(st, rec, morph_of(p, lb, inst), pose[2 * size_t(lb)], pose[2 * size_t(lb) + 1], ch.idx(j), ch.par(j),
                     uint32_t(rec.append_parent));
        inner(lb, rec.bits);
    }
    transform_at(st, trec, tmx, tt, tr, tidx, tpar, uint32_t(trec.append_parent));
    inner(ik.target, trec.bits);
    V3 tgt = {st.at(tidx, kStLocal + 12), st.at(tidx, kStLocal + 13), st.at(tidx, kStLocal + 14)};
    V3 err = {ik_pos.x - tgt.x, ik_pos.y - tgt.y, ik_pos.z - tgt.z};
    if (v_dot(err, err) < 1e-7f) return;
    const uint32_t ikt = ik.loop / 2;
    // A window chain (rig.cpp: link j hangs off link j+1, the target off link 0, no append bone, no nested solve) lets the loop
    // keep what cannot change during the solve, with the same values the reference recomputes every time: the target's own
    // rotation and translation (only its PARENT moves), the total rotation of the links below the one being turned, and the matrix
    // of the link placed last, which is the next one's parent -- handed on in registers instead of through the state.
    constexpr bool kWindow = std::is_same<S, ChainState>::value;
    Mat4 t_pre = {};                                     // the target's local matrix before its parent product
    if constexpr (kWindow) {
        t_pre = q_to_matrix(st.quat(tidx, kStTotalRot));
        t_pre.m[3][0] = st.at(tidx, kStTotalTr + 0) + trec.local_offset[0];
        t_pre.m[3][1] = st.at(tidx, kStTotalTr + 1) + trec.local_offset[1];
        t_pre.m[3][2] = st.at(tidx, kStTotalTr + 2) + trec.local_offset[2];
    }
    for (uint32_t i = 0; i < ik.loop; ++i) {
        bool changed = false;                               // some link's IK rotation came out different, bit for bit, in this sweep
        for (uint32_t j = 0; j < ik.nlinks; ++j) {
            const LinkInfo lk = ch.link(j);
            if (lk.fix == kFixAll) continue;
            const uint32_t ls = ch.idx(j);
            const int32_t lp = ch.par(j);
            const V3 lpos = {st.at(ls, kStLocal + 12), st.at(ls, kStLocal + 13), st.at(ls, kStLocal + 14)};
            const V3 tdir = v_normalize({lpos.x - tgt.x, lpos.y - tgt.y, lpos.z - tgt.z});
            const V3 idir = v_normalize({lpos.x - ik_pos.x, lpos.y - ik_pos.y, lpos.z - ik_pos.z});
            V3 axis = {tdir.y * idir.z - tdir.z * idir.y, tdir.z * idir.x - tdir.x * idir.z,
                       tdir.x * idir.y - tdir.y * idir.x};
            if (fabsf(axis.x) < 1e-7f) axis.x = 1e-7f;
            if (fabsf(axis.y) < 1e-7f) axis.y = 1e-7f;
            if (fabsf(axis.z) < 1e-7f) axis.z = 1e-7f;
            Mat4 loc;
            if (lp >= 0) {
                loc = st.local(uint32_t(lp));
            } else {
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int x = 0; x < 4; ++x) loc.m[y][x] = x == y ? 1.f : 0.f;
            }
            if (lk.limited && lk.fix != kFixNone && i < ikt) {
                const uint32_t row = lk.fix - kFixX;       // selects, not loc.m[row]: a lane-varying index would
                                                            // put the matrix in scratch memory
                const float m0 = row == 0 ? loc.m[0][0] : row == 1 ? loc.m[1][0] : loc.m[2][0];
                const float m1 = row == 0 ? loc.m[0][1] : row == 1 ? loc.m[1][1] : loc.m[2][1];
                const float m2 = row == 0 ? loc.m[0][2] : row == 1 ? loc.m[1][2] : loc.m[2][2];
                const float d = axis.x * m0 + axis.y * m1 + axis.z * m2;
                const float sgn = d >= 0.0f ? 1.0f : -1.0f;
                axis = {row == 0 ? sgn : 0.f, row == 1 ? sgn : 0.f, row == 2 ? sgn : 0.f};
            } else {                                       // rotate(axis, loc.Transpose()).Normalize()
                const V3 r = {axis.x * loc.m[0][0] + axis.y * loc.m[0][1] + axis.z * loc.m[0][2],
                              axis.x * loc.m[1][0] + axis.y * loc.m[1][1] + axis.z * loc.m[1][2],
                              axis.x * loc.m[2][0] + axis.y * loc.m[2][1] + axis.z * loc.m[2][2]};
                axis = v_normalize(r);
            }
            float dot = v_dot(tdir, idir);
            dot = dot < -1.0f ? -1.0f : dot;               // math::clamp = min(max(x, lo), hi)
            dot = 1.0f < dot ? 1.0f : dot;
            const float ac = d_acos(dot), cap = ik.angle_limit * float(j + 1);
            const float angle = cap < ac ? cap : ac;
            const Quat ikr_old = st.quat(ls, kStIkRot);
            Quat ikr = q_mul(axis_to_quat(axis, angle), ikr_old);
            if (lk.limited) {
                const Quat pre = st.quat(ls, kStPreIkRot);
                Quat lr = q_mul(ikr, pre);
                float e[3];
                quat_to_euler(lk.order, lr, e);
                limit_euler(e, lk.lo, lk.hi, i < ikt);
                lr = euler_to_quat(lk.order, e);
                ikr = q_mul(lr, q_inverse(pre));
            }
            st.set_quat(ls, kStIkRot, ikr);
            if constexpr (kWindow)
                changed = changed || __float_as_uint(ikr.i) != __float_as_uint(ikr_old.i) || __float_as_uint(ikr.j) != __float_as_uint(ikr_old.j) ||
                          __float_as_uint(ikr.k) != __float_as_uint(ikr_old.k) || __float_as_uint(ikr.e) != __float_as_uint(ikr_old.e);
            if constexpr (kWindow) {
                Mat4 prev = {};
                for (uint32_t k = 0; k <= j; ++k) {
                    const uint32_t jj = j - k, bs = ch.idx(jj);
                    Quat total;
                    if (k == 0) {                              // the link just turned: ik_rotation * pre-IK rotation, as the reference
                        total = q_mul(ikr, st.quat(bs, kStPreIkRot));
                        st.set_quat(bs, kStTotalRot, total);
                    } else {                                   // below it nothing changed: the stored product IS the recomputed one
                        total = st.quat(bs, kStTotalRot);
                    }
                    Mat4 L = q_to_matrix(total);
                    const V3 off = ch.offset(jj);
                    L.m[3][0] = st.at(bs, kStTotalTr + 0) + off.x;
                    L.m[3][1] = st.at(bs, kStTotalTr + 1) + off.y;
                    L.m[3][2] = st.at(bs, kStTotalTr + 2) + off.z;
                    if (k == 0) { if (lp >= 0) L = mul(L, loc); }   // its parent's matrix was read for the axis a moment ago
                    else L = mul(L, prev);                     // link jj's parent is link jj+1, placed a moment ago
                    st.set_local(bs, L);
                    prev = L;
                }
                const Mat4 T = mul(t_pre, prev);               // the target hangs off link 0, the last one placed
                st.set_local(tidx, T);
                tgt = {T.m[3][0], T.m[3][1], T.m[3][2]};
            } else {
                for (uint32_t k = 0; k <= j; ++k) {
                    const uint32_t jj = j - k, bs = ch.idx(jj);
                    const Quat total = q_mul(st.quat(bs, kStIkRot), st.quat(bs, kStPreIkRot));
                    st.set_quat(bs, kStTotalRot, total);
                    place_at(st, ch.offset(jj), total, V3{st.at(bs, kStTotalTr + 0), st.at(bs, kStTotalTr + 1), st.at(bs, kStTotalTr + 2)},
                             bs, ch.par(jj));
                }
                transform_at(st, trec, tmx, tt, tr, tidx, tpar, uint32_t(trec.append_parent));
                inner(ik.target, trec.bits);
                tgt = {st.at(tidx, kStLocal + 12), st.at(tidx, kStLocal + 13), st.at(tidx, kStLocal + 14)};
            }
        }
        err = {ik_pos.x - tgt.x, ik_pos.y - tgt.y, ik_pos.z - tgt.z};
        if (v_dot(err, err) < 1e-7f) return;
        // A sweep that reproduced every link's IK rotation bit for bit left the whole chain as it found it (each link's matrices
        // follow from its rotation and its parent's matrix), so every later sweep with the same `i < ikt` does the same again: the
        // rest of that half of the iterations is skipped -- the reference would run them and change nothing.  (A NaN never compares
        // equal, so NaN chains run to the end like the reference's.)  Window chains only: nothing outside the window is involved.
        if constexpr (kWindow) {
            if (!changed) {
                if (i >= ikt) return;
                i = ikt - 1;                                // fixed during the first half: go on with the second half's rules
            }
        }
    }
}

template <int DEPTH>
__device__ __noinline__ void solve_ik_nested(const State &st, const SerialParams &p, const float4 *pose, uint32_t inst, uint32_t b) {
    const IkRec ik = p.iks[p.bones[b].ik];
    const LinkRec *links = p.links + ik.link0;
    const V3 ik_pos = {st.at(b, kStLocal + 12), st.at(b, kStLocal + 13), st.at(b, kStLocal + 14)};
    ccd<State, TableChain, DEPTH>(st, p, pose, inst, ik, links, ik_pos, TableChain{p, links}, ik.target, p.bones[ik.target].parent);
}

// One IK bone.  Chains with the usual topology (every link's parent is the next link, the target hangs off
// the first link, no append bones inside; IkRec::fast, decided on the host) are solved on an LDS window:
// their few bones' state and the links' constants are copied in, the up-to-256-iteration loop runs at LDS
// latency instead of paying a round trip to the tables and the HBM scratch for every dependent access, and
// the result is copied back.  Same arithmetic.  `lds_lane` = this lane's cell of the window's state,
// `lds_consts` = the window's constants (the lanes of a window solve the same chain and write the same values).
template <bool NESTED>
__device__ void solve_ik(const State &st, const SerialParams &p, const float4 *pose, uint32_t inst, uint32_t b,
                         __attribute__((address_space(3))) float *lds_lane,
                         __attribute__((address_space(3))) float *lds_consts) {
    const IkRec ik = p.iks[p.bones[b].ik];
    const LinkRec *links = p.links + ik.link0;
    const V3 ik_pos = {st.at(b, kStLocal + 12), st.at(b, kStLocal + 13), st.at(b, kStLocal + 14)};
    if (ik.fast) {
        const ChainState cs = {lds_lane, 0};
        const uint32_t n = ik.nlinks;
        const int32_t outside = ik.outside_parent;         // parent of the root-most link, outside the chain
        auto copy = [&](uint32_t slot, uint32_t bone, bool in) {
#pragma unroll
            for (uint32_t f = 0; f < kSerialStateFloats; ++f) {
                if (in) cs.at(slot, f) = st.at(bone, f); else st.at(bone, f) = cs.at(slot, f);
            }
        };
        for (uint32_t j = 0; j < n; ++j) {
            const LinkRec lk = links[j];
            copy(j, lk.bone, true);
            const float *off = p.bones[lk.bone].local_offset;
            auto *c = lds_consts + j * kLinkConstFloats;
            c[0] = off[0]; c[1] = off[1]; c[2] = off[2];
            c[3] = __uint_as_float(lk.limited | lk.order << 8 | lk.fix << 16);
            c[4] = lk.lo[0]; c[5] = lk.lo[1]; c[6] = lk.lo[2];
            c[7] = lk.hi[0]; c[8] = lk.hi[1]; c[9] = lk.hi[2];
        }
        copy(n, ik.target, true);
        if (outside >= 0) copy(n + 1, uint32_t(outside), true);
        ccd(cs, p, pose, inst, ik, links, ik_pos, WindowChain{lds_consts, n, outside >= 0 ? int32_t(n + 1) : -1}, n, 0);
        for (uint32_t j = 0; j < n; ++j) copy(j, links[j].bone, false);
        copy(n, ik.target, false);
    } else {
        ccd<State, TableChain, NESTED ? 1 : 0>(st, p, pose, inst, ik, links, ik_pos, TableChain{p, links}, ik.target,
                                               p.bones[ik.target].parent);
    }
}

// Bone morphs -> per-bone morph_translation_ / morph_rotation_ (Poser::UpdateMorphTransform, MORPH_TYPE_BONE,
// L/motion/poser_impl.inl:347-354, after the reset of :369-370).  One thread per instance walks the
// applications in the reference's order (the quaternion products of one bone do not commute); rates below
// 1e-7 skip at every group level like the vertex morphs.  Output [bone][7][instance].
__global__ __launch_bounds__(kBoneMorphThreads) void bone_morph_kernel(const BoneMorphParams p) {
    const uint32_t inst = blockIdx.x * kBoneMorphThreads + threadIdx.x;
    if (inst >= p.ni) return;
    float *out = p.out + inst;
    const size_t n = p.ni;
    for (uint32_t b = 0; b < p.nb; ++b) {
        float *o = out + size_t(b) * kMorphStateFloats * n;
        o[0] = 0.f; o[n] = 0.f; o[2 * n] = 0.f;
        o[3 * n] = 0.f; o[4 * n] = 0.f; o[5 * n] = 0.f; o[6 * n] = 1.f;
    }
    const float *rates = p.rates + (p.shared ? 0 : size_t(inst) * p.nm);
    for (uint32_t a = 0; a < p.napps; ++a) {
        const BoneMorphApp app = p.apps[a];
        float r = rates[app.top];
        bool skip = r < 1e-7f;
        for (uint32_t c = 0; !skip && c < app.chain_len; ++c) {
            r = p.chain[app.chain_off + c] * r;
            skip = r < 1e-7f;
        }
        if (skip) continue;
        float *o = out + size_t(app.bone) * kMorphStateFloats * n;
        o[0] = o[0] + app.tr[0] * r; o[n] = o[n] + app.tr[1] * r; o[2 * n] = o[2 * n] + app.tr[2] * r;
        const Quat cur = {o[3 * n], o[4 * n], o[5 * n], o[6 * n]};
        const Quat q = q_mul(cur, q_slerp_from_identity({app.rot[0], app.rot[1], app.rot[2], app.rot[3]}, r));
        o[3 * n] = q.i; o[4 * n] = q.j; o[5 * n] = q.k; o[6 * n] = q.e;
    }
}

// The ordered solver.  A workgroup holds kSolveInstances instances x kSolveSlots slots (instance fastest, so
// the 16 lanes of a slot touch one 64-byte run of every state cell); the host cut the evaluation sequence
// into rounds of events that touch disjoint bones (rig.cpp schedule_rounds), slot k of every instance runs
// event k of the round, and a workgroup barrier separates the rounds.  Every event is the reference's
// UpdateBoneTransform on the state the serial sequence would have shown it: same arithmetic, same result.
// NESTED: the rig has an IK bone among some chain's links / targets (rig.cpp): the variant with the nested-solve call
// sites (a stack, all 256 VGPRs); rigs without -- nearly all -- run the variant that has none.
// DENSE: the register allocation held to 256 VGPRs (the plain variant takes 292: one workgroup per CU) so that, with the
// 72 KB LDS budget of the windows (rig.hpp), TWO workgroups share a CU.  A lone wave per SIMD is all a crowd of up to
// 16 x (number of CUs) instances can use, and there the plain variant is 5 % faster (37 spilled registers); beyond that the
// second wave per SIMD is worth 1.5 x the palettes per second (16 384 instances: 7.6 instead of 11.2 ms).  Chosen per launch.
template <bool NESTED, bool DENSE>
__global__ __launch_bounds__(kSolveInstances * kSolveSlots, DENSE ? 2 : 1) void skeleton_ordered_kernel(const SerialParams p) {
    const uint32_t slot = threadIdx.x / kSolveInstances;
    const uint32_t inst = blockIdx.x * kSolveInstances + threadIdx.x % kSolveInstances;
    const bool live = inst < p.ni;       // dead lanes skip the work but reach every barrier
    const State st = {p.state + (live ? inst : 0), p.ni};
    extern __shared__ float chain_lds[];   // (windows x instances) lanes x window_floats of state, then the
                                           // windows' link constants
    // Which event of a round this slot runs: the events are dealt over the WAVES first (slot 4w + k runs event 4k + w), so that
    // a round's IK solves -- the first events of the round -- sit in different waves as far as possible: lanes of one wave that
    // solve DIFFERENT chains take turns through every divergent piece of the CCD loop (two chains per wave instead of four on the
    // bench rig: measured).  The LDS windows belong to the round's first p.windows events, whichever slot runs them.
    constexpr uint32_t kSlotsPerWave = 64 / kSolveInstances, kSolveWaves = kSolveSlots / kSlotsPerWave;
    const uint32_t ev = (slot % kSlotsPerWave) * kSolveWaves + slot / kSlotsPerWave;
    const uint32_t wf = window_floats(p.fast_slots);
    auto *lds_lane = (__attribute__((address_space(3))) float *)chain_lds + (ev * kSolveInstances + threadIdx.x % kSolveInstances) * wf;
    auto *lds_consts = (__attribute__((address_space(3))) float *)chain_lds +
                       size_t(p.windows) * kSolveInstances * wf + ev * (kMaxFastLinks * kLinkConstFloats);
    const float4 *pose = reinterpret_cast<const float4 *>(p.poses) + size_t(live ? inst : 0) * p.nb * 2;
    if (live && (p.passes & 1u) && (p.seg_flags & 1u)) {
        for (uint32_t b = slot; b < p.nb; b += kSolveSlots) {   // PrePhysicsPosing's reset, poser_impl.inl:366-377
            st.set_quat(b, kStTotalRot, q_identity());
            st.set_quat(b, kStIkRot, q_identity());
            st.set_quat(b, kStPreIkRot, q_identity());
            st.at(b, kStTotalTr + 0) = 0.f; st.at(b, kStTotalTr + 1) = 0.f; st.at(b, kStTotalTr + 2) = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) st.at(b, kStLocal + k) = (k % 5 == 0) ? 1.f : 0.f;
        }
    }
    __syncthreads();
    float4 *out = reinterpret_cast<float4 *>(p.out) + size_t(live ? inst : 0) * p.nb * 4;
    for (uint32_t pass = 0; pass < 2; ++pass) {
        if (!(p.passes >> pass & 1u)) continue;              // the physics seam runs the two lists as two launches
        const uint32_t r0 = max(pass ? p.n_rounds_pre : 0u, p.seg_r0), r1 = min(pass ? p.n_rounds : p.n_rounds_pre, p.seg_r1);
        for (uint32_t r = r0; r < r1; ++r) {
            const RoundRec rr = p.rounds[r];
            if (live && ev < rr.count) {
                const uint32_t b = p.events[rr.first + ev];
                transform_bone(st, p, pose, inst, b);
                if (p.bones[b].bits & kBoneHasIk) solve_ik<NESTED>(st, p, pose, inst, b, lds_lane, lds_consts);
            }
            __syncthreads();
        }
        const uint32_t s0 = pass ? p.n_pre : 0, s1 = pass ? p.nb : p.n_pre;
        if (live && (p.seg_flags >> (1 + pass) & 1u)) {
            for (uint32_t s = s0 + slot; s < s1; s += kSolveSlots) {   // UpdateBoneSkinningMatrix of this list
                const uint32_t b = p.order[s];
                const BoneRec rec = p.bones[b];
                Mat4 G;
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int x = 0; x < 4; ++x) G.m[y][x] = x == y ? 1.f : 0.f;
                G.m[3][0] = rec.neg_rest[0]; G.m[3][1] = rec.neg_rest[1]; G.m[3][2] = rec.neg_rest[2];
                const Mat4 S = mul(G, st.local(b));
#pragma unroll
                for (int y = 0; y < 4; ++y) out[4 * size_t(b) + y] = make_float4(S.m[y][0], S.m[y][1], S.m[y][2], S.m[y][3]);
            }
        }
        __syncthreads();                                     // the second list's IK may rewrite these bones
    }
}

// ---- CCD-IK with SIXTEEN lanes per solve (round 4) ---------------------------------------------------------------------------------
// One solve on one lane (ccd above) is a dependent stream of ~230 k instructions: a 4x4 product is 112 of them, and after every
// link rotation the chain below it is re-placed with one product per link plus one for the target (poser_impl.inl:292-302).  A
// product's sixteen elements are independent 4-term sums, so here a solve owns 16 lanes, lane (r, c) holds element [r][c] of every
// matrix that is live, and a product is 4 quad broadcasts (row of A) + 4 broadcasts across the quads (column of B) + 4 multiplies
// + 3 adds -- the same per-element operation order (Matrix4x4::operator*, math_impl.inl:984-1003), so the same bits.  Everything
// else of a link step (directions, axis, angle, quaternions, Euler limits: poser_impl.inl:214-290) is computed by all 16 lanes
// alike from the same inputs: no exchange, identical values.  Window chains only (IkRec::fast): link j hangs off link j+1, the
// target off link 0, no append bone, no nested solve; the loop below is ccd()'s kWindow path statement for statement, with one more
// reuse of an unchanged value: the local matrix of a link BEFORE its parent product depends only on that link's own rotation and
// translation, so it is cached (lpre) and recomputed when the link is turned -- where the reference recomputes the same value.
// A block is 16 solves of ONE IK bone for 16 consecutive instances: the four solves of a wave run the same chain.
constexpr uint32_t kCoopLanes = 16, kCoopSolves = 16;
constexpr uint32_t kCoopCells = kMaxFastLinks + 2;                       // links, target, the root-most link's parent
constexpr uint32_t kCoopLpre = kCoopCells * kSerialStateFloats;          // float offsets inside a solve's LDS window
constexpr uint32_t kCoopConsts = kCoopLpre + kMaxFastLinks * 16;
constexpr uint32_t kCoopTpre = kCoopConsts + kMaxFastLinks * kLinkConstFloats;
constexpr uint32_t kCoopMisc = kCoopTpre + 16;                           // ik_pos xyz, run flag
constexpr uint32_t kCoopWindow = (kCoopMisc + 4) | 1u;                   // odd: the four solves of a wave fall into different banks

template <int K>
__device__ __forceinline__ float quad_lane(float v) {                    // the value lane K of this lane's quad holds
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float quad_of_group(float v) {                // ... lane (quad K, same position in the quad) of the 16-group
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x13 | ((K << 2) << 5)));
}
template <int L>
__device__ __forceinline__ float lane_of_group(float v) {                // ... lane L of the 16-group
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x10 | (L << 5)));
}
// element [r][c] of A x B from the elements of A and B this lane holds (lane = 4 r + c inside its group)
__device__ __forceinline__ float coop_mul(float a, float b) {
    const float a0 = quad_lane<0>(a), a1 = quad_lane<1>(a), a2 = quad_lane<2>(a), a3 = quad_lane<3>(a);
    const float b0 = quad_of_group<0>(b), b1 = quad_of_group<1>(b), b2 = quad_of_group<2>(b), b3 = quad_of_group<3>(b);
    return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
}
// this lane's element of a matrix every lane of the group holds whole
__device__ __forceinline__ float pick16(const Mat4 &m, uint32_t sub) {
    const bool b0 = sub & 1u, b1 = sub & 2u, b2 = sub & 4u, b3 = sub & 8u;
    const float r0 = b0 ? (b1 ? m.m[0][3] : m.m[0][1]) : (b1 ? m.m[0][2] : m.m[0][0]);
    const float r1 = b0 ? (b1 ? m.m[1][3] : m.m[1][1]) : (b1 ? m.m[1][2] : m.m[1][0]);
    const float r2 = b0 ? (b1 ? m.m[2][3] : m.m[2][1]) : (b1 ? m.m[2][2] : m.m[2][0]);
    const float r3 = b0 ? (b1 ? m.m[3][3] : m.m[3][1]) : (b1 ? m.m[3][2] : m.m[3][0]);
    return b3 ? (b2 ? r3 : r2) : (b2 ? r1 : r0);
}

// quat_to_euler / euler_to_quat with their independent double-precision calls spread over the lanes of a quad: the two atan2 of one
// conversion run in the even and the odd lanes, the three sincos of the other in lanes 0, 1, 2 -- the same function on the same
// argument as the one-lane versions, so the same values; every lane ends up with all of them (quad broadcasts).
__device__ __forceinline__ void quat_to_euler_coop(uint32_t order, const Quat q, float *r, uint32_t sub) {
    const float ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k;
    const float ei = q.e * q.i, ej = q.e * q.j, ek = q.e * q.k;
    const float ij = q.i * q.j, ik = q.i * q.k, jk = q.j * q.k;
    const bool zxy = order == kOrderZXY, xyz = order == kOrderXYZ;
    const float s_arg = zxy ? 2.0f * (ei + jk) : xyz ? 2.0f * (ej + ik) : 2.0f * (ek + ij);
    const float a_y = zxy ? 2.0f * (ej - ik) : 2.0f * (ei - jk);
    const float a_x = zxy ? 1 - 2.0f * (ii + jj) : xyz ? 1 - 2.0f * (ii + jj) : 1 - 2.0f * (ii + kk);
    const float b_y = (zxy || xyz) ? 2.0f * (ek - ij) : 2.0f * (ej - ik);
    const float b_x = zxy ? 1 - 2.0f * (ii + kk) : 1 - 2.0f * (jj + kk);
    const bool odd = sub & 1u;
    const float t = d_atan2(odd ? b_y : a_y, odd ? b_x : a_x);
    const float s = d_asin(s_arg), a = quad_lane<0>(t), b = quad_lane<1>(t);
    r[0] = zxy ? s : a;
    r[1] = zxy ? a : xyz ? s : b;
    r[2] = zxy ? b : xyz ? b : s;
}
__device__ __forceinline__ Quat euler_to_quat_coop(uint32_t order, const float *r, uint32_t sub) {
    const uint32_t q4 = sub & 3u;
    float sn, cs;
    d_sincos((q4 == 0 ? r[0] : q4 == 1 ? r[1] : r[2]) * 0.5f, &sn, &cs);
    const float sx = quad_lane<0>(sn), cx = quad_lane<0>(cs), sy = quad_lane<1>(sn), cy = quad_lane<1>(cs), sz = quad_lane<2>(sn),
                cz = quad_lane<2>(cs);
    const float ia = sx * cy * cz, ib = cx * sy * sz, ja = cx * sy * cz, jb = sx * cy * sz, ka = cx * cy * sz, kb = sx * sy * cz;
    Quat q;
    q.e = cx * cy * cz - sx * sy * sz;
    q.i = order == kOrderZXY ? ia - ib : ia + ib;
    q.j = order == kOrderXYZ ? ja - jb : ja + jb;
    q.k = order == kOrderYZX ? ka - kb : ka + kb;
    return q;
}

__global__ __launch_bounds__(kCoopLanes * kCoopSolves) void ik_coop_kernel(const SerialParams p, const uint32_t round) {
    extern __shared__ float coop_lds[];
    const RoundRec rr = p.rounds[round];
    const uint32_t nblk = (p.ni + kCoopSolves - 1) / kCoopSolves;
    const uint32_t ev = blockIdx.x / nblk, iblk = blockIdx.x - ev * nblk;       // which IK bone of the round, which 16 instances
    if (ev >= rr.count) return;
    const uint32_t solve = threadIdx.x / kCoopLanes, sub = threadIdx.x % kCoopLanes;
    const uint32_t inst = iblk * kCoopSolves + solve;
    if (inst >= p.ni) return;                                  // (whole 16-lane groups leave: nothing below synchronises across groups)
    const State st = {p.state + inst, p.ni};
    const float4 *pose = reinterpret_cast<const float4 *>(p.poses) + size_t(inst) * p.nb * 2;
    auto *win = (__attribute__((address_space(3))) float *)coop_lds + solve * kCoopWindow;
    const ChainState cs = {win, 0};
    const uint32_t b = p.events[rr.first + ev];
    const IkRec ik = p.iks[p.bones[b].ik];
    const LinkRec *links = p.links + ik.link0;
    const uint32_t n = ik.nlinks, tidx = n;
    const int32_t outside = ik.outside_parent;
    const WindowChain ch = {win + kCoopConsts, n, outside >= 0 ? int32_t(n + 1) : -1};
    auto group_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };

    // the event's own bone first, as the ordered kernel does (UpdateBoneTransform up to the solve), then the chain into the window
    if (sub == 0) {
        transform_bone(st, p, pose, inst, b);
        win[kCoopMisc + 0] = st.at(b, kStLocal + 12); win[kCoopMisc + 1] = st.at(b, kStLocal + 13); win[kCoopMisc + 2] = st.at(b, kStLocal + 14);
    }
    auto copy = [&](uint32_t slot, uint32_t bone, bool in) {
        for (uint32_t f = sub; f < kSerialStateFloats; f += kCoopLanes) {
            if (in) cs.at(slot, f) = st.at(bone, f); else st.at(bone, f) = cs.at(slot, f);
        }
    };
    for (uint32_t j = 0; j < n; ++j) copy(j, links[j].bone, true);
    copy(n, ik.target, true);
    if (outside >= 0) copy(n + 1, uint32_t(outside), true);
    if (sub < n) {                                             // link constants, as solve_ik lays them out
        const LinkRec lk = links[sub];
        const float *off = p.bones[lk.bone].local_offset;
        auto *c = win + kCoopConsts + sub * kLinkConstFloats;
        c[0] = off[0]; c[1] = off[1]; c[2] = off[2];
        c[3] = __uint_as_float(lk.limited | lk.order << 8 | lk.fix << 16);
        c[4] = lk.lo[0]; c[5] = lk.lo[1]; c[6] = lk.lo[2];
        c[7] = lk.hi[0]; c[8] = lk.hi[1]; c[9] = lk.hi[2];
    }
    group_sync();
    const V3 ik_pos = {win[kCoopMisc + 0], win[kCoopMisc + 1], win[kCoopMisc + 2]};
    const BoneRec trec = p.bones[ik.target];

    // ccd()'s preamble on the window, one lane: the links root-first, the target, the convergence test; then what the loop keeps
    if (sub == 0) {
        for (uint32_t i = 0; i < n; ++i) cs.set_quat(ch.idx(i), kStIkRot, q_identity());
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t j = n - i - 1, lb = links[j].bone;
            const BoneRec rec = p.bones[lb];
            transform_at(cs, rec, morph_of(p, lb, inst), pose[2 * size_t(lb)], pose[2 * size_t(lb) + 1], ch.idx(j), ch.par(j),
                         uint32_t(rec.append_parent));
        }
        transform_at(cs, trec, morph_of(p, ik.target, inst), pose[2 * size_t(ik.target)], pose[2 * size_t(ik.target) + 1], tidx, 0,
                     uint32_t(trec.append_parent));
        const V3 t0 = {cs.at(tidx, kStLocal + 12), cs.at(tidx, kStLocal + 13), cs.at(tidx, kStLocal + 14)};
        const V3 e0 = {ik_pos.x - t0.x, ik_pos.y - t0.y, ik_pos.z - t0.z};
        win[kCoopMisc + 3] = v_dot(e0, e0) < 1e-7f ? 0.f : 1.f;
        auto pre_parent = [&](uint32_t idx, const float *off, uint32_t at) {       // a bone's local matrix before its parent product
            Mat4 L = q_to_matrix(cs.quat(idx, kStTotalRot));
            L.m[3][0] = cs.at(idx, kStTotalTr + 0) + off[0];
            L.m[3][1] = cs.at(idx, kStTotalTr + 1) + off[1];
            L.m[3][2] = cs.at(idx, kStTotalTr + 2) + off[2];
#pragma unroll
            for (int k = 0; k < 16; ++k) win[at + k] = L.m[k / 4][k % 4];
        };
        for (uint32_t j = 0; j < n; ++j) {
            const V3 off = ch.offset(j);
            const float o[3] = {off.x, off.y, off.z};
            pre_parent(j, o, kCoopLpre + j * 16);
        }
        pre_parent(tidx, trec.local_offset, kCoopTpre);
    }
    group_sync();
    const bool run = win[kCoopMisc + 3] != 0.f;
    if (run) {
        const float tpre = win[kCoopTpre + sub];
        V3 tgt = {cs.at(tidx, kStLocal + 12), cs.at(tidx, kStLocal + 13), cs.at(tidx, kStLocal + 14)};
        const uint32_t ikt = ik.loop / 2;
        for (uint32_t i = 0; i < ik.loop; ++i) {
            bool changed = false;
            for (uint32_t j = 0; j < n; ++j) {
                const LinkInfo lk = ch.link(j);
                if (lk.fix == kFixAll) continue;
                const uint32_t ls = j;
                const int32_t lp = ch.par(j);
                const V3 lpos = {cs.at(ls, kStLocal + 12), cs.at(ls, kStLocal + 13), cs.at(ls, kStLocal + 14)};
                const V3 tdir = v_normalize({lpos.x - tgt.x, lpos.y - tgt.y, lpos.z - tgt.z});
                const V3 idir = v_normalize({lpos.x - ik_pos.x, lpos.y - ik_pos.y, lpos.z - ik_pos.z});
                V3 axis = {tdir.y * idir.z - tdir.z * idir.y, tdir.z * idir.x - tdir.x * idir.z,
                           tdir.x * idir.y - tdir.y * idir.x};
                if (fabsf(axis.x) < 1e-7f) axis.x = 1e-7f;
                if (fabsf(axis.y) < 1e-7f) axis.y = 1e-7f;
                if (fabsf(axis.z) < 1e-7f) axis.z = 1e-7f;
                // the parent's matrix: this lane's element for the product below, the rotation part whole for the axis
                const float loc = lp >= 0 ? cs.at(uint32_t(lp), kStLocal + sub) : (sub % 5u == 0u ? 1.f : 0.f);
                auto L = [&](uint32_t y, uint32_t x) { return lp >= 0 ? cs.at(uint32_t(lp), kStLocal + 4 * y + x) : (x == y ? 1.f : 0.f); };
                if (lk.limited && lk.fix != kFixNone && i < ikt) {
                    const uint32_t row = lk.fix - kFixX;
                    const float d = axis.x * L(row, 0) + axis.y * L(row, 1) + axis.z * L(row, 2);
                    const float sgn = d >= 0.0f ? 1.0f : -1.0f;
                    axis = {row == 0 ? sgn : 0.f, row == 1 ? sgn : 0.f, row == 2 ? sgn : 0.f};
                } else {                                       // rotate(axis, loc.Transpose()).Normalize()
                    const V3 r = {axis.x * L(0, 0) + axis.y * L(0, 1) + axis.z * L(0, 2),
                                  axis.x * L(1, 0) + axis.y * L(1, 1) + axis.z * L(1, 2),
                                  axis.x * L(2, 0) + axis.y * L(2, 1) + axis.z * L(2, 2)};
                    axis = v_normalize(r);
                }
                float dot = v_dot(tdir, idir);
                dot = dot < -1.0f ? -1.0f : dot;
                dot = 1.0f < dot ? 1.0f : dot;
                const float ac = d_acos(dot), cap = ik.angle_limit * float(j + 1);
                const float angle = cap < ac ? cap : ac;
                const Quat ikr_old = cs.quat(ls, kStIkRot);
                Quat ikr = q_mul(axis_to_quat(axis, angle), ikr_old);
                const Quat pre = cs.quat(ls, kStPreIkRot);
                if (lk.limited) {
                    Quat lr = q_mul(ikr, pre);
                    float e[3];
                    quat_to_euler_coop(lk.order, lr, e, sub);
                    limit_euler(e, lk.lo, lk.hi, i < ikt);
                    lr = euler_to_quat_coop(lk.order, e, sub);
                    ikr = q_mul(lr, q_inverse(pre));
                }
                changed = changed || __float_as_uint(ikr.i) != __float_as_uint(ikr_old.i) || __float_as_uint(ikr.j) != __float_as_uint(ikr_old.j) ||
                          __float_as_uint(ikr.k) != __float_as_uint(ikr_old.k) || __float_as_uint(ikr.e) != __float_as_uint(ikr_old.e);
                // the link just turned: ik_rotation * pre-IK rotation, as the reference; its matrix before the parent product
                const Quat total = q_mul(ikr, pre);
                Mat4 M = q_to_matrix(total);
                const V3 off = ch.offset(j);
                M.m[3][0] = cs.at(ls, kStTotalTr + 0) + off.x;
                M.m[3][1] = cs.at(ls, kStTotalTr + 1) + off.y;
                M.m[3][2] = cs.at(ls, kStTotalTr + 2) + off.z;
                const float mine = pick16(M, sub);
                group_sync();                                  // every lane has read what the stores below replace
                if (sub == 0) { cs.set_quat(ls, kStIkRot, ikr); cs.set_quat(ls, kStTotalRot, total); }
                win[kCoopLpre + ls * 16 + sub] = mine;
                float prev = lp >= 0 ? coop_mul(mine, loc) : mine;
                cs.at(ls, kStLocal + sub) = prev;
                for (uint32_t k = 1; k <= j; ++k) {            // below it nothing changed: the cached matrix IS the recomputed one
                    const uint32_t jj = j - k;
                    prev = coop_mul(win[kCoopLpre + jj * 16 + sub], prev);
                    cs.at(jj, kStLocal + sub) = prev;
                }
                const float T = coop_mul(tpre, prev);          // the target hangs off link 0, the last one placed
                cs.at(tidx, kStLocal + sub) = T;
                tgt = {lane_of_group<12>(T), lane_of_group<13>(T), lane_of_group<14>(T)};
                group_sync();                                  // the next link step reads the matrices just stored
            }
            const V3 err = {ik_pos.x - tgt.x, ik_pos.y - tgt.y, ik_pos.z - tgt.z};
            if (v_dot(err, err) < 1e-7f) break;
            if (!changed) {
                if (i >= ikt) break;
                i = ikt - 1;
            }
        }
    }
    group_sync();
    for (uint32_t j = 0; j < n; ++j) copy(j, links[j].bone, false);
    copy(n, ik.target, false);
}

// Matrix4x4<T>::Inverse(), L/util/math_impl.inl:822-897: Gauss-Jordan on [M | I] with scaled partial pivoting (a zero
// row or a zero last pivot gives the ZERO matrix), forward elimination, then the upper triangle is cleared column by
// column and every row divided by its pivot -- the reference's operations in the reference's order (f32 division is
// correctly rounded in HIP).  The reference swaps row pointers; here the two rows are exchanged by value, with every
// index a compile-time constant so that the 4x8 work matrix stays in registers.
__device__ Mat4 inverse(const Mat4 &in) {
    float s[4][8], scale[4];
    Mat4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[i][j] = in.m[i][j]; s[i][j + 4] = i == j ? 1.f : 0.f; out.m[i][j] = 0.f; }
    bool zero = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        scale[i] = fabsf(s[i][0]);
#pragma unroll
        for (int j = 1; j < 4; ++j) { const float x = fabsf(s[i][j]); if (x > scale[i]) scale[i] = x; }
        zero = zero || scale[i] == 0.f;
    }
    if (zero) return out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pivot = i;
        float best = fabsf(s[i][i] / scale[i]);
#pragma unroll
        for (int q = i + 1; q < 4; ++q) {
            const float x = fabsf(s[q][i] / scale[q]);
            if (x > best) { best = x; pivot = q; }
        }
#pragma unroll
        for (int q = i + 1; q < 4; ++q) {
            if (pivot == q) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float t = s[i][c]; s[i][c] = s[q][c]; s[q][c] = t; }
                const float t = scale[i]; scale[i] = scale[q]; scale[q] = t;
            }
        }
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const float m = s[j][i] / s[i][i];
            s[j][i] = 0.f;
#pragma unroll
            for (int jj = i + 1; jj < 8; ++jj) s[j][jj] = s[j][jj] - m * s[i][jj];
        }
    }
    if (s[3][3] == 0.f) return out;
#pragma unroll
    for (int i = 1; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const float m = s[j][i] / s[i][i];
#pragma unroll
            for (int jj = j + 1; jj < 8; ++jj) s[j][jj] = s[j][jj] - m * s[i][jj];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out.m[i][j] = s[i][j + 4] / s[i][i];
    return out;
}

// BulletPhysicsReactor::React's writes into the poser (mmd-bullet_impl.inl:312-326), one thread per instance:
// Synchronize (:34-40) for every listed bone -- the body's transform becomes the bone's skinning matrix -- then, in
// list order, Fix (:42-56) for the strict ones:
//     local = global_offset_inv * skinning;  with a parent: local = local * inverse(parent_local)
//     local.row3.xyz = total_translation + local_offset;  with a parent: local = local * parent_local
//     skinning = global_offset * local
// (parent_local is read once, as it stands when the bone's turn comes -- a parent fixed earlier in the list shows its
// new matrix, one fixed later its old one).
__global__ __launch_bounds__(kBoneMorphThreads) void physics_override_kernel(const PhysicsParams p) {
    const uint32_t inst = blockIdx.x * kBoneMorphThreads + threadIdx.x;
    if (inst >= p.ni) return;
    const State st = {p.state + inst, p.ni};
    float4 *out = reinterpret_cast<float4 *>(p.out) + size_t(inst) * p.nb * 4;
    const float4 *in = reinterpret_cast<const float4 *>(p.skinning) + size_t(inst) * p.k * 4;
    for (uint32_t k = 0; k < p.k; ++k) {
        const uint32_t b = p.bone[k];
#pragma unroll
        for (int y = 0; y < 4; ++y) out[4 * size_t(b) + y] = in[4 * size_t(k) + y];
    }
    for (uint32_t k = 0; k < p.k; ++k) {
        if (!p.strict[k]) continue;
        const uint32_t b = p.bone[k];
        const BoneRec rec = p.bones[b];
        Mat4 S, G;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const float4 r = out[4 * size_t(b) + y];
            S.m[y][0] = r.x; S.m[y][1] = r.y; S.m[y][2] = r.z; S.m[y][3] = r.w;
#pragma unroll
            for (int x = 0; x < 4; ++x) G.m[y][x] = x == y ? 1.f : 0.f;
        }
        G.m[3][0] = -rec.neg_rest[0]; G.m[3][1] = -rec.neg_rest[1]; G.m[3][2] = -rec.neg_rest[2];
        Mat4 L = mul(G, S), PL;
        if (rec.parent >= 0) {
            PL = st.local(uint32_t(rec.parent));
            L = mul(L, inverse(PL));
        }
        L.m[3][0] = st.at(b, kStTotalTr + 0) + rec.local_offset[0];
        L.m[3][1] = st.at(b, kStTotalTr + 1) + rec.local_offset[1];
        L.m[3][2] = st.at(b, kStTotalTr + 2) + rec.local_offset[2];
        if (rec.parent >= 0) L = mul(L, PL);
        st.set_local(b, L);
        G.m[3][0] = rec.neg_rest[0]; G.m[3][1] = rec.neg_rest[1]; G.m[3][2] = rec.neg_rest[2];
        S = mul(G, L);
#pragma unroll
        for (int y = 0; y < 4; ++y) out[4 * size_t(b) + y] = make_float4(S.m[y][0], S.m[y][1], S.m[y][2], S.m[y][3]);
    }
}

}  // namespace

hipError_t launch_physics_override(const PhysicsParams &p, hipStream_t stream) {
    if (p.ni == 0 || p.k == 0) return hipSuccess;
    hipLaunchKernelGGL(physics_override_kernel, dim3((p.ni + kBoneMorphThreads - 1) / kBoneMorphThreads),
                       dim3(kBoneMorphThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bone_track_eval(const BoneTrackParams &p, hipStream_t stream) {
    const size_t n = size_t(p.ni) * p.nb;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bone_track_eval_kernel, dim3(uint32_t((n + kRigThreads - 1) / kRigThreads)),
                       dim3(kRigThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_skeleton_fk(const SkeletonParams &p, hipStream_t stream) {
    const size_t n = size_t(p.ni) * p.nb;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(skeleton_fk_kernel, dim3(uint32_t((n + kRigThreads - 1) / kRigThreads)),
                       dim3(kRigThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_motion_fk(const BoneTrackParams &t, const SkeletonParams &p, hipStream_t stream) {
    if (p.ni == 0 || p.nb == 0) return hipSuccess;
    const size_t lds = size_t(p.nb) * 32;
    if (lds > kMotionFkMaxLds) return hipErrorInvalidValue;                  // callers check: the two-launch path takes over
    const uint32_t threads = std::min<uint32_t>(1024u, (p.nb + 63u) / 64u * 64u);
    hipLaunchKernelGGL(motion_fk_kernel, dim3(p.ni), dim3(threads), lds, stream, t, p);
    return hipGetLastError();
}

hipError_t launch_bone_morph(const BoneMorphParams &p, hipStream_t stream) {
    if (p.ni == 0 || p.nb == 0) return hipSuccess;
    hipLaunchKernelGGL(bone_morph_kernel, dim3((p.ni + kBoneMorphThreads - 1) / kBoneMorphThreads), dim3(kBoneMorphThreads), 0,
                       stream, p);
    return hipGetLastError();
}

static hipError_t launch_ordered_segment(const SerialParams &p, hipStream_t stream);

// The whole schedule.  Rounds that consist of window-chain IK solves only (`round_coop`, host array of p.n_rounds entries: the
// round's number of solves, 0 for every other round; nullptr: none) go to ik_coop_kernel, sixteen lanes per solve; the rounds between them to the ordered kernel, segment by segment:
// a handful of dependent launches (~2 us each) around solves that take milliseconds on one lane.  MMDX_IK_COOP=0: one launch, as before.
hipError_t launch_skeleton_ordered(const SerialParams &p0, const uint8_t *round_coop, hipStream_t stream) {
    if (p0.ni == 0 || p0.nb == 0) return hipSuccess;
    SerialParams p = p0;
    const int coop_env = env_int("MMDX_IK_COOP", 1);        // (read per call: an IK launch is milliseconds, tests flip it in one process)
    bool any = false;
    for (uint32_t r = 0; round_coop && coop_env != 0 && !p.nested && r < p.n_rounds; ++r) any = any || round_coop[r];
    if (!any) {
        p.seg_r0 = 0; p.seg_r1 = p.n_rounds; p.seg_flags = 7u;
        return launch_ordered_segment(p, stream);
    }
    bool first = true;
    auto segment = [&](uint32_t a, uint32_t b, uint32_t flags) -> hipError_t {
        if (first) flags |= 1u;
        if (a >= b && !flags) return hipSuccess;
        first = false;
        p.seg_r0 = a; p.seg_r1 = b; p.seg_flags = flags;
        return launch_ordered_segment(p, stream);
    };
    for (uint32_t pass = 0; pass < 2; ++pass) {
        if (!(p.passes >> pass & 1u)) continue;
        const uint32_t r0 = pass ? p.n_rounds_pre : 0u, r1 = pass ? p.n_rounds : p.n_rounds_pre;
        uint32_t a = r0;
        for (uint32_t r = r0; r < r1; ++r) {
            if (!round_coop[r]) continue;
            hipError_t e = segment(a, r, 0u);
            if (e != hipSuccess) return e;
            const uint32_t nblk = (p.ni + kCoopSolves - 1) / kCoopSolves;
            hipLaunchKernelGGL(ik_coop_kernel, dim3(nblk * round_coop[r]), dim3(kCoopLanes * kCoopSolves), kCoopWindow * kCoopSolves * sizeof(float),
                               stream, p, r);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            a = r + 1;
        }
        const hipError_t e = segment(a, r1, 2u << pass);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t launch_ordered_segment(const SerialParams &p, hipStream_t stream) {
    if (p.ni == 0 || p.nb == 0) return hipSuccess;
    const size_t lds = (size_t(window_floats(p.fast_slots)) * p.windows * kSolveInstances +
                        size_t(p.windows) * kMaxFastLinks * kLinkConstFloats) * sizeof(float);
    const uint32_t wgs = (p.ni + kSolveInstances - 1) / kSolveInstances;
    static const int dense_env = env_int("MMDX_SOLVE_DENSE", -1);             // A/B: 0 never, 1 whenever it fits
    bool dense = !p.nested && 2 * (lds + 1024) <= 160 * 1024;
    if (dense && dense_env != 1) {
        // CU count of the current device, asked once per device and process (two runtime calls per IK launch otherwise)
        static std::atomic<int> cu_count[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        int cus = dev >= 0 && dev < 16 ? cu_count[dev].load(std::memory_order_relaxed) : 0;
        if (cus == 0) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
                (void)hipGetLastError();
                cus = 256;
            }
            if (dev >= 0 && dev < 16) cu_count[dev].store(cus, std::memory_order_relaxed);
        }
        dense = dense_env != 0 && wgs > uint32_t(cus);
    }
    auto kernel = p.nested ? skeleton_ordered_kernel<true, false>
                           : (dense ? skeleton_ordered_kernel<false, true> : skeleton_ordered_kernel<false, false>);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(kSolveInstances * kSolveSlots), lds, stream, p);
    return hipGetLastError();
}

}  // namespace mmdx
