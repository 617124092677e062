// rig_kernels.hip -- per-frame producers of the bone palette, gfx950.
//   bone_track_eval_kernel : VMD bone tracks -> local poses, one thread per (instance, bone)
//   skeleton_fk_kernel     : local poses -> float[16] skinning palettes, one thread per (instance, bone)
// Both follow the reference's float operation order exactly (file built with -ffp-contract=off), so the
// palettes are bit-identical to libmmd's and the deform kernel downstream stays bit-exact end to end.
#include <hip/hip_runtime.h>

#include "rig.hpp"
#include "rig_kernels.hpp"

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

// Motion::GetBonePose(name, frame), L/motion/motion_impl.inl:255-319.
__global__ __launch_bounds__(kRigThreads) void bone_track_eval_kernel(const BoneTrackParams p) {
    const size_t idx = size_t(blockIdx.x) * kRigThreads + threadIdx.x;
    if (idx >= size_t(p.ni) * p.nb) return;
    const uint32_t i = uint32_t(idx / p.nb), bone = uint32_t(idx - size_t(i) * p.nb);
    const uint32_t b = p.key_off[bone], e = p.key_off[bone + 1], frame = p.frames[i];
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
    float4 *out = reinterpret_cast<float4 *>(p.out) + idx * 2;
    out[0] = t;
    out[1] = q;
}

struct Mat4 {
    float m[4][4];
};

// local_matrix_ of one bone before the parent product (Poser::UpdateBoneTransform,
// L/motion/poser_impl.inl:142-162, with no bone morph: morph_rotation_ = identity, morph_translation_ = 0).
__device__ __forceinline__ Mat4 local_matrix(const float4 t, const float4 r, const float4 off) {
    // total_rotation_ = identity * rotation_   (Quaternion::operator*, L/util/math_impl.inl:510-517)
    const float mi = 0.f, mj = 0.f, mk = 0.f, me = 1.f;
    const float i = (me * r.x + mi * r.w + mj * r.z) - mk * r.y;
    const float j = (me * r.y + mj * r.w + mk * r.x) - mi * r.z;
    const float k = (me * r.z + mi * r.y + mk * r.w) - mj * r.x;
    const float e = me * r.w - (mi * r.x + mj * r.y + mk * r.z);
    // total_translation_ = morph_translation_ + translation_
    const float tx = 0.f + t.x, ty = 0.f + t.y, tz = 0.f + t.z;
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
__global__ __launch_bounds__(kRigThreads) void skeleton_fk_kernel(const SkeletonParams p) {
    const size_t idx = size_t(blockIdx.x) * kRigThreads + threadIdx.x;
    if (idx >= size_t(p.ni) * p.nb) return;
    const uint32_t i = uint32_t(idx / p.nb), bone = uint32_t(idx - size_t(i) * p.nb);
    const float4 *pose = reinterpret_cast<const float4 *>(p.poses) + size_t(i) * p.nb * 2;
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
        const Mat4 L = local_matrix(pose[2 * size_t(b)], pose[2 * size_t(b) + 1], off[b]);
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
    float4 *out = reinterpret_cast<float4 *>(p.out) + idx * 4;
#pragma unroll
    for (int y = 0; y < 4; ++y) out[y] = make_float4(S.m[y][0], S.m[y][1], S.m[y][2], S.m[y][3]);
}

}  // namespace

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

}  // namespace mmdx
