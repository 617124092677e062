// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the deformation path.  Hand-written HIP; built
// with -ffp-contract=off so that every f32 operation is the reference's operation, in the
// reference's order (no FMA contraction): results are bit-identical to libmmd's CPU path.
//
// Replaces (reference file:line, L/ = 3rd_party/libmmd/include/mmd/):
//   morph accumulate   L/motion/poser_impl.inl:328-346, :362-365, :384-386  -> CSR gather below
//   Poser::Deform      L/motion/poser_impl.inl:396-437                      -> deform_kernel
//   Lerp / M*s / M+M   L/util/math_impl.inl:924-963, :1004-1023, :1241-1259  -> blend2 / blend4
//   transform / rotate L/util/math_impl.inl:1032-1045                        -> xform
//   32-byte repack     main.cpp:50-54, :838-859                              -> MMDX_OUT_VERTEX32
//
// Shape of the main kernel (one 256-thread workgroup = 4 wave64):
//   workgroup = one 512-vertex TILE x a GROUP of instances.  The tile's vertices are pre-sorted by
//   deform class (plan.cpp), so wavefronts are class-uniform and skip the other classes' code.
//   * the group's bone palettes are staged in LDS, only the bones the tile uses, transposed to
//     3 x float4 columns per bone (the 4th matrix column is never read by the path);
//   * static per-vertex data (position, normal, class data) is loaded ONCE into registers and reused
//     for every instance of the group -- static streams cost HBM/L2 traffic once per group;
//   * each instance's results are scattered to an LDS image of the tile's output range (undoing the
//     class sort), then written out cooperatively as 16-byte coalesced stores.
//   HBM-bound: ~24-32 B written per vertex-instance against ~100 VALU ops; no MFMA (gather of small
//   mat x vec, <= 3 flop/B).
#include "kernels.hpp"

namespace mmdx {
namespace {

constexpr int kThreads = 256;
constexpr int kVPT = int(kTileVerts) / kThreads;  // vertices per thread (2)
constexpr float kLerpLo = 1e-7f;                  // float(mmd_math_const_eps)
constexpr float kLerpHi = 0.99999988f;            // float(1.0 - mmd_math_const_eps)
constexpr float kMorphEps = 1e-7f;                // rate < 1e-7 (double) <=> rate < 1e-7f

// LDS staging images of one tile's output range (bytes; all multiples of 16)
constexpr uint32_t kSoaImgBytes = (kTileVerts * 3 + 4) * 4;        // f32 xyz + alignment slack
constexpr uint32_t kV32ImgBytes = kTileVerts * 32;
constexpr uint32_t kP16ImgBytes = (kTileVerts * 3 + 8) * 2;        // f16 xyz + alignment slack
static_assert(kSoaImgBytes % 16 == 0 && kP16ImgBytes % 16 == 0, "image alignment");

__host__ __device__ constexpr uint32_t stage_bytes(int layout) {
    return layout == MMDX_OUT_SOA ? 2 * kSoaImgBytes
                                  : (layout == MMDX_OUT_VERTEX32 ? kV32ImgBytes
                                                                 : kP16ImgBytes + kSoaImgBytes);
}

__device__ __forceinline__ float h2f(uint32_t bits16) {
    return float(__builtin_bit_cast(_Float16, (unsigned short)bits16));
}
__device__ __forceinline__ unsigned short f2h(float x) {
    return __builtin_bit_cast(unsigned short, _Float16(x));  // v_cvt_f16_f32, round to nearest even
}

// ---- matrix blend + mat*vec, reference operation order ------------------------------------------
// A palette entry in LDS is three float4 "columns": col[j] = (M[0][j], M[1][j], M[2][j], M[3][j]).
__device__ __forceinline__ float4 blend2_col(const float4 a, const float4 b, float s1, float l) {
    // (1-l)*a + l*b per element (math_impl.inl:1253 with scalar*M :1004-1023 and M+M :944-963)
    float4 r;
    r.x = s1 * a.x + l * b.x;
    r.y = s1 * a.y + l * b.y;
    r.z = s1 * a.z + l * b.z;
    r.w = s1 * a.w + l * b.w;
    return r;
}
__device__ __forceinline__ float4 blend4_col(const float4 m0, const float4 m1, const float4 m2,
                                             const float4 m3, float w0, float w1, float w2,
                                             float w3) {
    // ((m0*w0 + m1*w1) + m2*w2) + m3*w3 per element (poser_impl.inl:433; no normalisation)
    float4 r;
    r.x = ((m0.x * w0 + m1.x * w1) + m2.x * w2) + m3.x * w3;
    r.y = ((m0.y * w0 + m1.y * w1) + m2.y * w2) + m3.y * w3;
    r.z = ((m0.z * w0 + m1.z * w1) + m2.z * w2) + m3.z * w3;
    r.w = ((m0.w * w0 + m1.w * w1) + m2.w * w2) + m3.w * w3;
    return r;
}
__device__ __forceinline__ float xform_pos(const float4 c, float x, float y, float z) {
    return ((x * c.x + y * c.y) + z * c.z) + c.w;  // transform(), math_impl.inl:1039-1045
}
__device__ __forceinline__ float xform_nrm(const float4 c, float x, float y, float z) {
    return (x * c.x + y * c.y) + z * c.z;          // rotate(), math_impl.inl:1032-1038
}

// ---- cooperative copy of one LDS image to global, 16-byte stores where whole chunks fit ----------
// The image mirrors global memory from the 16-byte boundary below element `base`:
// LDS element (shift + i) <-> out[base + i], shift = base % (16/sizeof(T)).
template <typename T>
__device__ __forceinline__ void copy_out_image(const unsigned char *img, T *out, size_t base,
                                               uint32_t shift, uint32_t n, bool aligned16,
                                               int tid) {
    constexpr uint32_t EPC = 16 / sizeof(T);
    const uint32_t nchunks = (shift + n + EPC - 1) / EPC;
    T *g = out + base - shift;
    const T *l = reinterpret_cast<const T *>(img);
    for (uint32_t q = tid; q < nchunks; q += kThreads) {
        const uint32_t lo = q * EPC;
        if (aligned16 && lo >= shift && lo + EPC <= shift + n) {
            const float4 v = *reinterpret_cast<const float4 *>(img + size_t(q) * 16);
            *reinterpret_cast<float4 *>(g + lo) = v;
        } else {
#pragma unroll
            for (uint32_t e = 0; e < EPC; ++e) {
                const uint32_t i = lo + e;
                if (i >= shift && i < shift + n) g[i] = l[i];
            }
        }
    }
}

struct Slot {
    float px, py, pz, nx, ny, nz, u, v;
    float w0, w1, w2, w3;
    uint32_t b0, b1, b2, b3;  // float4 index of the bone's first column inside one instance's palette
    uint32_t perm;
    uint32_t rb, re;          // CSR row [rb, re)
    int cls;
    bool act;
};

template <bool F16>
__device__ __forceinline__ void load_entry(const void *entries, uint32_t e, float &ox, float &oy,
                                           float &oz, uint32_t &slot) {
    if constexpr (F16) {
        const uint2 r = reinterpret_cast<const uint2 *>(entries)[e];
        ox = h2f(r.x & 0xffffu);
        oy = h2f(r.x >> 16);
        oz = h2f(r.y & 0xffffu);
        slot = r.y >> 16;
    } else {
        const float4 r = reinterpret_cast<const float4 *>(entries)[e];
        ox = r.x; oy = r.y; oz = r.z;
        slot = __float_as_uint(r.w);
    }
}

// ---- the deformation kernel ----------------------------------------------------------------------
template <int LAYOUT, int MORPH, bool F16>
__global__ __launch_bounds__(kThreads) void deform_kernel(const DeformParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const TileHdr &th = p.tiles[blockIdx.x];
    const uint32_t v0 = th.v0, nvt = th.nv, n1 = th.n1, n12 = th.n1 + th.n2, nbt = th.nbt;
    const uint32_t inst0 = blockIdx.y * p.group;
    const uint32_t gcount = min(p.group, p.ni - inst0);
    float4 *pal = reinterpret_cast<float4 *>(smem);
    unsigned char *stage = smem + p.stage_off;
    constexpr uint32_t kStage = stage_bytes(LAYOUT);

    // 1. bone palettes of the group's instances -> LDS (only the tile's bones, transposed columns)
    for (uint32_t idx = tid; idx < gcount * nbt; idx += kThreads) {
        const uint32_t g = idx / nbt, lb = idx - g * nbt;
        const uint32_t bone = p.bone_list[th.bone_off + lb];
        const float4 *src =
            reinterpret_cast<const float4 *>(p.palettes + (size_t(inst0 + g) * p.nb + bone) * 16);
        const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
        float4 *dst = pal + size_t(g) * p.pal_stride + lb * 3;
        dst[0] = make_float4(r0.x, r1.x, r2.x, r3.x);
        dst[1] = make_float4(r0.y, r1.y, r2.y, r3.y);
        dst[2] = make_float4(r0.z, r1.z, r2.z, r3.z);
    }
    // 2. morph slot weights of the group -> LDS
    if constexpr (MORPH == kMorphFused1) {
        float *wl = reinterpret_cast<float *>(smem + p.w_off);
        for (uint32_t s = tid; s < p.ns; s += kThreads) wl[s] = p.wslot[size_t(inst0) * p.ns + s];
    } else if constexpr (MORPH == kMorphFused4) {
        float4 *wl4 = reinterpret_cast<float4 *>(smem + p.w_off);
        const float4 *src = reinterpret_cast<const float4 *>(p.wslot) + size_t(inst0 / 4) * p.ns;
        const uint32_t n = ((gcount + 3) / 4) * p.ns;
        for (uint32_t i = tid; i < n; i += kThreads) wl4[i] = src[i];
    }

    // 3. static per-vertex data -> registers (sorted slot s = tid + k*256)
    Slot sl[kVPT];
#pragma unroll
    for (int k = 0; k < kVPT; ++k) {
        Slot &q = sl[k];
        const uint32_t s = uint32_t(tid) + uint32_t(k) * kThreads;
        q.act = s < nvt;
        q.cls = s < n1 ? 0 : (s < n12 ? 1 : 2);
        q.px = q.py = q.pz = q.nx = q.ny = q.nz = q.u = q.v = 0.f;
        q.w0 = q.w1 = q.w2 = q.w3 = 0.f;
        q.b0 = q.b1 = q.b2 = q.b3 = 0;
        q.perm = 0; q.rb = q.re = 0;
        if (q.act) {
            const size_t gs = size_t(v0) + s;
            if constexpr (MORPH == kMorphShared) {
                q.px = p.morphed[gs * 3]; q.py = p.morphed[gs * 3 + 1]; q.pz = p.morphed[gs * 3 + 2];
            } else if constexpr (F16) {
                const uint2 r = reinterpret_cast<const uint2 *>(p.spos)[gs];
                q.px = h2f(r.x & 0xffffu); q.py = h2f(r.x >> 16); q.pz = h2f(r.y & 0xffffu);
            } else {
                const float *sp = reinterpret_cast<const float *>(p.spos) + gs * 3;
                q.px = sp[0]; q.py = sp[1]; q.pz = sp[2];
            }
            q.nx = p.snrm[gs * 3]; q.ny = p.snrm[gs * 3 + 1]; q.nz = p.snrm[gs * 3 + 2];
            if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
                const float2 uv = reinterpret_cast<const float2 *>(p.suv)[gs];
                q.u = uv.x; q.v = uv.y;
            }
            q.perm = p.perm[gs];
            if constexpr (MORPH == kMorphFused1 || MORPH == kMorphFused4) {
                q.rb = p.row_ptr[gs]; q.re = p.row_ptr[gs + 1];
            }
            if (q.cls == 0) {
                q.b0 = uint32_t(p.skin1[th.skin1_off + s]) * 3;
            } else if (q.cls == 1) {
                const uint32_t i = th.skin2_off + (s - n1);
                const uint32_t ids = p.skin2_ids[i];
                q.b0 = (ids & 0xffffu) * 3; q.b1 = (ids >> 16) * 3;
                q.w0 = p.skin2_w[i];
            } else {
                const uint32_t i = th.skin4_off + (s - n12);
                const uint2 ids = p.skin4_ids[i];
                const float4 w = p.skin4_w[i];
                q.b0 = (ids.x & 0xffffu) * 3; q.b1 = (ids.x >> 16) * 3;
                q.b2 = (ids.y & 0xffffu) * 3; q.b3 = (ids.y >> 16) * 3;
                q.w0 = w.x; q.w1 = w.y; q.w2 = w.z; q.w3 = w.w;
            }
        }
    }
    __syncthreads();

    uint32_t buf = 0;
    // one instance: skin the thread's slots, scatter to the LDS image, write the image out
    auto run_instance = [&](uint32_t g, const float (&cx)[kVPT], const float (&cy)[kVPT],
                            const float (&cz)[kVPT]) {
        const float4 *P = pal + size_t(g) * p.pal_stride;
        unsigned char *img = stage + buf * kStage;
        const size_t vbase = size_t(inst0 + g) * p.nv + v0;  // first output vertex of this tile
        const bool al = p.out_aligned != 0;
        const uint32_t sh4 = al ? uint32_t((vbase * 3) & 3) : 0u;
        const uint32_t sh8 = al ? uint32_t((vbase * 3) & 7) : 0u;
#pragma unroll
        for (int k = 0; k < kVPT; ++k) {
            const Slot &q = sl[k];
            if (!q.act) continue;
            float4 c0, c1, c2;
            if (q.cls == 0) {
                c0 = P[q.b0]; c1 = P[q.b0 + 1]; c2 = P[q.b0 + 2];
            } else if (q.cls == 1) {
                // Lerp(S[b1], S[b0])[w]  (poser_impl.inl:420-422, math_impl.inl:1246-1254)
                const float4 a0 = P[q.b1], a1 = P[q.b1 + 1], a2 = P[q.b1 + 2];
                const float4 e0 = P[q.b0], e1 = P[q.b0 + 1], e2 = P[q.b0 + 2];
                const float l = q.w0, s1 = 1.0f - l;
                c0 = blend2_col(a0, e0, s1, l);
                c1 = blend2_col(a1, e1, s1, l);
                c2 = blend2_col(a2, e2, s1, l);
                if (l < kLerpLo) { c0 = a0; c1 = a1; c2 = a2; }
                else if (l > kLerpHi) { c0 = e0; c1 = e1; c2 = e2; }
            } else {
                const float4 m00 = P[q.b0], m01 = P[q.b0 + 1], m02 = P[q.b0 + 2];
                const float4 m10 = P[q.b1], m11 = P[q.b1 + 1], m12 = P[q.b1 + 2];
                const float4 m20 = P[q.b2], m21 = P[q.b2 + 1], m22 = P[q.b2 + 2];
                const float4 m30 = P[q.b3], m31 = P[q.b3 + 1], m32 = P[q.b3 + 2];
                c0 = blend4_col(m00, m10, m20, m30, q.w0, q.w1, q.w2, q.w3);
                c1 = blend4_col(m01, m11, m21, m31, q.w0, q.w1, q.w2, q.w3);
                c2 = blend4_col(m02, m12, m22, m32, q.w0, q.w1, q.w2, q.w3);
            }
            // pos_scale is a separate multiply after the transform (main.cpp:848-850); x*1.0f == x
            const float ox = xform_pos(c0, cx[k], cy[k], cz[k]) * p.pos_scale;
            const float oy = xform_pos(c1, cx[k], cy[k], cz[k]) * p.pos_scale;
            const float oz = xform_pos(c2, cx[k], cy[k], cz[k]) * p.pos_scale;
            const float rx = xform_nrm(c0, q.nx, q.ny, q.nz);
            const float ry = xform_nrm(c1, q.nx, q.ny, q.nz);
            const float rz = xform_nrm(c2, q.nx, q.ny, q.nz);
            if constexpr (LAYOUT == MMDX_OUT_SOA) {
                float *A = reinterpret_cast<float *>(img) + sh4 + q.perm * 3;
                float *B = reinterpret_cast<float *>(img + kSoaImgBytes) + sh4 + q.perm * 3;
                A[0] = ox; A[1] = oy; A[2] = oz;
                B[0] = rx; B[1] = ry; B[2] = rz;
            } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
                float4 *I = reinterpret_cast<float4 *>(img) + q.perm * 2;
                I[0] = make_float4(ox, oy, oz, rx);
                I[1] = make_float4(ry, rz, q.u, q.v);
            } else {
                unsigned short *A = reinterpret_cast<unsigned short *>(img) + sh8 + q.perm * 3;
                float *B = reinterpret_cast<float *>(img + kP16ImgBytes) + sh4 + q.perm * 3;
                A[0] = f2h(ox); A[1] = f2h(oy); A[2] = f2h(oz);
                B[0] = rx; B[1] = ry; B[2] = rz;
            }
        }
        __syncthreads();
        if constexpr (LAYOUT == MMDX_OUT_SOA) {
            copy_out_image<float>(img, reinterpret_cast<float *>(p.out_a), vbase * 3, sh4, nvt * 3,
                                  al, tid);
            copy_out_image<float>(img + kSoaImgBytes, reinterpret_cast<float *>(p.out_b), vbase * 3,
                                  sh4, nvt * 3, al, tid);
        } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
            copy_out_image<float>(img, reinterpret_cast<float *>(p.out_a), vbase * 8, 0u, nvt * 8,
                                  al, tid);
        } else {
            copy_out_image<unsigned short>(img, reinterpret_cast<unsigned short *>(p.out_a),
                                           vbase * 3, sh8, nvt * 3, al, tid);
            copy_out_image<float>(img + kP16ImgBytes, reinterpret_cast<float *>(p.out_b), vbase * 3,
                                  sh4, nvt * 3, al, tid);
        }
        buf ^= 1u;  // double-buffered image: the next instance writes the other one, so one barrier
                    // per instance is enough
    };

    // 4. the group's instances
    if constexpr (MORPH == kMorphNone || MORPH == kMorphShared) {
        float cx[kVPT], cy[kVPT], cz[kVPT];
#pragma unroll
        for (int k = 0; k < kVPT; ++k) { cx[k] = sl[k].px; cy[k] = sl[k].py; cz[k] = sl[k].pz; }
        for (uint32_t g = 0; g < gcount; ++g) run_instance(g, cx, cy, cz);
    } else if constexpr (MORPH == kMorphFused1) {
        // vertex_image = 0; for each applied entry: image = image + offset*rate
        // (poser_impl.inl:340-346); coordinate = base + image (:407)
        const float *wl = reinterpret_cast<const float *>(smem + p.w_off);
        float cx[kVPT], cy[kVPT], cz[kVPT];
#pragma unroll
        for (int k = 0; k < kVPT; ++k) {
            float dx = 0.f, dy = 0.f, dz = 0.f;
            for (uint32_t e = sl[k].rb; e < sl[k].re; ++e) {
                float ox, oy, oz; uint32_t slot;
                load_entry<F16>(p.entries, e, ox, oy, oz, slot);
                const float w = wl[slot];
                if (!(w < kMorphEps)) { dx = dx + ox * w; dy = dy + oy * w; dz = dz + oz * w; }
            }
            cx[k] = sl[k].px + dx; cy[k] = sl[k].py + dy; cz[k] = sl[k].pz + dz;
        }
        run_instance(0, cx, cy, cz);
    } else {
        const float4 *wl4 = reinterpret_cast<const float4 *>(smem + p.w_off);
        for (uint32_t g0 = 0; g0 < gcount; g0 += 4) {
            float dx[kVPT][4], dy[kVPT][4], dz[kVPT][4];
            const float4 *wq = wl4 + size_t(g0 / 4) * p.ns;
#pragma unroll
            for (int k = 0; k < kVPT; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { dx[k][j] = 0.f; dy[k][j] = 0.f; dz[k][j] = 0.f; }
                for (uint32_t e = sl[k].rb; e < sl[k].re; ++e) {
                    float ox, oy, oz; uint32_t slot;
                    load_entry<F16>(p.entries, e, ox, oy, oz, slot);
                    const float4 w = wq[slot];
                    if (!(w.x < kMorphEps)) { dx[k][0] += ox * w.x; dy[k][0] += oy * w.x; dz[k][0] += oz * w.x; }
                    if (!(w.y < kMorphEps)) { dx[k][1] += ox * w.y; dy[k][1] += oy * w.y; dz[k][1] += oz * w.y; }
                    if (!(w.z < kMorphEps)) { dx[k][2] += ox * w.z; dy[k][2] += oy * w.z; dz[k][2] += oz * w.z; }
                    if (!(w.w < kMorphEps)) { dx[k][3] += ox * w.w; dy[k][3] += oy * w.w; dz[k][3] += oz * w.w; }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g0 + j < gcount) {
                    float cx[kVPT], cy[kVPT], cz[kVPT];
#pragma unroll
                    for (int k = 0; k < kVPT; ++k) {
                        cx[k] = sl[k].px + dx[k][j]; cy[k] = sl[k].py + dy[k][j]; cz[k] = sl[k].pz + dz[k][j];
                    }
                    run_instance(g0 + j, cx, cy, cz);
                }
            }
        }
    }
}

// ---- shared morph pass: morphed[gs] = base[gs] + sum(offset*rate), once per call ------------------
template <bool F16>
__global__ __launch_bounds__(kThreads) void morph_apply_kernel(const DeformParams p) {
    const size_t gs = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (gs >= p.nv) return;
    float bx, by, bz;
    if constexpr (F16) {
        const uint2 r = reinterpret_cast<const uint2 *>(p.spos)[gs];
        bx = h2f(r.x & 0xffffu); by = h2f(r.x >> 16); bz = h2f(r.y & 0xffffu);
    } else {
        const float *sp = reinterpret_cast<const float *>(p.spos) + gs * 3;
        bx = sp[0]; by = sp[1]; bz = sp[2];
    }
    float dx = 0.f, dy = 0.f, dz = 0.f;
    const uint32_t rb = p.row_ptr[gs], re = p.row_ptr[gs + 1];
    for (uint32_t e = rb; e < re; ++e) {
        float ox, oy, oz; uint32_t slot;
        load_entry<F16>(p.entries, e, ox, oy, oz, slot);
        const float w = p.wslot[slot];
        if (!(w < kMorphEps)) { dx = dx + ox * w; dy = dy + oy * w; dz = dz + oz * w; }
    }
    p.morphed[gs * 3] = bx + dx;
    p.morphed[gs * 3 + 1] = by + dy;
    p.morphed[gs * 3 + 2] = bz + dz;
}

// ---- group-morph flattening on the device (UpdateMorphTransform's recursion, per slot) ------------
__global__ __launch_bounds__(kThreads) void flatten_kernel(const FlattenParams f) {
    const size_t idx = size_t(blockIdx.x) * kThreads + threadIdx.x;
    const uint32_t rows = f.quad ? ((f.niw + 3) / 4) * 4 : f.niw;
    if (idx >= size_t(rows) * f.ns) return;
    const uint32_t i = uint32_t(idx / f.ns), s = uint32_t(idx - size_t(i) * f.ns);
    float out = 0.f;
    if (i < f.niw) {
        float r = f.rates[size_t(i) * f.nm + f.slot_top[s]];
        bool skip = r < kMorphEps;
        for (uint32_t c = f.chain_off[s]; !skip && c < f.chain_off[s + 1]; ++c) {
            r = f.chain_rate[c] * r;
            skip = r < kMorphEps;
        }
        out = skip ? 0.f : r;
    }
    if (f.quad) f.out[(size_t(i / 4) * f.ns + s) * 4 + (i & 3)] = out;
    else f.out[idx] = out;
}

// ---- streaming copy / fill: the practical HBM ceiling printed next to the roofline ---------------
__global__ __launch_bounds__(kThreads) void copy_kernel(float4 *dst, const float4 *src, size_t n) {
    for (size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += size_t(gridDim.x) * kThreads)
        dst[i] = src[i];
}
__global__ __launch_bounds__(kThreads) void fill_kernel(float4 *dst, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += size_t(gridDim.x) * kThreads)
        dst[i] = v;
}

using KernelFn = void (*)(const DeformParams);

template <int LAYOUT, bool F16>
KernelFn pick_morph(int morph) {
    switch (morph) {
    case kMorphNone: return deform_kernel<LAYOUT, kMorphNone, F16>;
    case kMorphShared: return deform_kernel<LAYOUT, kMorphShared, F16>;
    case kMorphFused1: return deform_kernel<LAYOUT, kMorphFused1, F16>;
    default: return deform_kernel<LAYOUT, kMorphFused4, F16>;
    }
}

KernelFn pick(int layout, int morph, bool f16) {
    if (f16) return layout == MMDX_OUT_SOA_POS16 ? pick_morph<MMDX_OUT_SOA_POS16, true>(morph) : nullptr;
    if (layout == MMDX_OUT_SOA) return pick_morph<MMDX_OUT_SOA, false>(morph);
    if (layout == MMDX_OUT_VERTEX32) return pick_morph<MMDX_OUT_VERTEX32, false>(morph);
    return nullptr;
}

}  // namespace

size_t deform_lds_bytes(int layout, int morph, uint32_t group, uint32_t max_tile_bones, uint32_t ns,
                        uint32_t *stage_off, uint32_t *w_off) {
    size_t off = size_t(group) * max_tile_bones * 48;
    *stage_off = uint32_t(off);
    off += 2 * size_t(stage_bytes(layout));
    *w_off = uint32_t(off);
    if (morph == kMorphFused1) off += (size_t(ns) * 4 + 15) / 16 * 16;
    else if (morph == kMorphFused4) off += size_t((group + 3) / 4) * ns * 16;
    return off;
}

hipError_t prepare_kernels() {
    for (int f16 = 0; f16 < 2; ++f16)
        for (int layout = 0; layout < 3; ++layout)
            for (int morph = 0; morph < 4; ++morph) {
                KernelFn fn = pick(layout, morph, f16 != 0);
                if (!fn) continue;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_deform(int layout, int morph, bool f16, const DeformParams &p, uint32_t ntiles,
                         size_t lds_bytes, hipStream_t stream) {
    KernelFn fn = pick(layout, morph, f16);
    if (!fn) return hipErrorInvalidValue;
    const dim3 grid(ntiles, (p.ni + p.group - 1) / p.group);
    hipLaunchKernelGGL(fn, grid, dim3(kThreads), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_morph_apply(bool f16, const DeformParams &p, hipStream_t stream) {
    const dim3 grid((p.nv + kThreads - 1) / kThreads);
    if (f16) hipLaunchKernelGGL(morph_apply_kernel<true>, grid, dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL(morph_apply_kernel<false>, grid, dim3(kThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_flatten(const FlattenParams &f, hipStream_t stream) {
    const uint32_t rows = f.quad ? ((f.niw + 3) / 4) * 4 : f.niw;
    const size_t n = size_t(rows) * f.ns;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(flatten_kernel, dim3(uint32_t((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, stream, f);
    return hipGetLastError();
}

hipError_t launch_copy(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(kThreads), 0, stream,
                       reinterpret_cast<float4 *>(dst), reinterpret_cast<const float4 *>(src),
                       bytes / 16);
    return hipGetLastError();
}

hipError_t launch_fill(void *dst, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(256 * 8), dim3(kThreads), 0, stream,
                       reinterpret_cast<float4 *>(dst), bytes / 16);
    return hipGetLastError();
}

}  // namespace mmdx
