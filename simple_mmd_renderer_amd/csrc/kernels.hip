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

#include <type_traits>
#ifdef PK_STAMPS
#include <algorithm>
#include <cstdio>
#include <vector>
#endif

// MMDX_FAST_MATH (kernels_fast.hip includes this file with it defined): the SAME kernels with multiply-add contraction allowed
// (v_fma_f32 / v_pk_fma_f32) for models created with MMDX_CREATE_FAST_MATH -- results within a stated tolerance of the
// reference's instead of bit-identical (include/mmdx.h).  Only the deform / frame / shared-morph kernels exist in that build,
// under *_fast launch names; everything else in this file is compiled once, uncontracted.
#ifdef MMDX_FAST_MATH
#pragma clang fp contract(fast)
#define MMDX_K(name) name##_fast
#else
#define MMDX_K(name) name
#endif

// Timing-diagnostic build knobs (PK_SKIP_WALK, PK_SKIP_SKIN, FUSED4_SKIP_WALK: a kernel without one of its halves -- WRONG results by
// design; PK_STAMPS: in-kernel cycle stamps): only together with MMDX_DIAGNOSTIC_BUILD, and never the product -- the build's flags
// are part of the source stamp (build.py source_sha), so bench.py / smoke() refuse such a library unless it is loaded explicitly
// through MMDX_LIB by a tool.
#if (defined(PK_SKIP_WALK) || defined(PK_SKIP_SKIN) || defined(FUSED4_SKIP_WALK) || defined(PK_STAMPS)) && !defined(MMDX_DIAGNOSTIC_BUILD)
#error "PK_SKIP_* / FUSED4_SKIP_WALK / PK_STAMPS are timing diagnostics (wrong results): add MMDX_DIAGNOSTIC_BUILD to MMDX_BUILD_DEFS"
#endif

namespace mmdx {
namespace {

constexpr int kThreads = 256;
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

// ---- cache policy of the OUTPUT stores ------------------------------------------------------------------------------------------
// 98.7 % of the crowd kernel's HBM bytes are write-once stores.  MMDX_STORE_POLICY (build-time A/B knob): 0 plain (the line stays
// in the XCD's L2 until evicted), 1 `nt` (non-temporal hint).  Shipped: 1 -- measured interleaved on the same arrays
// (profiles/r03/store_policy_*.txt): nt takes the crowd kernel from 218.7 to 206.2 us (fast placement; 264 -> 253 on a slow one),
// the 32-byte-vertex crowd from 271 to 246, config 3' from 382 to 341, config 5 x 64 from 144 to 136.  Both are compiler-visible
// stores, bit-identical results.  The write-through flavour (`sc1 nt`) exists ONLY as the buffer-store builtin of CopyFast below:
// round 3's inline-asm `sc1` flavours (policies 2-5) were invisible to the hazard recogniser and corrupted results -- removed.
#ifndef MMDX_STORE_POLICY
#define MMDX_STORE_POLICY 1
#endif
#if MMDX_STORE_POLICY != 0 && MMDX_STORE_POLICY != 1
#error "MMDX_STORE_POLICY: 0 (plain) or 1 (nt); the inline-asm sc0/sc1 flavours of round 3 corrupted data and no longer exist"
#endif
#ifndef MMDX_WALK_ZPAIR
#define MMDX_WALK_ZPAIR 1          // build-time A/B knob of the per-instance-morph walk (see there): z-pairing on (round 3: -2 % config 5 x 64, -1.5 % config 2 x 64, config 3' even; profiles/r03/walk_zpair_ab.txt)
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float4 *dst, const float4 v) {
#if MMDX_STORE_POLICY == 1
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(dst));
#else
    *dst = v;
#endif
}
// three consecutive floats (tile-order direct stores: 12-byte vertices, so no 16-byte alignment to promise)
__device__ __forceinline__ void store3(float *dst, float x, float y, float z) {
#if MMDX_STORE_POLICY == 1
    __builtin_nontemporal_store(x, dst); __builtin_nontemporal_store(y, dst + 1); __builtin_nontemporal_store(z, dst + 2);
#else
    dst[0] = x; dst[1] = y; dst[2] = z;
#endif
}
template <typename T>
__device__ __forceinline__ void store_elem(T *dst, const T v) {      // scalar pieces (ragged tiles, direct stores)
#if MMDX_STORE_POLICY == 1
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;           // (scalar sc1 stores are one fabric write each -- the guide prices a dword at 6x a dwordx4 per byte: never)
#endif
}

// ---- matrix blend + mat*vec, reference operation order, on PACKED f32 pairs ------------------------
// A wave64 issues one VALU instruction per 4 cycles, so at the occupancy this kernel runs at the
// number of VALU instructions is what counts: v_pk_mul_f32 / v_pk_add_f32 do two IEEE f32 operations
// per instruction (each half rounded exactly like v_mul_f32 / v_add_f32).  The palette is laid out
// in LDS so that every operand pair is an aligned register pair -- no shuffles:
//     entry = 3 x float4 = {m00 m01 | m10 m11} {m20 m21 | m30 m31} {m02 m12 | m22 m32}
// (M[r][c], row-vector convention; column 3 of the matrix is never read by the path).
typedef float v2f __attribute__((ext_vector_type(2)));

struct M12 {
    v2f p0, p1, p2, p3;  // (m_r0, m_r1) for r = 0..3
    v2f q0, q1;          // (m02, m12), (m22, m32)
};

__device__ __forceinline__ M12 load_m12(const float4 *P, uint32_t b) {
    const float4 A = P[b], B = P[b + 1], C = P[b + 2];
    M12 m;
    m.p0 = v2f{A.x, A.y}; m.p1 = v2f{A.z, A.w};
    m.p2 = v2f{B.x, B.y}; m.p3 = v2f{B.z, B.w};
    m.q0 = v2f{C.x, C.y}; m.q1 = v2f{C.z, C.w};
    return m;
}
// Lerp(a, b)[l] interior: (1-l)*a + l*b per element (math_impl.inl:1253; s*M :1004-1023, M+M :944-963)
__device__ __forceinline__ M12 blend2(const M12 &a, const M12 &b, float s1, float l) {
    M12 r;
    r.p0 = a.p0 * s1 + b.p0 * l; r.p1 = a.p1 * s1 + b.p1 * l;
    r.p2 = a.p2 * s1 + b.p2 * l; r.p3 = a.p3 * s1 + b.p3 * l;
    r.q0 = a.q0 * s1 + b.q0 * l; r.q1 = a.q1 * s1 + b.q1 * l;
    return r;
}
// BDEF4: ((m0*w0 + m1*w1) + m2*w2) + m3*w3 per element (poser_impl.inl:433; weights not normalised), evaluated in steps:
// a*wa + b*wb, then t + c*wc
__device__ __forceinline__ M12 mul_add(const M12 &a, float wa, const M12 &b, float wb) {
    M12 r;
    r.p0 = a.p0 * wa + b.p0 * wb; r.p1 = a.p1 * wa + b.p1 * wb;
    r.p2 = a.p2 * wa + b.p2 * wb; r.p3 = a.p3 * wa + b.p3 * wb;
    r.q0 = a.q0 * wa + b.q0 * wb; r.q1 = a.q1 * wa + b.q1 * wb;
    return r;
}
__device__ __forceinline__ M12 add_mul(const M12 &t, const M12 &c, float wc) {
    M12 r;
    r.p0 = t.p0 + c.p0 * wc; r.p1 = t.p1 + c.p1 * wc;
    r.p2 = t.p2 + c.p2 * wc; r.p3 = t.p3 + c.p3 * wc;
    r.q0 = t.q0 + c.q0 * wc; r.q1 = t.q1 + c.q1 * wc;
    return r;
}
// transform(): out[j] = ((x*m0j + y*m1j) + z*m2j) + m3j   (math_impl.inl:1039-1045)
__device__ __forceinline__ void xform_pos(const M12 &m, v2f xy, float z, v2f &oxy, float &oz) {
    oxy = ((m.p0 * xy.x + m.p1 * xy.y) + m.p2 * z) + m.p3;
    const v2f t = m.q0 * xy;
    oz = ((t.x + t.y) + z * m.q1.x) + m.q1.y;
}
// rotate(): out[j] = (x*m0j + y*m1j) + z*m2j   (math_impl.inl:1032-1038) -- same matrix, no
// inverse transpose, no renormalisation
__device__ __forceinline__ void xform_nrm(const M12 &m, v2f xy, float z, v2f &oxy, float &oz) {
    oxy = (m.p0 * xy.x + m.p1 * xy.y) + m.p2 * z;
    const v2f t = m.q0 * xy;
    oz = (t.x + t.y) + z * m.q1.x;
}

// ---- cooperative copy of LDS images to global, 16-byte stores where whole chunks fit -------------
// An image mirrors global memory from the 16-byte boundary below element `base`:
// LDS element (shift + i) <-> out[base + i], shift = base % (16/sizeof(T)).
template <typename T>
__device__ __forceinline__ void copy_chunk(const unsigned char *img, T *g, uint32_t q, uint32_t shift,
                                           uint32_t n, bool aligned16) {
    constexpr uint32_t EPC = 16 / sizeof(T);
    const uint32_t lo = q * EPC;
    if (aligned16 && lo >= shift && lo + EPC <= shift + n) {
        const float4 v = *reinterpret_cast<const float4 *>(img + size_t(q) * 16);
        store16(reinterpret_cast<float4 *>(g + lo), v);
    } else {
        const T *l = reinterpret_cast<const T *>(img);
#pragma unroll
        for (uint32_t e = 0; e < EPC; ++e) {
            const uint32_t i = lo + e;
            if (i >= shift && i < shift + n) store_elem(g + i, l[i]);
        }
    }
}

// two images (A then B) in one pass over the workgroup's threads
template <int THREADS, typename TA, typename TB>
__device__ __forceinline__ void copy_out2(const unsigned char *imgA, TA *outA, size_t baseA,
                                          uint32_t shA, uint32_t nA, const unsigned char *imgB,
                                          TB *outB, size_t baseB, uint32_t shB, uint32_t nB,
                                          bool aligned16, int tid) {
    const uint32_t ncA = (shA + nA + 16 / sizeof(TA) - 1) / (16 / sizeof(TA));
    const uint32_t ncB = (shB + nB + 16 / sizeof(TB) - 1) / (16 / sizeof(TB));
    TA *gA = outA + baseA - shA;
    TB *gB = outB + baseB - shB;
    for (uint32_t q = tid; q < ncA + ncB; q += THREADS) {
        if (q < ncA) copy_chunk<TA>(imgA, gA, q, shA, nA, aligned16);
        else copy_chunk<TB>(imgB, gB, q - ncA, shB, nB, aligned16);
    }
}


// Fast path for a FULL tile whose output range starts on a 16-byte boundary (every tile but the last
// when NV % 4 == 0): all LDS reads are issued first, then all stores -- one LDS round trip per
// instance instead of one per chunk, no per-chunk branching.  CA / CB = 16-byte chunks of image A / B;
// image B sits `gapB` bytes after image A in LDS.
// WT (write-through): the 16-byte stores carry `sc1 nt` instead of `nt` -- the line is written through to memory and dropped from
// the XCD's L2.  On output arrays whose physical backing is in the slow store mode (six plain allocations in
// seven) the SoA crowd kernel runs 4.6-5.0 % faster with it (251.6 -> 240.0 us), in the fast mode 2 % slower (207.0 -> 211.4):
// profiles/r03/store_policy_buffer_builtin_ab*.txt -- so the host picks per launch (api.cpp).  Issued through the buffer-store
// builtin: the compiler sees the instruction and keeps its data hazards (round 3's inline-asm flavours did not, and corrupted
// results: removed); descriptor = this instance's piece of the array (wave-uniform), lane offset in bytes.
template <int THREADS, int CA, int CB, int I, bool WT = false>
struct CopyFast {
    // step I of ceil((CA+CB)/THREADS): load chunk q = tid + I*THREADS, recurse (so every LDS read is
    // issued before the first store), then store it.  Scalars only -- an indexed float4 array here
    // ends up in scratch memory with a vmcnt(0) in front of every store.
    static __device__ __forceinline__ void run(const unsigned char *img, uint32_t gapB, float4 *outA,
                                               float4 *outB, int tid) {
        constexpr int TOTAL = CA + CB;
        if constexpr (I * THREADS < TOTAL) {
            constexpr bool full = (I + 1) * THREADS <= TOTAL;
            const int q = tid + I * THREADS;
            const bool inA = (I + 1) * THREADS <= CA ? true : (I * THREADS >= CA ? false : q < CA);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (full || q < TOTAL)
                v = *reinterpret_cast<const float4 *>(img + (inA ? 0u : gapB - uint32_t(CA) * 16u) +
                                                      size_t(q) * 16);
            CopyFast<THREADS, CA, CB, I + 1, WT>::run(img, gapB, outA, outB, tid);
            if (full || q < TOTAL) {
                if constexpr (WT) {
                    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                    constexpr int kSc1Nt = 2 | 16;                      // cache-policy bits of the buffer intrinsics: nt, sc1
                    const v4u d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    if (inA) {
                        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(outA, 0, uint32_t(CA) * 16u, 0x00027000);
                        __builtin_amdgcn_raw_buffer_store_b128(d, r, uint32_t(q) * 16u, 0, kSc1Nt);
                    } else {
                        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(outB, 0, uint32_t(CB) * 16u, 0x00027000);
                        __builtin_amdgcn_raw_buffer_store_b128(d, r, uint32_t(q - CA) * 16u, 0, kSc1Nt);
                    }
                } else {
                    float4 *dst = inA ? outA + q : outB + (q - CA);
                    store16(dst, v);
                }
            }
        }
    }
};
template <int THREADS, int CA, int CB, bool WT = false>
__device__ __forceinline__ void copy_out_fast(const unsigned char *img, uint32_t gapB, float4 *outA,
                                              float4 *outB, int tid) {
    CopyFast<THREADS, CA, CB, 0, WT>::run(img, gapB, outA, outB, tid);
}

struct Slot {
    v2f pxy, nxy, uv;
    float pz, nz;
    float w0, w1, w2, w3;
    uint32_t b0, b1, b2, b3;  // float4 index of the bone's entry inside one instance's palette
    uint32_t perm;
    uint32_t rb, rlen;        // morph-table row: first entry (incl. lane), padded length
    int cls;
    bool act;
};

// One row of the sliced-ELL morph table (plan.cpp): entry j of this lane's row sits at
// base + j*64 (base already includes the lane), `len` is the slice's padded row length and is
// wave-uniform.  Four independent, fully coalesced loads are issued before the first entry is
// consumed; the body still sees the entries one by one, in order, so the rounding sequence is the
// reference's.  Padding entries carry the dummy slot NS whose weight is always 0.
template <bool F16>
struct RawEntry { using type = float4; };
template <>
struct RawEntry<true> { using type = uint2; };

constexpr uint32_t kRowAhead = 4;   // entries in flight per lane in the deform kernels; deeper (8) costs the fused kernels a wave per
                                    // SIMD and loses: measured.  The stand-alone morph pass has the registers for 16.

template <bool F16, uint32_t B = kRowAhead>
struct RowHead {                    // the first B entries of a row, loaded ahead of their use
    typename RawEntry<F16>::type e[B];
};
// Entry j of a row sits at byte (base + j*64) * sizeof(entry) of the table: a wave-uniform 64-bit base plus a 32-bit
// per-lane offset (plan.cpp keeps the table under 4 GiB), so a lane carries one address register, not a pointer pair.
template <bool F16>
__device__ __forceinline__ typename RawEntry<F16>::type row_entry(const void *entries, uint32_t base, uint32_t j) {
    using Raw = typename RawEntry<F16>::type;
    const uint32_t off = (base + j * 64u) * uint32_t(sizeof(Raw));
    return *reinterpret_cast<const Raw *>(static_cast<const unsigned char *>(entries) + off);
}
template <bool F16, uint32_t B = kRowAhead>
__device__ __forceinline__ void row_prefetch(const void *entries, uint32_t base, uint32_t len, RowHead<F16, B> &h) {
    using Raw = typename RawEntry<F16>::type;
    const uint32_t last = len ? len - 1 : 0u;
    // every element is (re)defined on every path, so that a head requested for the next pack is not kept alive
    // across the code in front of the request
#pragma unroll
    for (uint32_t i = 0; i < B; ++i) {
        Raw v = Raw{};
        if (len) v = row_entry<F16>(entries, base, min(i, last));
        h.e[i] = v;
    }
}
template <bool F16, uint32_t B = kRowAhead, typename Body>
__device__ __forceinline__ void row_consume(const void *entries, uint32_t base, uint32_t len, const RowHead<F16, B> &h,
                                            Body body) {
    using Raw = typename RawEntry<F16>::type;
    auto apply = [&](const Raw r) {
        if constexpr (F16) {
            body(h2f(r.x & 0xffffu), h2f(r.x >> 16), h2f(r.y & 0xffffu), uint32_t(r.y >> 16));
        } else {
            body(r.x, r.y, r.z, __float_as_uint(r.w));
        }
    };
    // software-pipelined: the next B entries are in flight while the current B are consumed (the
    // gather is a chain of L2 round trips; with ~3 resident waves per SIMD nothing else hides them)
    if (len == 0) return;
    const uint32_t last = len - 1;
    Raw cur[B], nxt[B];
#pragma unroll
    for (uint32_t i = 0; i < B; ++i) cur[i] = h.e[i];
    for (uint32_t j = 0; j < len; j += B) {
#pragma unroll
        for (uint32_t i = 0; i < B; ++i) nxt[i] = cur[i];
        if (j + B < len) {
#pragma unroll
            for (uint32_t i = 0; i < B; ++i) nxt[i] = row_entry<F16>(entries, base, min(j + B + i, last));
        }
#pragma unroll
        for (uint32_t i = 0; i < B; ++i)
            if (i == 0 || j + i < len) apply(cur[i]);
#pragma unroll
        for (uint32_t i = 0; i < B; ++i) cur[i] = nxt[i];
    }
}
template <bool F16, uint32_t B = kRowAhead, typename Body>
__device__ __forceinline__ void for_row(const void *entries, uint32_t base, uint32_t len, Body body) {
    RowHead<F16, B> h;
    row_prefetch<F16, B>(entries, base, len, h);
    row_consume<F16, B>(entries, base, len, h, body);
}

// The same walk through a ROLLING window (round 4): entry j+D is requested as soon as entry j has been consumed -- D loads in
// flight in ONE register set (row_consume keeps a current and a next batch, 2 x 4 entries, and copies one into the other), and the
// steady-state loop has no branch in its body (a refill past the row's end re-reads the last entry, which is never applied
// twice), so the compiler counts the loads in flight and waits for exactly the oldest one instead of draining the queue at
// every batch.  `len` must be wave-uniform (a scalar: the slice's padded length).
#ifndef MMDX_WALK_DEPTH32
#define MMDX_WALK_DEPTH32 4        // entries in flight per lane, f32 table (16-byte entries: 4 registers each)
#endif
#ifndef MMDX_WALK_DEPTH16
#define MMDX_WALK_DEPTH16 4        // ... f16 table (8-byte entries: 2 registers each)
#endif
template <bool F16>
struct WalkHead {                   // the first D entries of a row, requested ahead of the walk that consumes them
    static constexpr uint32_t D = F16 ? MMDX_WALK_DEPTH16 : MMDX_WALK_DEPTH32;
    typename RawEntry<F16>::type e[D];
};
template <bool F16>
__device__ __forceinline__ void walk_head(const void *entries, uint32_t base, uint32_t len, WalkHead<F16> &h) {
    using Raw = typename RawEntry<F16>::type;
    const uint32_t last = len ? len - 1 : 0u;
#pragma unroll
    for (uint32_t i = 0; i < WalkHead<F16>::D; ++i) h.e[i] = len ? row_entry<F16>(entries, base, min(i, last)) : Raw{};
}
template <bool F16, typename Body>
__device__ __forceinline__ void walk_from_head(const void *entries, uint32_t base, uint32_t len, WalkHead<F16> &h, Body body) {
    using Raw = typename RawEntry<F16>::type;
    constexpr uint32_t D = WalkHead<F16>::D;
    if (len == 0) return;
    auto apply = [&](const Raw r) {
        if constexpr (F16) body(h2f(r.x & 0xffffu), h2f(r.x >> 16), h2f(r.y & 0xffffu), uint32_t(r.y >> 16));
        else body(r.x, r.y, r.z, __float_as_uint(r.w));
    };
    const uint32_t last = len - 1;
    uint32_t j = 0;
    for (; j + D < len; j += D) {
#pragma unroll
        for (uint32_t i = 0; i < D; ++i) {
            apply(h.e[i]);
            h.e[i] = row_entry<F16>(entries, base, min(j + D + i, last));
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < D; ++i)
        if (i == 0 || j + i < len) apply(h.e[i]);
}
template <bool F16, typename Body>
__device__ __forceinline__ void walk_rolling(const void *entries, uint32_t base, uint32_t len, Body body) {
    WalkHead<F16> h;
    walk_head<F16>(entries, base, len, h);
    walk_from_head<F16>(entries, base, len, h, body);
}

// Group-morph recursion of one slot (UpdateMorphTransform, poser_impl.inl:328-339): rate[top] times the
// nested groups' sub-rates, with the `< 1e-7` skip after every factor; a skipped slot weighs 0.
__device__ __forceinline__ float slot_weight(const float *rates, const uint32_t *slot_top,
                                             const uint32_t *chain_off, const float *chain_rate,
                                             uint32_t s) {
    float r = rates[slot_top[s]];
    bool skip = r < kMorphEps;
    for (uint32_t c = chain_off[s]; !skip && c < chain_off[s + 1]; ++c) {
        r = chain_rate[c] * r;
        skip = r < kMorphEps;
    }
    return skip ? 0.f : r;
}

// ---- pieces shared by the deform kernels ------------------------------------------------------------
// XCD-aware work mapping (speed only, never correctness): workgroup ids are dealt round-robin over
// the 8 XCDs, each with a private 4 MiB L2.  XCD x gets a CONTIGUOUS range of tiles for all
// instance groups, tile index fastest, so the ~96 workgroups resident on one XCD are (its ~12 tiles)
// x (8 groups): a tile's static streams and the palette rows neighbouring tiles share are fetched
// into that L2 once instead of once per XCD.  Tiles left over by ntiles % 8 are split by groups.
// Returns false for a padding workgroup.
// (Round 3 tried walking an XCD's tiles in CHUNKS -- few tiles x all their instance groups resident together, so that the
// per-instance-morph kernels fetch a tile's morph-table slice into L2 once per launch instead of once per group: config 5 x 64
// and config 2 x 64 unchanged within noise, config 3' 7 % slower at 1-2 tiles per chunk; profiles/r03/xcd_chunk_sweep.txt.)
__device__ __forceinline__ bool map_workgroup(const DeformParams &p, uint32_t &tile, uint32_t &grp) {
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const uint32_t T = p.ntiles >> 3, main_count = T * p.ngroups;
    if (k < main_count) {
        grp = k / T;
        tile = xcd * T + (k - grp * T);
        return true;
    }
    const uint32_t r = xcd * p.rem_per_xcd + (k - main_count);
    if (r >= (p.ntiles & 7u) * p.ngroups) return false;
    const uint32_t rt = r / p.ngroups;
    tile = 8u * T + rt;
    grp = r - rt * p.ngroups;
    return true;
}

// static per-vertex data of sorted slot s of tile `th` -> registers
template <int LAYOUT, int MORPH, bool F16>
__device__ __forceinline__ void load_slot(const DeformParams &p, const TileHdr &th, uint32_t s, Slot &q) {
    const uint32_t n1 = th.n1, n12 = th.n1 + th.n2;
    q.act = s < th.nv;
    q.cls = s < n1 ? 0 : (s < n12 ? 1 : 2);
    q.pxy = q.nxy = q.uv = v2f{0.f, 0.f};
    q.pz = q.nz = 0.f;
    q.w0 = q.w1 = q.w2 = q.w3 = 0.f;
    q.b0 = q.b1 = q.b2 = q.b3 = 0;
    q.perm = 0; q.rb = q.rlen = 0;
    if (!q.act) return;
    const size_t gs = size_t(th.v0) + s;
    if constexpr (MORPH == kMorphShared) {
        q.pxy = v2f{p.morphed[gs * 3], p.morphed[gs * 3 + 1]}; q.pz = p.morphed[gs * 3 + 2];
    } else if constexpr (F16) {
        const uint2 r = reinterpret_cast<const uint2 *>(p.spos)[gs];
        q.pxy = v2f{h2f(r.x & 0xffffu), h2f(r.x >> 16)}; q.pz = h2f(r.y & 0xffffu);
    } else {
        const float *sp = reinterpret_cast<const float *>(p.spos) + gs * 3;
        q.pxy = v2f{sp[0], sp[1]}; q.pz = sp[2];
    }
    q.nxy = v2f{p.snrm[gs * 3], p.snrm[gs * 3 + 1]}; q.nz = p.snrm[gs * 3 + 2];
    if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
        const float2 uv = reinterpret_cast<const float2 *>(p.suv)[gs];
        q.uv = v2f{uv.x, uv.y};
    }
    q.perm = p.perm[gs];
    if constexpr (MORPH == kMorphFused1 || MORPH == kMorphFused4) {
        const uint2 sl2 = p.ell[gs >> 6];            // slice of this wave-slot (wave-uniform)
        q.rb = sl2.x + uint32_t(gs & 63); q.rlen = sl2.y;
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

// bone palettes of `count` instances (first one `inst0`, then every `istep`-th) -> LDS: only the tile's bones,
// in the pair layout load_m12 reads.  One wave-instruction fetches ONE bone for 16 instances (lane = instance x
// matrix row: sixteen 64-byte pieces); the bone id is wave-uniform, so it comes through the scalar cache and the
// palette loads do not wait for a vector load of the bone list first -- one dependent round trip less in a
// set-up phase that runs while the CU's memory pipeline is full of other workgroups' stores.
// `qstep` != 0: the group is made of instance QUADS qstep quads apart (pack_kernel's interleaved mapping): group instance g is
// instance inst0 + (g >> 2) * 4 * qstep + (g & 3); instances past the crowd's end are left out.
template <int THREADS>
__device__ __forceinline__ void stage_palettes(const DeformParams &p, const TileHdr &th, float4 *pal, uint32_t inst0,
                                               uint32_t istep, uint32_t count, int tid, uint32_t qstep = 0u) {
    const uint32_t nbt = th.nbt;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(uint32_t(tid) >> 6), lane = uint32_t(tid) & 63u;
    const uint32_t gl = lane >> 2, r = lane & 3u;
    const uint32_t *bones = p.bone_list + th.bone_off;
    for (uint32_t g0 = 0; g0 < count; g0 += 16) {
        const uint32_t g = g0 + gl;
        uint32_t inst = inst0 + (g < count ? g : 0u) * istep;
        if (qstep) inst = inst0 + (g >> 2) * 4u * qstep + (g & 3u);
        const bool on = g < count && inst < p.ni;
        const float *src = p.palettes + size_t(on ? inst : inst0) * p.nb * 16 + r * 4;
        float *dst = reinterpret_cast<float *>(pal + size_t(g) * p.pal_stride);
#pragma unroll 4
        for (uint32_t lb = wave; lb < nbt; lb += THREADS / 64) {
            const uint32_t bone = bones[lb];                 // wave-uniform: scalar load
            if (on) {
                const float4 row = *reinterpret_cast<const float4 *>(src + size_t(bone) * 16);
                // entry = {m00 m01 | m10 m11} {m20 m21 | m30 m31} {m02 m12 | m22 m32}; this lane holds row r
                float *e = dst + lb * 12;
                *reinterpret_cast<float2 *>(e + (r >> 1) * 4 + (r & 1u) * 2) = make_float2(row.x, row.y);
                e[8 + r] = row.z;
            }
        }
    }
}

// Set-up of the one-set-of-rates kernel (kMorphFused1: a single-model frame, small crowds with a shared facial state), ordered
// for LATENCY: such a launch is a handful of workgroups whose run time is the length of their chain of dependent memory round
// trips (config 2, one frame: 9.6 MB in ~8 us).  Everything that depends on the tile header only is requested first (static
// vertex data, bone ids of the palette rows, slot tables), then everything that depends on those (the first BH entries of the
// threads' morph rows, the palette rows, the morph rates), and only then does anything wait: two round trips instead of the six
// or seven of "palettes, then weights, then vertices, then rows".  One batch of kPalBatch palette rows / kWgtBatch slots per
// thread is handled this way; what does not fit (large groups, thousands of slots) follows in plain loops.
constexpr int kPalBatch = 4, kWgtBatch = 4;

template <int THREADS, int LAYOUT, int MORPH, bool F16, int VPT, uint32_t BH>
__device__ __forceinline__ void setup_fused1(const DeformParams &p, const TileHdr &th, float4 *pal, float *wl, uint32_t inst0,
                                             uint32_t istep, uint32_t gcount, int tid, uint32_t slot_base, Slot (&sl)[VPT],
                                             RowHead<F16, BH> (&hd)[VPT]) {
    constexpr bool kWalk = MORPH == kMorphFused1;         // kMorphNone: palettes and static data only
    const uint32_t nb4 = th.nbt * 4u, rows = gcount * nb4, ns = p.ns;
    const uint32_t *bones = p.bone_list + th.bone_off;
    const bool flat = p.fused_rates == nullptr;          // slot weights already evaluated (p.wslot)
    // palette row e = (instance g of the group, tile bone lb, matrix row r): e = (g * nbt + lb) * 4 + r
    auto row_src = [&](uint32_t e, uint32_t bone) {
        const uint32_t g = gcount == 1 ? 0u : e / nb4;
        return reinterpret_cast<const float4 *>(p.palettes + size_t(inst0 + g * istep) * p.nb * 16 + size_t(bone) * 16 + (e & 3u) * 4);
    };
    auto row_put = [&](uint32_t e, const float4 row) {
        const uint32_t g = gcount == 1 ? 0u : e / nb4, rem = e - g * nb4, lb = rem >> 2, r = rem & 3u;
        // entry = {m00 m01 | m10 m11} {m20 m21 | m30 m31} {m02 m12 | m22 m32}; this element is row r
        float *en = reinterpret_cast<float *>(pal + size_t(g) * p.pal_stride) + lb * 12;
        *reinterpret_cast<float2 *>(en + (r >> 1) * 4 + (r & 1u) * 2) = make_float2(row.x, row.y);
        en[8 + r] = row.z;
    };
    auto chain = [&](float r, uint32_t c0, uint32_t c1) {     // slot_weight() from the rate of the slot's top-level morph on
        bool skip = r < kMorphEps;
        for (uint32_t c = c0; !skip && c < c1; ++c) {
            r = p.chain_rate[c] * r;
            skip = r < kMorphEps;
        }
        return skip ? 0.f : r;
    };

    // ---- round trip 1: static vertex data, bone ids, slot tables ---------------------------------------------------------
#pragma unroll
    for (int k = 0; k < VPT; ++k) load_slot<LAYOUT, MORPH, F16>(p, th, slot_base + uint32_t(tid) + uint32_t(k) * THREADS, sl[k]);
    uint32_t bone[kPalBatch];
#pragma unroll
    for (int i = 0; i < kPalBatch; ++i) {
        const uint32_t e = uint32_t(tid) + uint32_t(i) * THREADS;
        bone[i] = e < rows ? bones[((gcount == 1 ? e : e % nb4)) >> 2] : 0u;
    }
    uint32_t top[kWgtBatch], c0[kWgtBatch], c1[kWgtBatch];
    float wv[kWgtBatch];
#pragma unroll
    for (int i = 0; i < kWgtBatch; ++i) {
        const uint32_t sidx = uint32_t(tid) + uint32_t(i) * THREADS;
        top[i] = c0[i] = c1[i] = 0u;
        wv[i] = 0.f;
        if (kWalk && sidx < ns) {
            if (flat) wv[i] = p.wslot[sidx];
            else { top[i] = p.slot_top[sidx]; c0[i] = p.chain_off[sidx]; c1[i] = p.chain_off[sidx + 1]; }
        }
    }
    // ---- round trip 2: the heads of the morph rows, the palette rows, the rates ---------------------------------------------
    if constexpr (kWalk) {
#pragma unroll
        for (int k = 0; k < VPT; ++k) row_prefetch<F16, BH>(p.entries, sl[k].rb, sl[k].rlen, hd[k]);
    }
    float4 row[kPalBatch];
#pragma unroll
    for (int i = 0; i < kPalBatch; ++i) {
        const uint32_t e = uint32_t(tid) + uint32_t(i) * THREADS;
        row[i] = e < rows ? *row_src(e, bone[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (kWalk && !flat) {
#pragma unroll
        for (int i = 0; i < kWgtBatch; ++i)
            if (uint32_t(tid) + uint32_t(i) * THREADS < ns) wv[i] = p.fused_rates[top[i]];
    }
    // ---- into LDS -----------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < kPalBatch; ++i) {
        const uint32_t e = uint32_t(tid) + uint32_t(i) * THREADS;
        if (e < rows) row_put(e, row[i]);
    }
    if constexpr (kWalk) {
#pragma unroll
        for (int i = 0; i < kWgtBatch; ++i) {
            const uint32_t sidx = uint32_t(tid) + uint32_t(i) * THREADS;
            if (sidx <= ns) wl[sidx] = sidx < ns ? (flat ? wv[i] : chain(wv[i], c0[i], c1[i])) : 0.f;   // slot ns = table padding, weight 0
        }
    }
    // ---- what did not fit the first batch -------------------------------------------------------------------------------------
    for (uint32_t e = uint32_t(kPalBatch) * THREADS + uint32_t(tid); e < rows; e += THREADS)
        row_put(e, *row_src(e, bones[(e % nb4) >> 2]));
    if constexpr (kWalk) {
        for (uint32_t sidx = uint32_t(kWgtBatch) * THREADS + uint32_t(tid); sidx <= ns; sidx += THREADS)
            wl[sidx] = sidx < ns ? (flat ? p.wslot[sidx] : slot_weight(p.fused_rates, p.slot_top, p.chain_off, p.chain_rate, sidx)) : 0.f;
    }
}

// The vertex's blended palette matrix: BDEF1 the bone's, BDEF2-like Lerp(S[b1], S[b0])[w] with the epsilon short circuits,
// BDEF4 ((S0*w0 + S1*w1) + S2*w2) + S3*w3 (poser_impl.inl:412-434).  P = the instance's palette in LDS.
__device__ __forceinline__ M12 skin_matrix(const Slot &q, const float4 *P) {
    M12 m;
    if (q.cls == 0) {
        m = load_m12(P, q.b0);
    } else if (q.cls == 1) {
        // Lerp(S[b1], S[b0])[w]  (poser_impl.inl:420-422, math_impl.inl:1246-1254)
        const M12 a = load_m12(P, q.b1), e = load_m12(P, q.b0);
        const float l = q.w0;
        m = blend2(a, e, 1.0f - l, l);
        // epsilon short-circuits: rare, so only waves that hold such a weight pay for them
        const bool lo = l < kLerpLo, hi = l > kLerpHi;
        if (__builtin_amdgcn_ballot_w64(lo || hi) != 0) {
            if (lo) m = a;
            else if (hi) m = e;
        }
    } else {
        // ((S0*w0 + S1*w1) + S2*w2) + S3*w3, two palette entries in registers at a time: the scheduler would
        // otherwise issue all twelve LDS reads first and hold four matrices (48 VGPRs) at once -- the peak
        // of the kernel's register pressure
        {
            const M12 a = load_m12(P, q.b0), b = load_m12(P, q.b1);
            m = mul_add(a, q.w0, b, q.w1);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const M12 c = load_m12(P, q.b2);
            m = add_mul(m, c, q.w2);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const M12 d = load_m12(P, q.b3);
            m = add_mul(m, d, q.w3);
        }
    }
    return m;
}

// One instance: skin the thread's slots with the palette at P (LDS), scatter the results to the LDS image `img`
// (undoing the class sort), ONE workgroup barrier, then write the image out with coalesced 16-byte stores.
// `inst` = the instance's index in the output arrays, cxy / cz = the (morphed) positions of the thread's slots.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_barrier`: called between the barrier and the copy-out -- whatever it requests from memory is in the queue IN FRONT of this
// instance's stores (vmcnt is in order: a load issued behind them would wait for their drain).
template <int THREADS, int LAYOUT, int VPT, bool TILE, bool ALL_FAST, bool WT = false, typename Hook = NoHook>
__device__ __forceinline__ void skin_instance(const DeformParams &p, const Slot (&sl)[VPT], const float4 *P,
                                              unsigned char *img, uint32_t inst, uint32_t v0, uint32_t nvt,
                                              const v2f (&cxy)[VPT], const float (&cz)[VPT], int tid, Hook after_barrier = Hook()) {
    const size_t vbase = size_t(inst) * p.nv + v0;  // first output vertex of this tile
    const bool al = p.out_aligned != 0;
    const uint32_t sh4 = al ? uint32_t((vbase * 3) & 3) : 0u;
    const uint32_t sh8 = al ? uint32_t((vbase * 3) & 7) : 0u;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const Slot &q = sl[k];
        if (!q.act) continue;
        const M12 m = skin_matrix(q, P);
        v2f oxy, rxy;
        float oz, rz;
        xform_pos(m, cxy[k], cz[k], oxy, oz);
        xform_nrm(m, q.nxy, q.nz, rxy, rz);
        // pos_scale is a separate multiply after the transform (main.cpp:848-850); x*1.0f == x
        oxy = oxy * p.pos_scale;
        oz = oz * p.pos_scale;
        if constexpr (TILE) {
            // MMDX_CREATE_TILE_ORDER (a compile-time variant: the default kernels' code is untouched by it): the vertex keeps its sorted slot in the output -- consecutive lanes write consecutive
            // vertices (12 / 32 / 6 bytes apart) straight from registers: no image, no barrier, no wave waits for another
            const size_t v = vbase + uint32_t(tid) + uint32_t(k) * THREADS;
            if constexpr (LAYOUT == MMDX_OUT_SOA) {
                float *A = reinterpret_cast<float *>(p.out_a) + v * 3, *B = reinterpret_cast<float *>(p.out_b) + v * 3;
                store3(A, oxy.x, oxy.y, oz);
                store3(B, rxy.x, rxy.y, rz);
            } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
                float *A = reinterpret_cast<float *>(p.out_a) + v * 8;
                if (al) {
                    store16(reinterpret_cast<float4 *>(A), make_float4(oxy.x, oxy.y, oz, rxy.x));
                    store16(reinterpret_cast<float4 *>(A) + 1, make_float4(rxy.y, rz, q.uv.x, q.uv.y));
                } else {
                    A[0] = oxy.x; A[1] = oxy.y; A[2] = oz; A[3] = rxy.x; A[4] = rxy.y; A[5] = rz; A[6] = q.uv.x; A[7] = q.uv.y;
                }
            } else {
                unsigned short *A = reinterpret_cast<unsigned short *>(p.out_a) + v * 3;
                float *B = reinterpret_cast<float *>(p.out_b) + v * 3;
                A[0] = f2h(oxy.x); A[1] = f2h(oxy.y); A[2] = f2h(oz);
                B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
            }
            continue;
        }
        if constexpr (LAYOUT == MMDX_OUT_SOA) {
            float *A = reinterpret_cast<float *>(img) + sh4 + q.perm * 3;
            float *B = reinterpret_cast<float *>(img + kSoaImgBytes) + sh4 + q.perm * 3;
            A[0] = oxy.x; A[1] = oxy.y; A[2] = oz;
            B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
        } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
            float4 *I = reinterpret_cast<float4 *>(img) + q.perm * 2;
            I[0] = make_float4(oxy.x, oxy.y, oz, rxy.x);
            I[1] = make_float4(rxy.y, rz, q.uv.x, q.uv.y);
        } else {
            unsigned short *A = reinterpret_cast<unsigned short *>(img) + sh8 + q.perm * 3;
            float *B = reinterpret_cast<float *>(img + kP16ImgBytes) + sh4 + q.perm * 3;
            A[0] = f2h(oxy.x); A[1] = f2h(oxy.y); A[2] = f2h(oz);
            B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
        }
    }
    if constexpr (TILE) return;
    __syncthreads();
    after_barrier();
    // ALL_FAST: the kernel found, once per workgroup, that every instance of this full tile starts on a 16-byte boundary; the
    // generic copy-out is then not even compiled into the instance loop (1-2 % of the crowd step: measured)
    const bool fast = ALL_FAST || (al && nvt == kTileVerts && sh4 == 0 && sh8 == 0);
    if constexpr (LAYOUT == MMDX_OUT_SOA) {
        float *oa = reinterpret_cast<float *>(p.out_a), *ob = reinterpret_cast<float *>(p.out_b);
        if (fast)
            copy_out_fast<THREADS, kTileVerts * 12 / 16, kTileVerts * 12 / 16, WT>(
                img, kSoaImgBytes, reinterpret_cast<float4 *>(oa + vbase * 3),
                reinterpret_cast<float4 *>(ob + vbase * 3), tid);
        else
            copy_out2<THREADS, float, float>(img, oa, vbase * 3, sh4, nvt * 3, img + kSoaImgBytes, ob,
                                             vbase * 3, sh4, nvt * 3, al, tid);
    } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
        float *oa = reinterpret_cast<float *>(p.out_a);
        if (fast)
            copy_out_fast<THREADS, kTileVerts * 32 / 16, 0>(
                img, 0u, reinterpret_cast<float4 *>(oa + vbase * 8), nullptr, tid);
        else
            copy_out2<THREADS, float, float>(img, oa, vbase * 8, 0u, nvt * 8, img, oa, 0, 0u, 0u, al,
                                             tid);
    } else {
        unsigned short *oa = reinterpret_cast<unsigned short *>(p.out_a);
        float *ob = reinterpret_cast<float *>(p.out_b);
        if (fast)
            copy_out_fast<THREADS, kTileVerts * 6 / 16, kTileVerts * 12 / 16>(
                img, kP16ImgBytes, reinterpret_cast<float4 *>(oa + vbase * 3),
                reinterpret_cast<float4 *>(ob + vbase * 3), tid);
        else
            copy_out2<THREADS, unsigned short, float>(img, oa, vbase * 3, sh8, nvt * 3,
                                                      img + kP16ImgBytes, ob, vbase * 3, sh4, nvt * 3,
                                                      al, tid);
    }
}

// Workgroups of one CU that run the same program start together and stay in step: all in their walk (vector-memory loads), then
// all in their stores.  First-round workgroups are therefore started out of phase: residency slot k of a CU (blocks are dealt
// round-robin over the 8 XCDs, then over an XCD's 32 CUs: observed, speed only) sleeps k * stagger * 64 cycles first.  Later
// workgroups inherit the offset of the one whose place they take.
__device__ __forceinline__ void stagger_start(const DeformParams &p) {
    if (p.stagger == 0u || p.slots_per_cu < 2u) return;
    const uint32_t round = blockIdx.x >> 8;                 // 8 XCDs x 32 CUs
    if (round >= p.slots_per_cu) return;
    for (uint32_t k = 0; k < round * p.stagger; k += 100) __builtin_amdgcn_s_sleep(100);
}

// The walk of the per-instance-morph mode: MMDX_WALK_ROLLING=1 (build-time A/B knob) takes the rolling window; a sorted slot's row
// length is the same for the 64 lanes of its wave (one slice), lanes past the tile's end included or all of them 0.
#ifndef MMDX_WALK_ROLLING
#define MMDX_WALK_ROLLING 1
#endif
#ifndef MMDX_WALK_PREFETCH
#define MMDX_WALK_PREFETCH 0       // the next pack's row head requested in front of a pack's last stores: 144 VGPRs instead of 118 (f32), i.e.
#endif                             // one workgroup per CU instead of two, or 44 spills under a 128-register bound whose reloads sit between the
                                   // instances' stores -- off.  (It could hide the first round trip only: every later request of the walk still
                                   // queues behind those stores.  pack_kernel has the registers for it and loses all the same.)
template <bool F16, bool ROLLING, typename Body>
__device__ __forceinline__ void fused4_walk(const void *entries, uint32_t rb, uint32_t rlen, Body body) {
    if constexpr (ROLLING && MMDX_WALK_ROLLING != 0) walk_rolling<F16>(entries, rb, __builtin_amdgcn_readfirstlane(rlen), body);
    else for_row<F16>(entries, rb, rlen, body);
}

// ---- the deformation kernel ----------------------------------------------------------------------
// THREADS = 512: one sorted slot per lane, 8 waves per workgroup; THREADS = 256: two slots per lane.
template <int THREADS, int LAYOUT, int MORPH, bool F16, bool TILE, bool WT = false>
__global__ __launch_bounds__(THREADS) void deform_kernel(const DeformParams p) {
    constexpr int VPT = int(kTileVerts) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    uint32_t tile, grp;
    if (!map_workgroup(p, tile, grp)) return;       // padding workgroup
    if constexpr (MORPH == kMorphFused4) stagger_start(p);
    const TileHdr &th = p.tiles[tile];
    const uint32_t v0 = th.v0, nvt = th.nv;
    // Instances of this workgroup.  Blocked: grp*group + g.  Interleaved (crowd modes): g*ngroups + grp,
    // so that the workgroups running at the same time (neighbouring grp) write NEIGHBOURING instances:
    // chip-wide the stores then sweep a few contiguous megabytes of each output array, like a linear
    // fill, instead of 8+ streams 9.6 MB apart.
    const bool ilv = (MORPH == kMorphNone || MORPH == kMorphShared || MORPH == kMorphFused1) && p.interleave != 0;
    const uint32_t inst0 = ilv ? grp : grp * p.group;
    const uint32_t istep = ilv ? p.ngroups : 1u;
    const uint32_t gcount = ilv ? (p.ni > grp ? min(p.group, (p.ni - grp + p.ngroups - 1) / p.ngroups) : 0u)
                                : min(p.group, p.ni - inst0);
    float4 *pal = reinterpret_cast<float4 *>(smem);
    unsigned char *stage = smem + p.stage_off;
    constexpr uint32_t kStage = stage_bytes(LAYOUT);

    Slot sl[VPT];
    // kMorphFused1: entries of a morph row in flight per lane (a single frame is latency-bound and has the registers)
    constexpr uint32_t BH = MORPH == kMorphFused1 ? (F16 ? 16u : 8u) : kRowAhead;
    RowHead<F16, BH> hd[VPT];
    // kMorphFused4: slot weights of one PACK of instances (kQuads quads of four) at a time; the first pack's are
    // requested here, under the set-up's other loads
    constexpr int kQuads = VPT == 1 ? 2 : 1, kPack = 4 * kQuads;
    const uint32_t wstride = p.ns + 1, wcount = uint32_t(kQuads) * wstride;
    // float4 i of the weights of the pack that starts at group instance g0 (a missing second quad reads as zeros:
    // weight +0 adds nothing, bit for bit; its instances are never written out)
    auto pack_weight = [&](uint32_t g0, uint32_t i) {
        const float4 *src = reinterpret_cast<const float4 *>(p.wslot) + size_t((inst0 + g0) / 4) * wstride;
        return (i < wstride || g0 + 4 < gcount) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if constexpr (MORPH == kMorphFused1) {
        // palettes, slot weights (ONE set of morph rates for the whole launch; group morphs flattened in here when the raw
        // rates are passed), static vertex data and the heads of the morph rows, in latency order
        setup_fused1<THREADS, LAYOUT, kMorphFused1, F16, VPT, BH>(p, th, pal, reinterpret_cast<float *>(smem + p.w_off), inst0, istep,
                                                                   gcount, tid, 0u, sl, hd);
    } else {
        // 1. bone palettes of the group's instances -> LDS: only the tile's bones, in the pair layout
        stage_palettes<THREADS>(p, th, pal, inst0, istep, gcount, tid);
        // 2. morph slot weights of the first pack -> LDS
        if constexpr (MORPH == kMorphFused4) {
            float4 *wq0 = reinterpret_cast<float4 *>(smem + p.w_off);
            for (uint32_t i = tid; i < wcount && gcount; i += THREADS) wq0[i] = pack_weight(0, i);
        }
        // 3. static per-vertex data -> registers (sorted slot s = tid + k*THREADS)
#pragma unroll
        for (int k = 0; k < VPT; ++k) load_slot<LAYOUT, MORPH, F16>(p, th, uint32_t(tid) + uint32_t(k) * THREADS, sl[k]);
    }
    __syncthreads();

    uint32_t buf = 0;
    // Every output piece of this workgroup starts on a 16-byte boundary and the tile is full (all but the last tile of a model
    // whose vertex count is a multiple of 4 -- 8 for the f16 layout): decided once, the instance loop below is instantiated twice.
    constexpr uint32_t kAlignVerts = LAYOUT == MMDX_OUT_SOA_POS16 ? 8u : (LAYOUT == MMDX_OUT_SOA ? 4u : 1u);
    const bool all_fast = !TILE && p.out_aligned != 0 && nvt == kTileVerts && p.nv % kAlignVerts == 0 && v0 % kAlignVerts == 0;
    auto instances = [&](auto all_fast_tag) {
    constexpr bool kAllFast = decltype(all_fast_tag)::value;
    // the image is double buffered: the next instance writes the other one, so one barrier per instance is enough
    auto run_instance = [&](uint32_t g, const v2f (&cxy)[VPT], const float (&cz)[VPT], auto hook) {
        skin_instance<THREADS, LAYOUT, VPT, TILE, kAllFast, WT>(p, sl, pal + size_t(g) * p.pal_stride, stage + buf * kStage,
                                                           inst0 + g * istep, v0, nvt, cxy, cz, tid, hook);
        buf ^= 1u;
    };

    // 4. the group's instances
    if constexpr (MORPH == kMorphNone || MORPH == kMorphShared) {
        v2f cxy[VPT];
        float cz[VPT];
#pragma unroll
        for (int k = 0; k < VPT; ++k) { cxy[k] = sl[k].pxy; cz[k] = sl[k].pz; }
        for (uint32_t g = 0; g < gcount; ++g) run_instance(g, cxy, cz, NoHook{});
    } else if constexpr (MORPH == kMorphFused1) {
        // vertex_image = 0; for each applied entry: image = image + offset*rate
        // (poser_impl.inl:340-346); coordinate = base + image (:407)
        const float *wl = reinterpret_cast<const float *>(smem + p.w_off);
        v2f cxy[VPT];
        float cz[VPT];
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            v2f dxy = v2f{0.f, 0.f};
            float dz = 0.f;
            row_consume<F16, BH>(p.entries, sl[k].rb, sl[k].rlen, hd[k], [&](float ox, float oy, float oz, uint32_t slot) {
                const float w = wl[slot];
                if (!(w < kMorphEps)) { dxy = dxy + v2f{ox, oy} * w; dz = dz + oz * w; }
            });
            cxy[k] = sl[k].pxy + dxy; cz[k] = sl[k].pz + dz;
            // the morphed positions are kept (sorted order) for later calls that declare the rates unchanged; the record of
            // which rates `morphed` belongs to (morph_apply_kernel) no longer holds
            if (k == 0 && grp == 0 && tile == 0 && tid == 0 && p.morphed && p.morph_seen) p.morph_seen[kSeenValid] = 0u;
            if (grp == 0 && p.morphed && sl[k].act) {
                float *mo = p.morphed + (size_t(v0) + uint32_t(tid) + uint32_t(k) * THREADS) * 3;
                mo[0] = cxy[k].x; mo[1] = cxy[k].y; mo[2] = cz[k];
            }
        }
        // the group's instances: one for a single-model frame; for a crowd with a shared facial state every workgroup
        // repeats its tile's short walk (the table stays in its XCD's L2) instead of waiting for a separate morph pass
        for (uint32_t g = 0; g < gcount; ++g) run_instance(g, cxy, cz, NoHook{});
    } else {
        // Slot weights of kQuads instance quads live in LDS at a time (kQuads x (NS+1) x float4): one pass over
        // a vertex's morph row then serves 4*kQuads instances, so the table is walked (and its L2 latency paid)
        // that much less often.  One slot per lane (512 threads) has the registers for two quads; two slots per
        // lane do not (a third wave per SIMD is worth more there: measured).  The accumulators pair x with y of ONE instance; pairing two
        // instances per component instead (three packed multiply-adds per entry and instance pair, a quarter fewer
        // instructions) measured 10 % slower: v_pk_mul_f32 / v_pk_add_f32 occupy the SIMD for two passes, so the
        // arithmetic time is the same and the position fix-up comes on top.
        float4 *wq = reinterpret_cast<float4 *>(smem + p.w_off);
        // The first pack's weights were staged during the set-up.  When one float4 per thread covers a pack's weights,
        // the NEXT pack's are fetched into a register before this pack's instances are skinned and reach LDS after
        // them: their load round trip is hidden and one barrier separates the packs.  (Otherwise: staged between the
        // packs.)
        const bool wreg = wcount <= uint32_t(THREADS);
        // One slot per lane: the head of the row (the same row for every pack) is requested ahead of the walk -- by the set-up for
        // the first pack, then IN FRONT of each pack's last stores for the next one, so that its round trip runs beside their
        // drain instead of behind it (MMDX_WALK_PREFETCH, build-time A/B knob).
        constexpr bool kHead = VPT == 1 && !TILE && MMDX_WALK_ROLLING != 0 && MMDX_WALK_PREFETCH != 0;
        WalkHead<F16> head;
        const uint32_t hlen = __builtin_amdgcn_readfirstlane(sl[0].rlen);
        if constexpr (kHead) walk_head<F16>(p.entries, sl[0].rb, hlen, head);
        for (uint32_t g0 = 0; g0 < gcount;) {       // (the pack loop continues only out of the branch that prefetched `head`)
            v2f dxy[VPT][kPack];
            float dz[VPT][kPack];
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
#pragma unroll
                for (int j = 0; j < kPack; ++j) { dxy[k][j] = v2f{0.f, 0.f}; dz[k][j] = 0.f; }
                auto weights = [&](uint32_t slot, float (&w)[kPack]) {
                    const float4 a4 = wq[slot];
                    w[0] = a4.x; w[1] = a4.y; w[2] = a4.z; w[3] = a4.w;
                    if constexpr (kQuads == 2) {
                        const float4 b4 = wq[wstride + slot];
                        w[4] = b4.x; w[5] = b4.y; w[6] = b4.z; w[7] = b4.w;
                    }
                };
                auto walk = [&](auto body) {
                    if constexpr (kHead) walk_from_head<F16>(p.entries, sl[k].rb, hlen, head, body);
                    else fused4_walk<F16, !TILE>(p.entries, sl[k].rb, sl[k].rlen, body);
                };
#ifdef FUSED4_SKIP_WALK       // timing diagnostic only (wrong results): this kernel without its walk
                if (false) {
#else
                if (p.finite_offsets) {
#endif
                    // A skipped slot carries w = +0 exactly (flatten_kernel) and a running sum that
                    // started at +0 can never be -0, so with FINITE offsets "image + offset*0" leaves
                    // the image bit-for-bit unchanged: the skip needs no branch.
                    walk([&](float ox, float oy, float oz, uint32_t slot) {
                        float w[kPack];
                        weights(slot, w);
                        const v2f oxy = v2f{ox, oy};
#if MMDX_WALK_ZPAIR
                        // z of two instances per packed instruction: (dz_j, dz_j+1) += (oz, oz) * (w_j, w_j+1) -- the weights of
                        // an instance quad arrive as such pairs, and each half is the same IEEE mul / add as before
                        const v2f ozz = v2f{oz, oz};
#pragma unroll
                        for (int j = 0; j < kPack; j += 2) {
                            dxy[k][j] += oxy * w[j]; dxy[k][j + 1] += oxy * w[j + 1];
                            const v2f zz = v2f{dz[k][j], dz[k][j + 1]} + ozz * v2f{w[j], w[j + 1]};
                            dz[k][j] = zz.x; dz[k][j + 1] = zz.y;
                        }
#else
#pragma unroll
                        for (int j = 0; j < kPack; ++j) { dxy[k][j] += oxy * w[j]; dz[k][j] += oz * w[j]; }
#endif
                    });
                } else {
#ifndef FUSED4_SKIP_WALK
                    walk([&](float ox, float oy, float oz, uint32_t slot) {
                        float w[kPack];
                        weights(slot, w);
                        const v2f oxy = v2f{ox, oy};
#pragma unroll
                        for (int j = 0; j < kPack; ++j)
                            if (!(w[j] < kMorphEps)) { dxy[k][j] += oxy * w[j]; dz[k][j] += oz * w[j]; }
                    });
#endif
                }
            }
            const bool more = g0 + kPack < gcount;
            float4 wnext = make_float4(0.f, 0.f, 0.f, 0.f);
            if (more && wreg && uint32_t(tid) < wcount) wnext = pack_weight(g0 + kPack, uint32_t(tid));
            auto one = [&](int j, auto hook) {
                v2f cxy[VPT];
                float cz[VPT];
#pragma unroll
                for (int k = 0; k < VPT; ++k) {
                    cxy[k] = sl[k].pxy + dxy[k][j]; cz[k] = sl[k].pz + dz[k][j];
                }
                run_instance(g0 + j, cxy, cz, hook);
            };
#pragma unroll
            for (int j = 0; j + 1 < kPack; ++j)
                if (g0 + j < gcount) one(j, NoHook{});
            if (!more) {
                if (g0 + kPack - 1 < gcount) one(kPack - 1, NoHook{});
                break;
            }
            if constexpr (kHead) one(kPack - 1, [&]() {
                uint32_t rb = sl[0].rb;
                asm volatile("" : "+v"(rb));          // (pins the requests behind the barrier: hoisted into the skinning they cost it 16 registers)
                walk_head<F16>(p.entries, rb, hlen, head);
            });
            else one(kPack - 1, NoHook{});
            {
                // every wave has left this pack's walk (it passed the barriers of the pack's instances): the weights can
                // be replaced; one barrier before the next walk reads them.  Tile-order outputs have no per-instance barrier
                // (skin_instance returns before it), so there a wave with short morph rows could get here while another is
                // still reading this pack's weights: one barrier per pack closes the walk explicitly.
                if constexpr (TILE) __syncthreads();
                if (wreg) {
                    if (uint32_t(tid) < wcount) wq[tid] = wnext;
                } else {
                    for (uint32_t i = tid; i < wcount; i += THREADS) wq[i] = pack_weight(g0 + kPack, i);
                }
                __syncthreads();
            }
            g0 += kPack;
        }
    }
    };   // instances
    // (the one-set-of-rates kernel keeps ONE instantiation: two cost it 30 VGPRs -- 98 -> 130 -- and with them its second
    // workgroup per CU: config 5, one frame, 14.9 -> 19.7 us)
    if constexpr (MORPH == kMorphFused1) {
        instances(std::false_type{});
    } else {
        if (all_fast) instances(std::true_type{});
        else instances(std::false_type{});
    }
}

// ---- per-instance morph weights, second shape (round 4): packs of 4 instances, walk and skinning in SEPARATE phases -------------
// deform_kernel<512, ., kMorphFused4> holds, at the same time, the walk's accumulators for 8 instances, the entries in flight, the
// vertex's skin data and the blend's matrices: 126 VGPRs, two 8-wave workgroups per CU, and its counters say it waits (VALU issue
// 59 %, LDS 50 % of the cycles, waves parked 53 % of their life: profiles/r03/fused_gather_pmc.csv).  This kernel is laid out from
// the register budget down -- 80 VGPRs, three workgroups per CU, six waves per SIMD:
//   * a PACK is 4 instances (one float4 of slot weights per table entry): 12 accumulator registers;
//   * WALK phase: the row is consumed through a rolling window of 4 entries in flight (no second register set, no copies);
//     at its end `coordinate = base + image` (poser_impl.inl:407) of instances 1..3 goes to LDS (`mp`, component-major: every
//     access is lane-consecutive), instance 0's stays in registers -- the accumulators are dead before the first matrix is read;
//   * SKIN phase, per instance: the vertex's coordinate comes back from LDS (the thread's own words: no barrier), blend + transform
//     as in deform_kernel, results scattered to the output images, ONE barrier, coalesced 16-byte copy-out.  The position image of
//     instance j is the `mp` region of instance j (its coordinates were read before the previous instance's barrier: everybody is
//     done with them), the normal image is double buffered -- 48.6 KB of LDS for 8 instances of the 50k model.
// Same operations in the same order as the reference (and as deform_kernel): bit-identical results.
constexpr int kPkThreads = 512;
constexpr uint32_t kPkPack = 4;
constexpr uint32_t kPkRegion = kSoaImgBytes;          // one `mp` region: 512 x 3 f32 coordinates, or a position image (f32: 6160 B, f16: 3088 B)
static_assert(kPkRegion >= kTileVerts * 12 && kPkRegion >= kP16ImgBytes && kPkRegion % 16 == 0, "mp region");

// The vertex's static data as the pack kernel keeps it across its loops: 14 registers (load_slot's Slot: 21).
struct PackSlot {
    v2f pxy, nxy;
    float pz, nz;
    float w0, w1, w2, w3;
    uint32_t b01, b23;        // palette entries of the vertex's bones (float4 index inside one instance's palette), two per register
    uint32_t meta;            // perm | deform class << 10 | active << 12
};
__device__ __forceinline__ PackSlot pack_slot(const Slot &q) {
    PackSlot s;
    s.pxy = q.pxy; s.nxy = q.nxy; s.pz = q.pz; s.nz = q.nz;
    s.w0 = q.w0; s.w1 = q.w1; s.w2 = q.w2; s.w3 = q.w3;
    s.b01 = q.b0 | (q.b1 << 16); s.b23 = q.b2 | (q.b3 << 16);
    s.meta = q.perm | (uint32_t(q.cls) << 10) | (q.act ? 1u << 12 : 0u);
    return s;
}
// skin_matrix() from the packed form.  `b01` / `b23` arrive through an opaque copy made inside the instance loop, so that the
// unpacked indices are not hoisted out of it as four loop-invariant registers.
__device__ __forceinline__ M12 skin_matrix_packed(uint32_t cls, uint32_t b01, uint32_t b23, float w0, float w1, float w2, float w3,
                                                  const float4 *P) {
    M12 m;
    if (cls == 0) {
        m = load_m12(P, b01 & 0xffffu);
    } else if (cls == 1) {
        const M12 a = load_m12(P, b01 >> 16), e = load_m12(P, b01 & 0xffffu);      // Lerp(S[b1], S[b0])[w]
        m = blend2(a, e, 1.0f - w0, w0);
        const bool lo = w0 < kLerpLo, hi = w0 > kLerpHi;
        if (__builtin_amdgcn_ballot_w64(lo || hi) != 0) {
            if (lo) m = a;
            else if (hi) m = e;
        }
    } else {
        {
            const M12 a = load_m12(P, b01 & 0xffffu), b = load_m12(P, b01 >> 16);
            m = mul_add(a, w0, b, w1);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const M12 c = load_m12(P, b23 & 0xffffu);
            m = add_mul(m, c, w2);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const M12 d = load_m12(P, b23 >> 16);
            m = add_mul(m, d, w3);
        }
    }
    return m;
}

#ifndef PK_WAVES
#define PK_WAVES 6
#endif
template <int LAYOUT, bool F16, bool FIN>
__global__ __launch_bounds__(kPkThreads, PK_WAVES) void pack_kernel(const DeformParams p) {
    static_assert(LAYOUT == MMDX_OUT_SOA || LAYOUT == MMDX_OUT_SOA_POS16, "two output arrays");
    static_assert(kTileVerts == 512, "one sorted slot per lane");
    using Raw = typename RawEntry<F16>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    uint32_t tile, grp;
    if (!map_workgroup(p, tile, grp)) return;       // padding workgroup
    const TileHdr &th = p.tiles[tile];
    const uint32_t v0 = th.v0, nvt = th.nv;
    // Instances of this workgroup: p.group / 4 quads.  Blocked: consecutive quads.  Interleaved (default): quad k of the workgroup is
    // quad grp + k * ngroups of the crowd, so that the workgroups running at the same time (neighbouring grp) write neighbouring
    // instances -- chip-wide the stores then sweep a few contiguous megabytes of each output array, as the crowd kernel's do.
    const bool ilv = p.interleave != 0;
    const uint32_t nquads = (p.ni + 3u) >> 2, qstep = ilv ? p.ngroups : 1u;
    const uint32_t quad0 = ilv ? grp : grp * (p.group >> 2);
    const uint32_t npacks = quad0 < nquads ? min(p.group >> 2, (nquads - quad0 + qstep - 1) / qstep) : 0u;
    const uint32_t gcount = npacks * kPkPack;        // group instances incl. the slack of a ragged last quad (never written out)
    auto first_instance = [&](uint32_t g0) { return (quad0 + (g0 >> 2) * qstep) * 4u; };
    stagger_start(p);
    float4 *pal = reinterpret_cast<float4 *>(smem);
    float4 *wq = reinterpret_cast<float4 *>(smem + p.w_off);
    unsigned char *mp = smem + p.mp_off;             // kPkPack regions
    unsigned char *imgB = smem + p.stage_off;        // 2 normal images, BEHIND mp (CopyFast wants image B after image A)
    const uint32_t wstride = p.ns + 1;
    auto pack_weight = [&](uint32_t g0, uint32_t i) {
        return reinterpret_cast<const float4 *>(p.wslot)[size_t(quad0 + (g0 >> 2) * qstep) * wstride + i];
    };
    // the lane's row of the morph table: lane `tid & 63` of the wave's slice (lanes past the tile's last vertex walk their
    // slice's padding: dummy slot, weight 0), padded length wave-uniform -> a scalar.  The row is the same for every pack.
    const uint2 sl2 = p.ell[(v0 + uint32_t(tid)) >> 6];
    const uint32_t rbase = sl2.x + (uint32_t(tid) & 63u);
    const uint32_t rlen = __builtin_amdgcn_readfirstlane(sl2.y);
    const uint32_t last = rlen ? rlen - 1 : 0u;
    Raw e[4];                                        // rolling window over the row: four entries in flight
    auto row_head = [&]() {
        uint32_t rb = rbase;      // (opaque copies like this one keep loop-invariant address arithmetic and operand splats from being
        asm volatile("" : "+v"(rb));   // carried around the loops in registers: they are recomputed where they are used)
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) e[i] = rlen ? row_entry<F16>(p.entries, rb, min(i, last)) : Raw{};
    };

    // set-up: the group's palettes, the first pack's slot weights, the head of the row, the vertex's static data
    // (one piece after the other: interleaved, their loads in flight would claim more registers than the loops below ever need,
    // and the allocator would answer by spilling the vertex's skin data for the whole kernel)
    stage_palettes<kPkThreads>(p, th, pal, quad0 * 4u, 1u, gcount, tid, qstep);
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t i = tid; i < wstride; i += kPkThreads) wq[i] = pack_weight(0, i);
    __builtin_amdgcn_sched_barrier(0);
    PackSlot ps;
    {
        Slot q;
        load_slot<LAYOUT, kMorphFused4, F16>(p, th, uint32_t(tid), q);
        ps = pack_slot(q);
    }
    __builtin_amdgcn_sched_barrier(0);
    row_head();
    const bool wreg = wstride <= uint32_t(kPkThreads);
    constexpr uint32_t kAlignVerts = LAYOUT == MMDX_OUT_SOA_POS16 ? 8u : 4u;
    const bool al = p.out_aligned != 0;
    const bool fast = al && nvt == kTileVerts && p.nv % kAlignVerts == 0 && v0 % kAlignVerts == 0;
    __syncthreads();

#ifdef PK_STAMPS
    // diagnostic build: where a wave's cycles go (s_memtime, one record of 8 counters per wave; never in the product)
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t;
#define PK_STAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#else
#define PK_STAMP(k) do { } while (0)
#endif
    uint32_t buf = 0;
    for (uint32_t g0 = 0; g0 < gcount; g0 += kPkPack) {
        PK_STAMP(0);
        // ---- WALK: vertex_image = 0; for each applied entry: image = image + offset*rate (poser_impl.inl:340-346), 4 instances --
        v2f axy[kPkPack], az01 = v2f{0.f, 0.f}, az23 = v2f{0.f, 0.f};
#pragma unroll
        for (uint32_t j = 0; j < kPkPack; ++j) axy[j] = v2f{0.f, 0.f};
        auto apply = [&](const Raw r) {
            float ox, oy, oz;
            uint32_t slot;
            if constexpr (F16) {
                ox = h2f(r.x & 0xffffu); oy = h2f(r.x >> 16); oz = h2f(r.y & 0xffffu); slot = r.y >> 16;
            } else {
                ox = r.x; oy = r.y; oz = r.z; slot = __float_as_uint(r.w);
            }
            const float4 w = wq[slot];
            const v2f oxy = v2f{ox, oy}, ozz = v2f{oz, oz};
            if constexpr (FIN) {
                // a skipped slot weighs +0 exactly and a sum that started at +0 is never -0: with finite offsets
                // "image + offset*0" leaves the image bit for bit unchanged -- no branch
                axy[0] += oxy * w.x; axy[1] += oxy * w.y; axy[2] += oxy * w.z; axy[3] += oxy * w.w;
                az01 += ozz * v2f{w.x, w.y};
                az23 += ozz * v2f{w.z, w.w};
            } else {
                const v2f t0 = axy[0] + oxy * w.x, t1 = axy[1] + oxy * w.y, t2 = axy[2] + oxy * w.z, t3 = axy[3] + oxy * w.w;
                const v2f u01 = az01 + ozz * v2f{w.x, w.y}, u23 = az23 + ozz * v2f{w.z, w.w};
                const bool k0 = w.x < kMorphEps, k1 = w.y < kMorphEps, k2 = w.z < kMorphEps, k3 = w.w < kMorphEps;
                axy[0] = k0 ? axy[0] : t0; axy[1] = k1 ? axy[1] : t1; axy[2] = k2 ? axy[2] : t2; axy[3] = k3 ? axy[3] : t3;
                az01 = v2f{k0 ? az01.x : u01.x, k1 ? az01.y : u01.y};
                az23 = v2f{k2 ? az23.x : u23.x, k3 ? az23.y : u23.y};
            }
        };
#ifdef PK_SKIP_WALK          // timing diagnostic only (wrong results): what the kernel costs without its walk
        if (false) {
#else
        if (rlen) {
#endif
            // Entry j+4 is requested as soon as entry j has been consumed.  The steady-state loop has no branch in its body (a
            // refill past the row's end re-reads the last entry, which is never applied twice): the compiler counts the loads in
            // flight and waits for exactly the oldest one.  The window's first four entries were requested by the set-up / in
            // front of the previous pack's last stores.
            uint32_t rb = rbase;
            asm volatile("" : "+v"(rb));
            uint32_t j = 0;
            for (; j + 4 < rlen; j += 4) {
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    apply(e[i]);
                    e[i] = row_entry<F16>(p.entries, rb, min(j + 4 + i, last));
#ifndef PK_NO_SCHED_BARRIER
                    // one entry at a time: interleaving the four entries' products would need the registers that hold the vertex's
                    // skin data, which would then be spilled here and RELOADED between the instances' stores (see SKIN)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
            apply(e[0]);
            if (j + 1 < rlen) apply(e[1]);
            if (j + 2 < rlen) apply(e[2]);
            if (j + 3 < rlen) apply(e[3]);
        }
        // the next pack's slot weights: requested now, in LDS after this pack's instances
        const bool more = g0 + kPkPack < gcount;
        float4 wnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (more && wreg && uint32_t(tid) < wstride) wnext = pack_weight(g0 + kPkPack, uint32_t(tid));
        // coordinate = base + image (poser_impl.inl:407): instance 0 in registers, 1..3 to this thread's words of mp
        v2f cxy = ps.pxy + axy[0];
        float cz = ps.pz + az01.x;
        {
            int t = tid;
            asm volatile("" : "+v"(t));
            float *m1 = reinterpret_cast<float *>(mp + 1 * kPkRegion), *m2 = reinterpret_cast<float *>(mp + 2 * kPkRegion),
                  *m3 = reinterpret_cast<float *>(mp + 3 * kPkRegion);
            const v2f c1 = ps.pxy + axy[1], c2 = ps.pxy + axy[2], c3 = ps.pxy + axy[3];
            m1[t] = c1.x; m1[kTileVerts + t] = c1.y; m1[2 * kTileVerts + t] = ps.pz + az01.y;
            m2[t] = c2.x; m2[kTileVerts + t] = c2.y; m2[2 * kTileVerts + t] = ps.pz + az23.x;
            m3[t] = c3.x; m3[kTileVerts + t] = c3.y; m3[2 * kTileVerts + t] = ps.pz + az23.y;
        }
        PK_STAMP(1);
        // ---- SKIN: the pack's instances, one barrier each.  No vector-memory LOAD may sit in here: it would have to wait for the
        // previous instance's stores to drain (vmcnt is in order), microseconds under a saturated HBM. ---------------------------
        auto instance = [&](const uint32_t j, auto last_tag) {
            constexpr bool kLast = decltype(last_tag)::value;       // the pack's fourth instance: the next walk's row head goes out with it
            const uint32_t inst = first_instance(g0) + j;
            const size_t vbase = size_t(inst) * p.nv + v0;
            const uint32_t sh4 = al ? uint32_t((vbase * 3) & 3) : 0u, sh8 = al ? uint32_t((vbase * 3) & 7) : 0u;
            unsigned char *ia = mp + j * kPkRegion, *ib = imgB + buf * kSoaImgBytes;
            // (in place: the values stay in their registers, the compiler merely stops treating them as loop invariants -- whose
            // derived values it would otherwise carry around the loops in registers of their own)
            asm volatile("" : "+v"(ps.meta), "+v"(ps.b01), "+v"(ps.b23), "+v"(ps.w0), "+v"(ps.w1), "+v"(ps.w2), "+v"(ps.w3), "+v"(ps.nxy),
                         "+v"(ps.nz));
            const uint32_t meta = ps.meta, b01 = ps.b01, b23 = ps.b23;
            const float w0 = ps.w0, w1 = ps.w1, w2 = ps.w2, w3 = ps.w3, nx = ps.nxy.x, ny = ps.nxy.y, nz = ps.nz;
            int t = tid;
            asm volatile("" : "+v"(t));
            if (meta >> 12) {
                const M12 m = skin_matrix_packed((meta >> 10) & 3u, b01, b23, w0, w1, w2, w3, pal + size_t(g0 + j) * p.pal_stride);
                v2f oxy, rxy;
                float oz, rz;
                xform_pos(m, cxy, cz, oxy, oz);
                xform_nrm(m, v2f{nx, ny}, nz, rxy, rz);
                oxy = oxy * p.pos_scale;                 // a separate multiply after the transform (main.cpp:848-850)
                oz = oz * p.pos_scale;
                const uint32_t perm3 = (meta & 1023u) * 3;
                float *B = reinterpret_cast<float *>(ib) + sh4 + perm3;
                if constexpr (LAYOUT == MMDX_OUT_SOA) {
                    float *A = reinterpret_cast<float *>(ia) + sh4 + perm3;
                    A[0] = oxy.x; A[1] = oxy.y; A[2] = oz;
                } else {
                    unsigned short *A = reinterpret_cast<unsigned short *>(ia) + sh8 + perm3;
                    A[0] = f2h(oxy.x); A[1] = f2h(oxy.y); A[2] = f2h(oz);
                }
                B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
            }
            if (!kLast) {                   // the next instance's coordinate, before anybody may overwrite it with that instance's image
                const float *mn = reinterpret_cast<const float *>(mp + (j + 1) * kPkRegion);
                cxy = v2f{mn[t], mn[kTileVerts + t]};
                cz = mn[2 * kTileVerts + t];
            }
            PK_STAMP(2);
            __syncthreads();
            PK_STAMP(3);
            // the head of the row for the next pack's walk goes out IN FRONT of the pack's last stores: its round trip then runs
            // beside their drain instead of behind it
            if constexpr (kLast) { if (more) row_head(); }
            if constexpr (LAYOUT == MMDX_OUT_SOA) {
                float *oa = reinterpret_cast<float *>(p.out_a), *ob = reinterpret_cast<float *>(p.out_b);
                if (fast)
                    copy_out_fast<kPkThreads, kTileVerts * 12 / 16, kTileVerts * 12 / 16>(
                        ia, uint32_t(ib - ia), reinterpret_cast<float4 *>(oa + vbase * 3), reinterpret_cast<float4 *>(ob + vbase * 3), t);
                else
                    copy_out2<kPkThreads, float, float>(ia, oa, vbase * 3, sh4, nvt * 3, ib, ob, vbase * 3, sh4, nvt * 3, al, t);
            } else {
                unsigned short *oa = reinterpret_cast<unsigned short *>(p.out_a);
                float *ob = reinterpret_cast<float *>(p.out_b);
                if (fast)
                    copy_out_fast<kPkThreads, kTileVerts * 6 / 16, kTileVerts * 12 / 16>(
                        ia, uint32_t(ib - ia), reinterpret_cast<float4 *>(oa + vbase * 3), reinterpret_cast<float4 *>(ob + vbase * 3), t);
                else
                    copy_out2<kPkThreads, unsigned short, float>(ia, oa, vbase * 3, sh8, nvt * 3, ib, ob, vbase * 3, sh4, nvt * 3, al, t);
            }
            buf ^= 1u;
            PK_STAMP(4);
        };
#ifdef PK_SKIP_SKIN
        if (cz == 12345.678f && cxy.x == 9.75f) instance(0, std::true_type{});      // (keeps the walk's results alive)
#else
#pragma unroll 1
        for (uint32_t j = 0; j + 1 < kPkPack; ++j) {
            if (first_instance(g0) + j >= p.ni) break;
            instance(j, std::false_type{});
        }
        if (first_instance(g0) + kPkPack - 1 < p.ni) instance(kPkPack - 1, std::true_type{});
#endif
        if (more) {
            // every wave has left this pack's walk (it passed the instances' barriers): the weights can be replaced.  The barrier
            // also closes the last instance's copy-out reads before the next walk's coordinates land in mp.
            if (wreg) {
                if (uint32_t(tid) < wstride) wq[tid] = wnext;
            } else {
                for (uint32_t i = tid; i < wstride; i += kPkThreads) wq[i] = pack_weight(g0 + kPkPack, i);
            }
            __syncthreads();
        }
        PK_STAMP(5);
    }
#ifdef PK_STAMPS
    if ((tid & 63) == 0 && p.stamps) {
        unsigned long long *o = p.stamps + (size_t(blockIdx.x) * 8 + (uint32_t(tid) >> 6)) * 8;
        for (int k = 0; k < 6; ++k) o[k] = st_acc[k];
        o[6] = st_begin; o[7] = st_t;
    }
#endif
}

// ---- ONE frame of ONE model (ni == 1): the reference's per-frame call (main.cpp:1821) -------------------------------------
// A single frame is a few megabytes: its run time is the latency chain of a workgroup, not bandwidth.  So, unlike the crowd
// kernel: a workgroup is a PART of a tile (THREADS sorted slots, one per lane: config 2's 98 tiles become 392 workgroups of two
// waves, on every CU), requests are issued in latency order (setup_fused1), and every lane stores its own vertex straight from
// registers to its original index -- no LDS image, no second barrier; the scattered 12-byte stores of 1.2 MB merge in L2.
// Same arithmetic, same order.
template <int THREADS, int LAYOUT, int MORPH, bool F16>
__global__ __launch_bounds__(THREADS) void frame_kernel(const DeformParams p) {
    constexpr uint32_t kParts = kTileVerts / THREADS;
    constexpr uint32_t BH = F16 ? 16u : 8u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x / kParts, part = blockIdx.x - tile * kParts;
    const TileHdr &th = p.tiles[tile];
    if (part * THREADS >= th.nv) return;                       // the whole workgroup: nothing of this tile left for it
    float4 *pal = reinterpret_cast<float4 *>(smem);
    float *wl = reinterpret_cast<float *>(smem + p.w_off);
    Slot sl[1];
    RowHead<F16, BH> hd[1];
    setup_fused1<THREADS, LAYOUT, MORPH, F16, 1, BH>(p, th, pal, wl, 0u, 1u, 1u, tid, part * THREADS, sl, hd);
    __syncthreads();
    const Slot &q = sl[0];
    if (!q.act) return;
    v2f cxy = q.pxy;
    float cz = q.pz;
    if constexpr (MORPH == kMorphFused1) {
        // vertex_image = 0; for each applied entry: image = image + offset*rate (poser_impl.inl:340-346); coordinate = base + image
        v2f dxy = v2f{0.f, 0.f};
        float dz = 0.f;
        row_consume<F16, BH>(p.entries, q.rb, q.rlen, hd[0], [&](float ox, float oy, float oz, uint32_t slot) {
            const float w = wl[slot];
            if (!(w < kMorphEps)) { dxy = dxy + v2f{ox, oy} * w; dz = dz + oz * w; }
        });
        cxy = q.pxy + dxy; cz = q.pz + dz;
        if (p.morphed) {   // kept (sorted order) for later calls that declare the rates unchanged
            float *mo = p.morphed + (size_t(th.v0) + part * THREADS + uint32_t(tid)) * 3;
            mo[0] = cxy.x; mo[1] = cxy.y; mo[2] = cz;
        }
    }
    const M12 m = skin_matrix(q, pal);
    v2f oxy, rxy;
    float oz, rz;
    xform_pos(m, cxy, cz, oxy, oz);
    xform_nrm(m, q.nxy, q.nz, rxy, rz);
    oxy = oxy * p.pos_scale;                                   // a separate multiply after the transform (main.cpp:848-850)
    oz = oz * p.pos_scale;
    // original vertex index -- or, for MMDX_CREATE_TILE_ORDER models, the vertex's sorted slot
    const size_t v = size_t(th.v0) + (p.tile_order ? part * THREADS + uint32_t(tid) : q.perm);
    if constexpr (LAYOUT == MMDX_OUT_SOA) {
        float *A = reinterpret_cast<float *>(p.out_a) + v * 3, *B = reinterpret_cast<float *>(p.out_b) + v * 3;
        A[0] = oxy.x; A[1] = oxy.y; A[2] = oz;
        B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
    } else if constexpr (LAYOUT == MMDX_OUT_VERTEX32) {
        float *A = reinterpret_cast<float *>(p.out_a) + v * 8;
        if (p.out_aligned) {
            reinterpret_cast<float4 *>(A)[0] = make_float4(oxy.x, oxy.y, oz, rxy.x);
            reinterpret_cast<float4 *>(A)[1] = make_float4(rxy.y, rz, q.uv.x, q.uv.y);
        } else {
            A[0] = oxy.x; A[1] = oxy.y; A[2] = oz; A[3] = rxy.x; A[4] = rxy.y; A[5] = rz; A[6] = q.uv.x; A[7] = q.uv.y;
        }
    } else {
        unsigned short *A = reinterpret_cast<unsigned short *>(p.out_a) + v * 3;
        float *B = reinterpret_cast<float *>(p.out_b) + v * 3;
        A[0] = f2h(oxy.x); A[1] = f2h(oxy.y); A[2] = f2h(oz);
        B[0] = rxy.x; B[1] = rxy.y; B[2] = rz;
    }
}

// ---- shared morph pass: morphed[gs] = base[gs] + sum(offset*rate), once per call ------------------
// FUSED_FLATTEN: every workgroup first evaluates the group-morph chains of all slots into LDS
// (UpdateMorphTransform's recursion; NS is a few hundred), saving the separate flatten launch.
template <bool F16, bool FUSED_FLATTEN>
__global__ __launch_bounds__(kThreads) void morph_apply_kernel(const DeformParams p,
                                                               const FlattenParams f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t last_block;
    const float *wsl = p.wslot;
    bool walk = true;
    if constexpr (FUSED_FLATTEN) {
        // vertex_images_ depends on morph_rates_ only (poser_impl.inl:362-386): when this call's rates are, bit for bit, the ones
        // `morphed` was last computed from (f.seen, written by the last workgroup of that launch), there is nothing to do.
        if (f.seen) {
            bool same = f.seen[kSeenValid] == 1u;
            for (uint32_t m = threadIdx.x; m < f.nm; m += kThreads) same = same && __float_as_uint(f.rates[m]) == f.seen[kSeenRates + m];
            walk = __syncthreads_and(same) == 0;
        }
        if (walk) {
            float *wl = reinterpret_cast<float *>(smem);
            for (uint32_t s = threadIdx.x; s <= f.ns; s += kThreads)     // slot ns = padding, weight 0
                wl[s] = s < f.ns ? slot_weight(f.rates, f.slot_top, f.chain_off, f.chain_rate, s) : 0.f;
            __syncthreads();
            wsl = wl;
        }
    }
    const size_t gs = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (walk && gs < p.nv) {
        float bx, by, bz;
        if constexpr (F16) {
            const uint2 r = reinterpret_cast<const uint2 *>(p.spos)[gs];
            bx = h2f(r.x & 0xffffu); by = h2f(r.x >> 16); bz = h2f(r.y & 0xffffu);
        } else {
            const float *sp = reinterpret_cast<const float *>(p.spos) + gs * 3;
            bx = sp[0]; by = sp[1]; bz = sp[2];
        }
        v2f dxy = v2f{0.f, 0.f};
        float dz = 0.f;
        const uint2 sl2 = p.ell[gs >> 6];
        // a thread of this pass has nothing else in its registers: 16 entries in flight cover most rows in one round trip
        for_row<F16, 16>(p.entries, sl2.x + uint32_t(gs & 63), sl2.y, [&](float ox, float oy, float oz, uint32_t slot) {
            const float w = wsl[slot];
            if (!(w < kMorphEps)) { dxy = dxy + v2f{ox, oy} * w; dz = dz + oz * w; }
        });
        p.morphed[gs * 3] = bx + dxy.x;
        p.morphed[gs * 3 + 1] = by + dxy.y;
        p.morphed[gs * 3 + 2] = bz + dz;
    }
    if constexpr (FUSED_FLATTEN) {
        // The workgroup that finishes LAST records the rates (every other one has compared by then: its ticket comes after its
        // reads), so no workgroup ever compares against a half-written record; the next launch on the stream sees it whole.
        if (f.seen) {
            if (threadIdx.x == 0) last_block = atomicAdd(&f.seen[kSeenTicket], 1u) == gridDim.x - 1 ? 1u : 0u;
            __syncthreads();
            if (last_block) {
                for (uint32_t m = threadIdx.x; m < f.nm; m += kThreads) f.seen[kSeenRates + m] = __float_as_uint(f.rates[m]);
                if (threadIdx.x == 0) {
                    f.seen[kSeenValid] = 1u;
                    f.seen[kSeenTicket] = 0u;
                    f.seen[walk ? kSeenWalks : kSeenSkips] += 1u;
                }
            }
        }
    }
}

#ifndef MMDX_FAST_MATH
// ---- group-morph flattening on the device (UpdateMorphTransform's recursion, per slot) ------------
__global__ __launch_bounds__(kThreads) void flatten_kernel(const FlattenParams f) {
    // output rows have ns+1 columns: column ns is the padding slot of the morph table, always 0
    const size_t idx = size_t(blockIdx.x) * kThreads + threadIdx.x;
    const uint32_t rows = f.quad ? ((f.niw + 3) / 4) * 4 : f.niw, cols = f.ns + 1;
    if (idx >= size_t(rows) * cols) return;
    const uint32_t i = uint32_t(idx / cols), s = uint32_t(idx - size_t(i) * cols);
    float out = 0.f;
    if (i < f.niw && s < f.ns)
        out = slot_weight(f.rates + size_t(i) * f.nm, f.slot_top, f.chain_off, f.chain_rate, s);
    if (f.quad) f.out[(size_t(i / 4) * cols + s) * 4 + (i & 3)] = out;
    else f.out[idx] = out;
}

// ---- VMD morph tracks -> per-instance morph rates (Motion::GetMorphPose, motion_impl.inl:382-424) ---
// One thread per (instance, model morph): clamp to the first / last key, exact hit, else the linear
// blend l*(1-t) + r*t with t = float(frame-left)/float(right-left) (IEEE division: hipcc keeps f32
// division correctly rounded by default).  A morph without a track keeps rate 0, as after ResetPosing.
__global__ __launch_bounds__(kThreads) void morph_track_eval_kernel(const MorphTrackParams t) {
    const size_t idx = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (idx >= size_t(t.ni) * t.nm) return;
    const uint32_t i = uint32_t(idx / t.nm), m = uint32_t(idx - size_t(i) * t.nm);
    const uint32_t b = t.key_off[m], e = t.key_off[m + 1], frame = t.frames[i];
    float w = 0.f;
    if (e > b) {
        if (t.key_frames[b] >= frame) {
            w = t.key_weights[b];
        } else if (t.key_frames[e - 1] <= frame) {
            w = t.key_weights[e - 1];
        } else {
            uint32_t lo = b, hi = e - 1;               // key_frames[lo] < frame < key_frames[hi]
            while (hi - lo > 1) {                      // first key whose frame is > `frame`
                const uint32_t mid = (lo + hi) / 2;
                if (t.key_frames[mid] > frame) hi = mid; else lo = mid;
            }
            const uint32_t lf = t.key_frames[lo], rf = t.key_frames[hi];
            if (lf == frame) {
                w = t.key_weights[lo];
            } else {
                const float bary = float(frame - lf) / float(rf - lf);
                w = t.key_weights[lo] * (1.0f - bary) + t.key_weights[hi] * bary;
            }
        }
    }
    t.out[idx] = w;
}

// ---- streaming copy / fill: the practical HBM ceiling printed next to the roofline ---------------
// Every workgroup owns one contiguous 4 KiB chunk, workgroups in address order: the shape that
// reached the highest store rate on MI355X in tools/archive/probes/bw_probe (a few-thousand-block grid-stride loop
// is 30 % slower).
__global__ __launch_bounds__(kThreads) void copy_kernel(float4 *dst, const float4 *src, size_t n) {
    const size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(kThreads) void fill_kernel(float4 *dst, size_t n) {
    const size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// Store-only replay of the deform kernel's output pattern: workgroup = (512-vertex tile, group of 16
// instances) writing one piece (6 KiB for SoA) into each output array per instance.  Its rate is BIMODAL
// on MI355X: for some placements of the arrays it runs at the linear-fill rate, for others ~25 % below,
// and the deform kernel follows it (tools/archive/probes/alloc_kernel_probe.py).  Used as the ceiling bench.py prints
// and as the probe of mmdx_crowd_output_alloc().
__global__ __launch_bounds__(kThreads) void pattern_fill_kernel(float4 *a, float4 *b, uint32_t nv,
                                                                uint32_t ni, uint32_t ntiles, uint32_t bpva,
                                                                uint32_t bpvb) {
    const uint32_t tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const uint32_t v0 = tile * kTileVerts, nvt = min(kTileVerts, nv - v0);
    const uint32_t pa = nvt * bpva / 16, pb = nvt * bpvb / 16;   // callers keep nv * bytes-per-vertex % 16 == 0
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t g = grp * 16; g < min(ni, grp * 16 + 16); ++g) {
        const size_t base_a = (size_t(g) * nv + v0) * bpva / 16, base_b = (size_t(g) * nv + v0) * bpvb / 16;
        for (uint32_t q = threadIdx.x; q < pa + pb; q += kThreads) {
            store16(q < pa ? a + base_a + q : b + base_b + (q - pa), v);      // the deform kernel's store instruction (policy and all)
        }
    }
}

#endif  // !MMDX_FAST_MATH

using KernelFn = void (*)(const DeformParams);

template <int THREADS, int LAYOUT, bool F16, bool TILE>
KernelFn pick_morph(int morph) {
    switch (morph) {
    case kMorphNone: return deform_kernel<THREADS, LAYOUT, kMorphNone, F16, TILE>;
    case kMorphShared: return deform_kernel<THREADS, LAYOUT, kMorphShared, F16, TILE>;
    case kMorphFused1: return deform_kernel<THREADS, LAYOUT, kMorphFused1, F16, TILE>;
    default: return deform_kernel<THREADS, LAYOUT, kMorphFused4, F16, TILE>;
    }
}

template <int THREADS, bool TILE>
KernelFn pick_t(int layout, int morph, bool f16) {
    if (f16) return layout == MMDX_OUT_SOA_POS16 ? pick_morph<THREADS, MMDX_OUT_SOA_POS16, true, TILE>(morph) : nullptr;
    if (layout == MMDX_OUT_SOA) return pick_morph<THREADS, MMDX_OUT_SOA, false, TILE>(morph);
    if (layout == MMDX_OUT_VERTEX32) return pick_morph<THREADS, MMDX_OUT_VERTEX32, false, TILE>(morph);
    return nullptr;
}

// tile = outputs in the engine's vertex order (MMDX_CREATE_TILE_ORDER): the direct-store variants.
// wt = write-through stores (CopyFast): instantiated where it was measured to pay -- the SoA f32 crowd kernels (256 threads, no
// morphs or shared morphs, original vertex order); every other shape keeps its nt stores whatever the hint says.
KernelFn pick(int threads, int layout, int morph, bool f16, bool tile, bool wt = false) {
    if (wt && deform_has_write_through(threads, layout, morph, f16, tile))
        return morph == kMorphNone ? deform_kernel<256, MMDX_OUT_SOA, kMorphNone, false, false, true>
                                   : deform_kernel<256, MMDX_OUT_SOA, kMorphShared, false, false, true>;
#if MMDX_TILE >= 512
    if (threads != 256) return tile ? pick_t<512, true>(layout, morph, f16) : pick_t<512, false>(layout, morph, f16);
#endif
    return tile ? pick_t<256, true>(layout, morph, f16) : pick_t<256, false>(layout, morph, f16);
}

}  // namespace

namespace {
template <int THREADS, int LAYOUT, bool F16>
KernelFn pick_frame_morph(int morph) {
    return morph == kMorphNone ? frame_kernel<THREADS, LAYOUT, kMorphNone, F16> : frame_kernel<THREADS, LAYOUT, kMorphFused1, F16>;
}
KernelFn pick_frame(int threads, int layout, int morph, bool f16) {
    if (morph != kMorphNone && morph != kMorphFused1) return nullptr;
    if (f16) {
        if (layout != MMDX_OUT_SOA_POS16) return nullptr;
        return threads == 128 ? pick_frame_morph<128, MMDX_OUT_SOA_POS16, true>(morph) : pick_frame_morph<256, MMDX_OUT_SOA_POS16, true>(morph);
    }
    if (layout == MMDX_OUT_SOA)
        return threads == 128 ? pick_frame_morph<128, MMDX_OUT_SOA, false>(morph) : pick_frame_morph<256, MMDX_OUT_SOA, false>(morph);
    if (layout == MMDX_OUT_VERTEX32)
        return threads == 128 ? pick_frame_morph<128, MMDX_OUT_VERTEX32, false>(morph) : pick_frame_morph<256, MMDX_OUT_VERTEX32, false>(morph);
    return nullptr;
}
}  // namespace

#ifndef MMDX_FAST_MATH
size_t deform_lds_bytes(int threads, int layout, int morph, uint32_t group, uint32_t max_tile_bones, uint32_t ns,
                        uint32_t *stage_off, uint32_t *w_off, bool tile_order) {
    size_t off = size_t(group) * max_tile_bones * 48;
    *stage_off = uint32_t(off);
    if (!tile_order) off += 2 * size_t(stage_bytes(layout));     // tile-order outputs need no LDS image
    *w_off = uint32_t(off);
    if (morph == kMorphFused1) off += (size_t(ns + 1) * 4 + 15) / 16 * 16;
    else if (morph == kMorphFused4) off += (threads == 512 ? 2 : 1) * size_t(ns + 1) * 16;   // one or two instance quads
    return off;
}

#endif  // !MMDX_FAST_MATH

#ifndef MMDX_FAST_MATH
// pack_kernel: [palettes of the group][slot weights of one pack][kPkPack coordinate / position-image regions][2 normal images]
size_t pack_lds_bytes(uint32_t group, uint32_t max_tile_bones, uint32_t ns, uint32_t *stage_off, uint32_t *w_off, uint32_t *mp_off) {
    size_t off = size_t(group) * max_tile_bones * 48;
    *w_off = uint32_t(off);
    off += size_t(ns + 1) * 16;
    *mp_off = uint32_t(off);
    off += size_t(kPkPack) * kPkRegion;
    *stage_off = uint32_t(off);
    off += 2 * size_t(kSoaImgBytes);
    return off;
}
#endif  // !MMDX_FAST_MATH

namespace {
KernelFn pick_pack(int layout, bool f16, bool finite) {
    if (f16) {
        if (layout != MMDX_OUT_SOA_POS16) return nullptr;
        return finite ? pack_kernel<MMDX_OUT_SOA_POS16, true, true> : pack_kernel<MMDX_OUT_SOA_POS16, true, false>;
    }
    if (layout != MMDX_OUT_SOA) return nullptr;
    return finite ? pack_kernel<MMDX_OUT_SOA, false, true> : pack_kernel<MMDX_OUT_SOA, false, false>;
}
}  // namespace

hipError_t MMDX_K(launch_pack)(int layout, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes, hipStream_t stream) {
    KernelFn fn = pick_pack(layout, f16, p.finite_offsets != 0);
    if (!fn || kTileVerts != 512) return hipErrorInvalidValue;
    DeformParams q = p;
    q.ntiles = ntiles;
    q.ngroups = (p.ni + p.group - 1) / p.group;
    q.rem_per_xcd = ((ntiles & 7u) * q.ngroups + 7u) / 8u;
    const dim3 grid(8u * ((ntiles >> 3) * q.ngroups + q.rem_per_xcd));
#ifdef PK_STAMPS
    static unsigned long long *stamps = nullptr;
    static size_t stamps_n = 0;
    const size_t need = size_t(grid.x) * 64;
    if (stamps_n < need) { if (stamps) (void)hipFree(stamps); (void)hipMalloc(&stamps, need * 8); stamps_n = need; }
    (void)hipMemsetAsync(stamps, 0, need * 8, stream);
    q.stamps = stamps;
#endif
    hipLaunchKernelGGL(fn, grid, dim3(kPkThreads), lds_bytes, stream, q);
#ifdef PK_STAMPS
    {
        static int calls = 0;
        if (++calls % 16 == 0) {            // now and then: wait, fetch, print the means (diagnostic build only)
            (void)hipStreamSynchronize(stream);
            std::vector<unsigned long long> h(need);
            (void)hipMemcpy(h.data(), stamps, need * 8, hipMemcpyDeviceToHost);
            const char *names[6] = {"sync/top", "walk", "skin", "barrier", "copy-out", "pack-end"};
            for (int w : {0, 3, 7}) {
                double acc[6] = {0, 0, 0, 0, 0, 0}, life = 0; size_t n = 0;
                unsigned long long t0 = ~0ull, t1 = 0;
                for (size_t b = 0; b < grid.x; ++b) {
                    const unsigned long long *r = h.data() + (b * 8 + w) * 8;
                    if (!r[7]) continue;
                    for (int k = 0; k < 6; ++k) acc[k] += double(r[k]);
                    life += double(r[7] - r[6]); ++n;
                    t0 = std::min(t0, r[6]); t1 = std::max(t1, r[7]);
                }
                if (!n) continue;
                std::fprintf(stderr, "pk-stamps wave %d: n=%zu life %.0f ticks |", w, n, life / n);
                for (int k = 0; k < 6; ++k) std::fprintf(stderr, " %s %.0f (%.0f%%)", names[k], acc[k] / n, 100.0 * acc[k] / life);
                std::fprintf(stderr, " | kernel span %llu ticks\n", t1 - t0);
            }
        }
    }
#endif
    return hipGetLastError();
}

hipError_t MMDX_K(prepare_kernels)() {
    for (int f16 = 0; f16 < 2; ++f16)
        for (int fin = 0; fin < 2; ++fin) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pick_pack(f16 ? MMDX_OUT_SOA_POS16 : MMDX_OUT_SOA, f16 != 0, fin != 0)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    for (int threads = 256; threads <= 512; threads += 256)
      for (int f16 = 0; f16 < 2; ++f16)
        for (int layout = 0; layout < 3; ++layout)
            for (int morph = 0; morph < 8; ++morph) {
                KernelFn fn = pick(threads, layout, morph & 3, f16 != 0, morph >= 4);
                if (!fn) continue;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   160 * 1024);
                if (e != hipSuccess) return e;
            }
    for (int morph : {int(kMorphNone), int(kMorphShared)}) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pick(256, MMDX_OUT_SOA, morph, false, false, true)),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    for (int threads = 128; threads <= 256; threads += 128)
      for (int f16 = 0; f16 < 2; ++f16)
        for (int layout = 0; layout < 3; ++layout)
            for (int morph = 0; morph <= kMorphFused1; morph += kMorphFused1) {
                KernelFn fn = pick_frame(threads, layout, morph, f16 != 0);
                if (!fn) continue;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t MMDX_K(launch_deform)(int threads, int layout, int morph, bool f16, const DeformParams &p,
                         uint32_t ntiles, size_t lds_bytes, hipStream_t stream) {
    KernelFn fn = pick(threads, layout, morph, f16, p.tile_order != 0, p.write_through != 0);
    if (!fn) return hipErrorInvalidValue;
    DeformParams q = p;
    q.ntiles = ntiles;
    q.ngroups = (p.ni + p.group - 1) / p.group;
    q.rem_per_xcd = ((ntiles & 7u) * q.ngroups + 7u) / 8u;
    const dim3 grid(8u * ((ntiles >> 3) * q.ngroups + q.rem_per_xcd));
    hipLaunchKernelGGL(fn, grid, dim3((threads == 256 || kTileVerts < 512) ? 256 : 512), lds_bytes, stream, q);
    return hipGetLastError();
}

#ifndef MMDX_FAST_MATH
size_t frame_lds_bytes(int morph, uint32_t max_tile_bones, uint32_t ns, uint32_t *w_off) {
    size_t off = (size_t(max_tile_bones) * 48 + 15) / 16 * 16;
    *w_off = uint32_t(off);
    if (morph == kMorphFused1) off += (size_t(ns + 1) * 4 + 15) / 16 * 16;
    return off;
}

#endif  // !MMDX_FAST_MATH

hipError_t MMDX_K(launch_frame)(int threads, int layout, int morph, bool f16, const DeformParams &p, uint32_t ntiles, size_t lds_bytes,
                        hipStream_t stream) {
    threads = threads == 128 ? 128 : 256;
    KernelFn fn = pick_frame(threads, layout, morph, f16);
    if (!fn) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3(ntiles * (kTileVerts / uint32_t(threads))), dim3(threads), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t MMDX_K(launch_morph_apply)(bool f16, const DeformParams &p, const FlattenParams *fused,
                              hipStream_t stream) {
    const dim3 grid((p.nv + kThreads - 1) / kThreads);
    if (fused) {
        const size_t lds = size_t(fused->ns + 1) * 4;
        if (f16) hipLaunchKernelGGL((morph_apply_kernel<true, true>), grid, dim3(kThreads), lds, stream, p, *fused);
        else hipLaunchKernelGGL((morph_apply_kernel<false, true>), grid, dim3(kThreads), lds, stream, p, *fused);
    } else {
        FlattenParams none{};
        if (f16) hipLaunchKernelGGL((morph_apply_kernel<true, false>), grid, dim3(kThreads), 0, stream, p, none);
        else hipLaunchKernelGGL((morph_apply_kernel<false, false>), grid, dim3(kThreads), 0, stream, p, none);
    }
    return hipGetLastError();
}

#ifndef MMDX_FAST_MATH
hipError_t launch_pattern_fill(void *a, void *b, uint32_t nv, uint32_t ni, uint32_t bpva, uint32_t bpvb,
                               hipStream_t stream) {
    const uint32_t ntiles = (nv + kTileVerts - 1) / kTileVerts;
    hipLaunchKernelGGL(pattern_fill_kernel, dim3(ntiles * ((ni + 15) / 16)), dim3(kThreads), 0, stream,
                       reinterpret_cast<float4 *>(a), reinterpret_cast<float4 *>(b), nv, ni, ntiles, bpva, bpvb);
    return hipGetLastError();
}

hipError_t launch_flatten(const FlattenParams &f, hipStream_t stream) {
    const uint32_t rows = f.quad ? ((f.niw + 3) / 4) * 4 : f.niw;
    const size_t n = size_t(rows) * (f.ns + 1);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(flatten_kernel, dim3(uint32_t((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, stream, f);
    return hipGetLastError();
}

hipError_t launch_morph_track_eval(const MorphTrackParams &t, hipStream_t stream) {
    const size_t n = size_t(t.ni) * t.nm;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(morph_track_eval_kernel, dim3(uint32_t((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, stream, t);
    return hipGetLastError();
}

hipError_t launch_copy(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(copy_kernel, dim3(uint32_t((bytes / 16 + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, stream, reinterpret_cast<float4 *>(dst),
                       reinterpret_cast<const float4 *>(src), bytes / 16);
    return hipGetLastError();
}

hipError_t launch_fill(void *dst, size_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(uint32_t((bytes / 16 + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, stream, reinterpret_cast<float4 *>(dst), bytes / 16);
    return hipGetLastError();
}

#endif  // !MMDX_FAST_MATH

}  // namespace mmdx
