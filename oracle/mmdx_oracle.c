/* oracle/mmdx_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * From-scratch, scalar, single-thread C restatement of the reference's per-frame deformation path
 * (CU-Production/simple_mmd_renderer: vendored libmmd + main.cpp repack).  It is the CHECKER for the
 * HIP path; the product never calls it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Parity status: PINNED.  The reference ships no golden vectors for this path, so
 * the restatement is pinned (a) against the real libmmd compiled in the build container
 * (oracle/_ref/libmmd_ref.so, tests/test_oracle_vs_reference.py: bit-exact on random + edge-case
 * models) and (b) against the golden vectors that library produced, committed under tests/golden/
 * (tests/test_oracle_golden.py runs everywhere, including the GPU box where /root/reference is absent).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no -march flags: plain SSE2 scalar IEEE f32, same
 * arithmetic as the reference built with g++ -O2; no FMA contraction, no x87 excess precision).
 *
 * Semantics followed (L/ = /root/reference/3rd_party/libmmd/include/mmd/):
 *   morph pass      L/motion/poser_impl.inl:328-346, :362-365, :384-386
 *   skinning        L/motion/poser_impl.inl:396-437
 *   matrix blend    L/util/math_impl.inl:924-963 (M*s, M+M), :1004-1023 (s*M), :1241-1259 (Lerp)
 *   mat * vec       L/util/math_impl.inl:1032-1045 (rotate / transform)
 *   retagging       L/model/model_impl.inl:406-452 (Model::Normalize)
 *   repack          /root/reference/main.cpp:50-54, :838-859
 *   morph tracks    L/motion/motion_impl.inl:382-424
 *   bone tracks     L/motion/motion_impl.inl:255-319, Bezier L/util/math_impl.inl:1379-1428, NLerp :1260-1282
 *   bone solve      L/motion/poser_impl.inl:29-109 (ctor), :142-326 (UpdateBoneTransform incl. append + CCD-IK)
 * Matrix convention: row vector, row-major float[16], translation in elements 12..14
 *   (L/util/math.inl:383-395).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#define MMDX_EPS_D 1e-7 /* L/util/math.inl:24 `#define mmd_math_const_eps 1e-7` (a double literal) */

enum { SKIN_BDEF1 = 0, SKIN_BDEF2 = 1, SKIN_BDEF4 = 2, SKIN_SDEF = 3 };
enum { MORPH_GROUP = 0, MORPH_VERTEX = 1 };

/* ---- Model::Normalize (model_impl.inl:406-452) -------------------------------------------- */
/* type[NV] int32, ids[NV][4] int64, w[NV][4]; parent[NB] int64 (-1 = none -> size_t(-1)).        */
void mmdx_oracle_normalize(uint32_t nv, int32_t *type, int64_t *ids, float *w,
                           const int64_t *parent) {
    for (uint32_t i = 0; i < nv; ++i) {
        int64_t *id = ids + 4 * (size_t)i;
        float weight = w[4 * (size_t)i];
        if (type[i] == SKIN_BDEF2) {
            if (weight == 0.0f) { id[0] = id[1]; type[i] = SKIN_BDEF1; }
            else if (weight == 1.0f) { type[i] = SKIN_BDEF1; }
        } else if (type[i] == SKIN_SDEF) {
            int64_t b0 = id[0], b1 = id[1];
            if (parent[b0] != b1 && parent[b1] != b0) {
                if (weight == 0.0f) { id[0] = id[1]; type[i] = SKIN_BDEF1; }
                else if (weight == 1.0f) { type[i] = SKIN_BDEF1; }
                else { type[i] = SKIN_BDEF2; }
            }
        }
    }
}

/* ---- morph pass (poser_impl.inl:328-346) --------------------------------------------------- */
typedef struct {
    const int32_t *morph_type;
    const uint32_t *morph_off;
    const uint32_t *morph_index;
    const float *morph_value; /* [E][3] */
    float *vimg;              /* [NV][3] */
} morph_ctx;

static void apply_morph(const morph_ctx *c, uint32_t index, float rate) {
    if ((double)rate < MMDX_EPS_D) return; /* float promoted to double, as in the reference */
    uint32_t b = c->morph_off[index], e = c->morph_off[index + 1];
    if (c->morph_type[index] == MORPH_GROUP) {
        for (uint32_t j = b; j < e; ++j)
            apply_morph(c, c->morph_index[j], c->morph_value[3 * (size_t)j] * rate);
    } else if (c->morph_type[index] == MORPH_VERTEX) {
        for (uint32_t j = b; j < e; ++j) {
            float *v = c->vimg + 3 * (size_t)c->morph_index[j];
            const float *o = c->morph_value + 3 * (size_t)j;
            float t0 = o[0] * rate, t1 = o[1] * rate, t2 = o[2] * rate;
            v[0] = v[0] + t0;
            v[1] = v[1] + t1;
            v[2] = v[2] + t2;
        }
    } /* bone morphs feed the (out-of-scope) bone solve; uv / material morphs are ignored */
}

/* vimg[NV][3] := 0, then every morph in index order (poser_impl.inl:362-365, :384-386). */
void mmdx_oracle_morph(uint32_t nv, uint32_t nm, const int32_t *morph_type,
                       const uint32_t *morph_off, const uint32_t *morph_index,
                       const float *morph_value, const float *rates, float *vimg) {
    morph_ctx c = {morph_type, morph_off, morph_index, morph_value, vimg};
    memset(vimg, 0, (size_t)nv * 12);
    for (uint32_t i = 0; i < nm; ++i) apply_morph(&c, i, rates[i]);
}

/* ---- skinning (poser_impl.inl:396-437) ------------------------------------------------------ */
/* Only columns 0..2 of the blended matrix are ever read by transform/rotate, so 12 of the 16
 * elements are blended; element k of row r, column j is S[4*r + j]. */
static void blend2(const float *a /*S[b1]*/, const float *b /*S[b0]*/, float l, float *m) {
    if (l < (float)MMDX_EPS_D) {
        for (int k = 0; k < 16; ++k) m[k] = a[k];
    } else if (l > (float)(1.0 - MMDX_EPS_D)) {
        for (int k = 0; k < 16; ++k) m[k] = b[k];
    } else {
        float s = 1.0f - l;
        for (int k = 0; k < 16; ++k) {
            float ta = s * a[k];
            float tb = l * b[k];
            m[k] = ta + tb;
        }
    }
}

static void blend4(const float *m0, const float *m1, const float *m2, const float *m3,
                   const float *w, float *m) {
    for (int k = 0; k < 16; ++k) {
        float t0 = m0[k] * w[0];
        float t1 = m1[k] * w[1];
        float t2 = m2[k] * w[2];
        float t3 = m3[k] * w[3];
        float s = t0 + t1;
        s = s + t2;
        s = s + t3;
        m[k] = s;
    }
}

void mmdx_oracle_skin(uint32_t nv, const float *pos, const float *nrm, const float *vimg,
                      const int32_t *type, const int64_t *ids, const float *w,
                      const float *palette /*[NB][16]*/, float *out_pos, float *out_nrm) {
    for (uint32_t i = 0; i < nv; ++i) {
        const float *p = pos + 3 * (size_t)i, *n = nrm + 3 * (size_t)i;
        const int64_t *id = ids + 4 * (size_t)i;
        const float *wt = w + 4 * (size_t)i;
        float c[3] = {p[0], p[1], p[2]};
        if (vimg) {
            const float *d = vimg + 3 * (size_t)i;
            c[0] = p[0] + d[0]; c[1] = p[1] + d[1]; c[2] = p[2] + d[2];
        }
        float mbuf[16];
        const float *m;
        if (type[i] == SKIN_BDEF1) {
            m = palette + 16 * (size_t)id[0];
        } else if (type[i] == SKIN_BDEF4) {
            blend4(palette + 16 * (size_t)id[0], palette + 16 * (size_t)id[1],
                   palette + 16 * (size_t)id[2], palette + 16 * (size_t)id[3], wt, mbuf);
            m = mbuf;
        } else { /* BDEF2, SDEF (true SDEF is commented out upstream) and every unknown tag */
            blend2(palette + 16 * (size_t)id[1], palette + 16 * (size_t)id[0], wt[0], mbuf);
            m = mbuf;
        }
        float *op = out_pos + 3 * (size_t)i, *on = out_nrm + 3 * (size_t)i;
        for (int j = 0; j < 3; ++j) {
            float a0 = c[0] * m[0 + j], a1 = c[1] * m[4 + j], a2 = c[2] * m[8 + j];
            float s = a0 + a1;
            s = s + a2;
            s = s + m[12 + j];
            op[j] = s;
            float b0 = n[0] * m[0 + j], b1 = n[1] * m[4 + j], b2 = n[2] * m[8 + j];
            float t = b0 + b1;
            t = t + b2;
            on[j] = t;
        }
    }
}

/* ---- repack to the viewer's 32-byte Vertex (main.cpp:50-54, :838-859) ------------------------ */
void mmdx_oracle_repack32(uint32_t nv, const float *pos, const float *nrm, const float *uv,
                          float pos_scale, float *out /*[NV][8]*/) {
    for (uint32_t i = 0; i < nv; ++i) {
        float *o = out + 8 * (size_t)i;
        o[0] = pos[3 * (size_t)i + 0] * pos_scale;
        o[1] = pos[3 * (size_t)i + 1] * pos_scale;
        o[2] = pos[3 * (size_t)i + 2] * pos_scale;
        o[3] = nrm[3 * (size_t)i + 0];
        o[4] = nrm[3 * (size_t)i + 1];
        o[5] = nrm[3 * (size_t)i + 2];
        o[6] = uv ? uv[2 * (size_t)i + 0] : 0.0f;
        o[7] = uv ? uv[2 * (size_t)i + 1] : 0.0f;
    }
}

/* ---- VMD morph track evaluation (Motion::GetMorphPose, L/motion/motion_impl.inl:382-424) ------- */
/* key_off[NM+1], frames/weights sorted ascending inside a morph; out[NI][NM].  A morph without keys
 * keeps 0 (SeekFrame never touches it after ResetPosing, L/motion/poser_impl.inl:539-542, :131-133). */
void mmdx_oracle_morph_tracks(uint32_t nm, const uint32_t *key_off, const uint32_t *frames,
                              const float *weights, uint32_t ni, const uint32_t *at, float *out) {
    for (uint32_t i = 0; i < ni; ++i) {
        const uint32_t frame = at[i];
        for (uint32_t m = 0; m < nm; ++m) {
            const uint32_t b = key_off[m], e = key_off[m + 1];
            float w = 0.0f;
            if (e > b) {
                if (frames[b] >= frame) {
                    w = weights[b];
                } else if (frames[e - 1] <= frame) {
                    w = weights[e - 1];
                } else {
                    uint32_t r = b;
                    while (frames[r] <= frame) ++r;      /* first key after `frame` (upper_bound) */
                    const uint32_t l = r - 1;
                    if (frames[l] == frame) {
                        w = weights[l];
                    } else {
                        const float bary = (float)(frame - frames[l]) / (float)(frames[r] - frames[l]);
                        const float a = weights[l] * (1 - bary);
                        const float c = weights[r] * bary;
                        w = a + c;
                    }
                }
            }
            out[(size_t)i * nm + m] = w;
        }
    }
}

/* ---- VMD bone track evaluation (Motion::GetBonePose, L/motion/motion_impl.inl:255-319) ---------- */
/* Interpolation curve of one channel: control bytes c[0],c[4],c[8],c[12] of a 16-byte block scaled by
 * 1.0f/127.0f (L/reader/vmd_reader_impl.inl:31-60), stored times 3 by Bezier::SetC, presampled at 32
 * points unless linear (L/util/math_impl.inl:1393-1428).  `abs` in the bisection is the float overload:
 * see the include-order note in oracle/ref_harness.cpp. */
typedef struct { int linear; float pre[32]; } curve_t;

static float curve_poly(float lm, float a, float b) {
    const float rm = 1.0f - lm;
    return lm * (rm * (rm * a + lm * b) + lm * lm);
}

static void curve_setup(const int8_t *c, curve_t *cv) {
    const float r = 1.0f / 127.0f;
    const float c0x = (c[0] * r) * 3.0f, c0y = (c[4] * r) * 3.0f;
    const float c1x = (c[8] * r) * 3.0f, c1y = (c[12] * r) * 3.0f;
    cv->linear = (c0x == c0y) && (c1x == c1y);
    if (cv->linear) return;
    for (size_t i = 0; i < 32; ++i) {
        const float x = i / 31.0f;
        float l = 0.0f, rr = 1.0f, lm = 0.0f;
        for (size_t it = 0; it < 32; ++it) {
            lm = (l + rr) * 0.5f;
            const float m = curve_poly(lm, c0x, c1x);
            const float d = m - x;
            if ((d < 0.0f ? -d : d) < 1e-7f) break;
            if (m > x) rr = lm; else l = lm;
        }
        cv->pre[i] = curve_poly(lm, c0y, c1y);
    }
}

static float curve_eval(const curve_t *cv, float x) {
    if (cv->linear) return x;
    x *= 31;
    const size_t ix = (size_t)x;
    const float r = x - ix;
    if (ix < 31) return (1.0f - r) * cv->pre[ix] + r * cv->pre[ix + 1];
    return cv->pre[31];
}

/* One bone track: frames[n] ascending, tr[n][3], rot[n][4], interp[n][64]; out = t.xyz, 0, q.xyzw.
 * n == 0: the pose ResetPosing leaves (zero translation, identity rotation). */
void mmdx_oracle_bone_pose(uint32_t n, const uint32_t *frames, const float *tr, const float *rot,
                           const int8_t *interp, uint32_t frame, float *out) {
    float t[3] = {0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    if (n) {
        uint32_t k = n;                                   /* key to copy verbatim, or n = interpolate */
        if (frames[0] >= frame) k = 0;
        else if (frames[n - 1] <= frame) k = n - 1;
        uint32_t r = 0;
        if (k == n) {
            while (frames[r] <= frame) ++r;               /* upper_bound */
            if (frames[r - 1] == frame) k = r - 1;
        }
        if (k < n) {
            memcpy(t, tr + 3 * (size_t)k, 12);
            memcpy(q, rot + 4 * (size_t)k, 16);
        } else {
            const uint32_t l = r - 1;
            const float bary = (float)(frame - frames[l]) / (float)(frames[r] - frames[l]);
            const float *lt = tr + 3 * (size_t)l, *rt = tr + 3 * (size_t)r;
            const float *lq = rot + 4 * (size_t)l, *rq = rot + 4 * (size_t)r;
            curve_t cv;
            float lambda;
            for (int c = 0; c < 3; ++c) {
                curve_setup(interp + 64 * (size_t)l + 16 * c, &cv);
                lambda = curve_eval(&cv, bary);
                t[c] = lt[c] * (1 - lambda) + rt[c] * lambda;
            }
            curve_setup(interp + 64 * (size_t)l + 48, &cv);
            lambda = curve_eval(&cv, bary);
            /* NLerp, L/util/math_impl.inl:1260-1282 */
            if (lambda < (float)MMDX_EPS_D) {
                memcpy(q, lq, 16);
            } else if (lambda > (1.0f - (float)MMDX_EPS_D)) {
                memcpy(q, rq, 16);
            } else {
                const float dot = lq[0] * rq[0] + lq[1] * rq[1] + lq[2] * rq[2] + lq[3] * rq[3];
                const float a = 1.0f - lambda;
                float v[4];
                for (int c = 0; c < 4; ++c) {
                    const float x = a * lq[c], y = lambda * rq[c];
                    v[c] = dot < 0.0f ? x - y : x + y;
                }
                const float norm = (float)sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
                const float inv = 1.0f / norm;
                for (int c = 0; c < 4; ++c) q[c] = v[c] * inv;
            }
        }
    }
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = 0.0f;
    memcpy(out + 4, q, 16);
}

/* ---- bone solve without IK / append (Poser::UpdateBoneTransform, L/motion/poser_impl.inl:142-166;
 * UpdateBoneSkinningMatrix :320-326; order :99-109, :500-510; reset :366-377) --------------------- */
static void mat_mul(const float *a, const float *b, float *r) {    /* L/util/math_impl.inl:984-1003 */
    float t[16];
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x)
            t[4 * y + x] = a[4 * y] * b[x] + a[4 * y + 1] * b[4 + x] + a[4 * y + 2] * b[8 + x] + a[4 * y + 3] * b[12 + x];
    memcpy(r, t, 64);
}

static void mat_identity(float *m) {
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* rest[NB][3], parent[NB] (outside [0,NB) = none), level[NB] or NULL, flags[NB] or NULL (only the
 * post-physics bit 0x1000 matters here), poses[NB][8]; out[NB][16].  `scratch` = NB*16 floats + NB
 * uint32 (local matrices, order).  Returns -1 if a bone has IK or append flags. */
int mmdx_oracle_bone_solve(uint32_t nb, const float *rest, const int64_t *parent, const int32_t *level,
                           const uint16_t *flags, const float *poses, float *out, void *scratch) {
    float *local = (float *)scratch;
    uint32_t *order = (uint32_t *)(local + 16 * (size_t)nb);
    uint32_t n = 0;
    for (uint32_t b = 0; b < nb; ++b)
        if (flags && (flags[b] & (0x0020 | 0x0100 | 0x0200))) return -1;
    for (int pass = 0; pass < 2; ++pass) {                /* pre-physics list, then post-physics list */
        const uint32_t first = n;
        for (uint32_t b = 0; b < nb; ++b)
            if ((int)((flags ? flags[b] : 0) >> 12 & 1) == pass) order[n++] = b;
        for (uint32_t i = first + 1; i < n; ++i) {        /* insertion sort by (size_t level, index) */
            const uint32_t b = order[i];
            const uint64_t lb = (uint64_t)(int64_t)(level ? level[b] : 0);
            uint32_t j = i;
            while (j > first) {
                const uint32_t c = order[j - 1];
                const uint64_t lc = (uint64_t)(int64_t)(level ? level[c] : 0);
                if (lc < lb || (lc == lb && c < b)) break;
                order[j] = c;
                --j;
            }
            order[j] = b;
        }
    }
    for (uint32_t b = 0; b < nb; ++b) mat_identity(local + 16 * (size_t)b);
    for (uint32_t s = 0; s < nb; ++s) {
        const uint32_t b = order[s];
        const float *t = poses + 8 * (size_t)b, *r = poses + 8 * (size_t)b + 4;
        const int has_parent = parent[b] >= 0 && (uint64_t)parent[b] < nb;
        /* total_rotation_ = morph_rotation_(identity) * rotation_, L/util/math_impl.inl:510-517 */
        const float ai = 0.0f, aj = 0.0f, ak = 0.0f, ae = 1.0f;
        const float qi = (ae * r[0] + ai * r[3] + aj * r[2]) - ak * r[1];
        const float qj = (ae * r[1] + aj * r[3] + ak * r[0]) - ai * r[2];
        const float qk = (ae * r[2] + ai * r[1] + ak * r[3]) - aj * r[0];
        const float qe = ae * r[3] - (ai * r[0] + aj * r[1] + ak * r[2]);
        float tt[3], off[3];
        for (int k = 0; k < 3; ++k) {
            tt[k] = 0.0f + t[k];
            off[k] = has_parent ? rest[3 * (size_t)b + k] - rest[3 * (size_t)parent[b] + k] : rest[3 * (size_t)b + k];
        }
        float *m = local + 16 * (size_t)b;
        /* Quaternion::ToRotateMatrix, L/util/math_impl.inl:540-563 */
        const float ii = qi * qi, jj = qj * qj, kk = qk * qk, ij = qi * qj, jk = qj * qk, ki = qi * qk;
        const float ie = qi * qe, je = qj * qe, ke = qk * qe;
        m[0] = 1.0f - 2.0f * (jj + kk); m[1] = 2.0f * (ij + ke); m[2] = 2.0f * (ki - je); m[3] = 0.0f;
        m[4] = 2.0f * (ij - ke); m[5] = 1.0f - 2.0f * (kk + ii); m[6] = 2.0f * (jk + ie); m[7] = 0.0f;
        m[8] = 2.0f * (ki + je); m[9] = 2.0f * (jk - ie); m[10] = 1.0f - 2.0f * (ii + jj); m[11] = 0.0f;
        m[12] = tt[0] + off[0]; m[13] = tt[1] + off[1]; m[14] = tt[2] + off[2]; m[15] = 1.0f;
        if (has_parent) mat_mul(m, local + 16 * (size_t)parent[b], m);
    }
    for (uint32_t b = 0; b < nb; ++b) {
        float g[16];
        mat_identity(g);
        g[12] = -rest[3 * (size_t)b]; g[13] = -rest[3 * (size_t)b + 1]; g[14] = -rest[3 * (size_t)b + 2];
        mat_mul(g, local + 16 * (size_t)b, out + 16 * (size_t)b);
    }
    return 0;
}

/* ---- full bone solve: append (inherit) bones and CCD-IK (Poser ctor L/motion/poser_impl.inl:47-97,
 * Poser::UpdateBoneTransform :142-310, PrePhysicsPosing reset :366-377, Pre/PostPhysicsPosing :384-394).
 * Transcendentals go through double libm and back to float exactly like L/util/math.inl:27-45.  Bone
 * morphs are not an input here (morph_rotation_ = identity, morph_translation_ = 0).
 * `abs` in the ctor's limit classification is the float overload (include-order note in ref_harness). */
typedef struct { float i, j, k, e; } quat_t;
typedef struct {
    quat_t total_rot, ik_rot, pre_ik_rot, morph_rot;
    float total_tr[3], morph_tr[3], local[16];
} bone_state_t;

typedef struct {
    uint32_t nb;
    const float *rest;
    const int64_t *parent, *append_parent, *ik_target, *ik_link_bone;
    const uint16_t *flags;
    const float *append_ratio, *ik_angle, *ik_link_lo, *ik_link_hi;
    const int32_t *ik_loop;
    const uint32_t *ik_link_off;
    const uint8_t *ik_link_limited, *is_ik_link;
    const float *poses;
    bone_state_t *st;
} solve_ctx;

enum { FIX_NONE = 0, FIX_X, FIX_Y, FIX_Z, FIX_ALL };
enum { ORDER_ZXY = 0, ORDER_XYZ, ORDER_YZX };

/* Optional trace of the transcendental calls of the bone solve (tools/archive/probes/rig_mismatch_probe.py re-evaluates them on
 * the device to tell a libm difference from anything else): records of 4 words -- function (0 sqrt, 1 sin, 2 cos,
 * 3 asin, 4 acos, 5 atan2), argument bits, second argument bits, result bits.  Not thread-safe; off by default. */
static uint32_t *g_trace;
static size_t g_trace_cap, g_trace_n;
void mmdx_oracle_trace_libm(uint32_t *records, size_t capacity) { g_trace = records; g_trace_cap = capacity; g_trace_n = 0; }
size_t mmdx_oracle_trace_count(void) { return g_trace_n; }
static float traced(uint32_t fn, float a, float b, float r) {
    if (g_trace) {
        if (g_trace_n < g_trace_cap) {
            uint32_t *rec = g_trace + 4 * g_trace_n;
            rec[0] = fn;
            memcpy(rec + 1, &a, 4); memcpy(rec + 2, &b, 4); memcpy(rec + 3, &r, 4);
        }
        ++g_trace_n;
    }
    return r;
}
static float f_sqrt(float x) { return traced(0, x, 0.0f, (float)sqrt((double)x)); }
static float f_sin(float x) { return traced(1, x, 0.0f, (float)sin((double)x)); }
static float f_cos(float x) { return traced(2, x, 0.0f, (float)cos((double)x)); }
static float f_asin(float x) { return traced(3, x, 0.0f, (float)asin((double)x)); }
static float f_acos(float x) { return traced(4, x, 0.0f, (float)acos((double)x)); }
static float f_atan2(float y, float x) { return traced(5, y, x, (float)atan2((double)y, (double)x)); }

static quat_t q_identity(void) { quat_t q = {0.0f, 0.0f, 0.0f, 1.0f}; return q; }
static quat_t q_mul(quat_t a, quat_t q) {                  /* L/util/math_impl.inl:510-517 */
    quat_t r;
    r.i = (a.e * q.i + a.i * q.e + a.j * q.k) - a.k * q.j;
    r.j = (a.e * q.j + a.j * q.e + a.k * q.i) - a.i * q.k;
    r.k = (a.e * q.k + a.i * q.j + a.k * q.e) - a.j * q.i;
    r.e = a.e * q.e - (a.i * q.i + a.j * q.j + a.k * q.k);
    return r;
}
static quat_t q_scale(quat_t a, float s) { quat_t r = {a.i * s, a.j * s, a.k * s, a.e * s}; return r; }
static quat_t q_add(quat_t a, quat_t b) { quat_t r = {a.i + b.i, a.j + b.j, a.k + b.k, a.e + b.e}; return r; }
static quat_t q_inverse(quat_t a) {                        /* :474-477 */
    const float n = 1.0f / (a.i * a.i + a.j * a.j + a.k * a.k + a.e * a.e);
    quat_t c = {-a.i, -a.j, -a.k, a.e};
    return q_scale(c, n);
}
static quat_t q_slerp_from_identity(quat_t b, float l) {   /* SLerp(Identity, b)[l], :1312-1337 */
    const quat_t a = q_identity();
    float comega = a.e * b.e + a.i * b.i + a.j * b.j + a.k * b.k;
    const int flip = comega < 0.0f;
    if (flip) comega = -comega;
    const float omega = f_acos(comega);
    if (omega > (float)MMDX_EPS_D) {
        const float rs = 1.0f / f_sin(omega);
        const float p = f_sin((1.0f - l) * omega) * rs;
        l = f_sin(l * omega) * rs;
        if (flip) l = -l;
        return q_add(q_scale(a, p), q_scale(b, l));
    }
    return a;
}
static void q_to_matrix(quat_t q, float *m) {              /* :540-563 */
    const float ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k, ij = q.i * q.j, jk = q.j * q.k, ki = q.i * q.k;
    const float ie = q.i * q.e, je = q.j * q.e, ke = q.k * q.e;
    m[0] = 1.0f - 2.0f * (jj + kk); m[1] = 2.0f * (ij + ke); m[2] = 2.0f * (ki - je); m[3] = 0.0f;
    m[4] = 2.0f * (ij - ke); m[5] = 1.0f - 2.0f * (kk + ii); m[6] = 2.0f * (jk + ie); m[7] = 0.0f;
    m[8] = 2.0f * (ki + je); m[9] = 2.0f * (jk - ie); m[10] = 1.0f - 2.0f * (ii + jj); m[11] = 0.0f;
    m[12] = m[13] = m[14] = 0.0f; m[15] = 1.0f;
}
static void v3_normalize(float *v) {                       /* :390-400 */
    const float n = 1.0f / f_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] = v[0] * n; v[1] = v[1] * n; v[2] = v[2] * n;
}
static float v3_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static quat_t axis_to_quat(const float *axis, float angle) {   /* :1047-1058 */
    const float norm = f_sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    if (norm < (float)MMDX_EPS_D) return q_identity();
    angle *= 0.5f;
    const float s = f_sin(angle) / norm;
    quat_t q = {s * axis[0], s * axis[1], s * axis[2], f_cos(angle)};
    return q;
}

/* Quaternion <-> Euler for the three orders the IK uses, :1059-1071 / :1110-1135 and :1137-1223 */
static void quat_to_euler(int order, quat_t q, float *r) {
    const float ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k;
    const float ei = q.e * q.i, ej = q.e * q.j, ek = q.e * q.k;
    const float ij = q.i * q.j, ik = q.i * q.k, jk = q.j * q.k;
    if (order == ORDER_ZXY) {
        r[0] = f_asin(2.0f * (ei + jk));
        r[1] = f_atan2(2.0f * (ej - ik), 1 - 2.0f * (ii + jj));
        r[2] = f_atan2(2.0f * (ek - ij), 1 - 2.0f * (ii + kk));
    } else if (order == ORDER_XYZ) {
        r[0] = f_atan2(2.0f * (ei - jk), 1 - 2.0f * (ii + jj));
        r[1] = f_asin(2.0f * (ej + ik));
        r[2] = f_atan2(2.0f * (ek - ij), 1 - 2.0f * (jj + kk));
    } else {
        r[0] = f_atan2(2.0f * (ei - jk), 1 - 2.0f * (ii + kk));
        r[1] = f_atan2(2.0f * (ej - ik), 1 - 2.0f * (jj + kk));
        r[2] = f_asin(2.0f * (ek + ij));
    }
}
static quat_t euler_to_quat(int order, const float *r) {
    const float cx = f_cos(r[0] * 0.5f), sx = f_sin(r[0] * 0.5f);
    const float cy = f_cos(r[1] * 0.5f), sy = f_sin(r[1] * 0.5f);
    const float cz = f_cos(r[2] * 0.5f), sz = f_sin(r[2] * 0.5f);
    quat_t q;
    if (order == ORDER_ZXY) {
        q.e = cx * cy * cz - sx * sy * sz; q.i = sx * cy * cz - cx * sy * sz;
        q.j = cx * sy * cz + sx * cy * sz; q.k = cx * cy * sz + sx * sy * cz;
    } else if (order == ORDER_XYZ) {
        q.e = cx * cy * cz - sx * sy * sz; q.i = sx * cy * cz + cx * sy * sz;
        q.j = cx * sy * cz - sx * cy * sz; q.k = sx * sy * cz + cx * cy * sz;
    } else {
        q.e = cx * cy * cz - sx * sy * sz; q.i = sx * cy * cz + cx * sy * sz;
        q.j = cx * sy * cz + sx * cy * sz; q.k = cx * cy * sz - sx * sy * cz;
    }
    return q;
}
static void limit_euler(float *e, const float *lo, const float *hi, int ikt) {   /* poser_impl.inl:178-194 */
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

static int has_parent(const solve_ctx *c, uint32_t b) { return c->parent[b] >= 0 && (uint64_t)c->parent[b] < c->nb; }

static void place(const solve_ctx *c, uint32_t b) {        /* rotation -> local matrix, then the parent product */
    bone_state_t *s = &c->st[b];
    q_to_matrix(s->total_rot, s->local);
    for (int k = 0; k < 3; ++k) {
        const float off = has_parent(c, b) ? c->rest[3 * (size_t)b + k] - c->rest[3 * (size_t)c->parent[b] + k]
                                           : c->rest[3 * (size_t)b + k];
        s->local[12 + k] = s->total_tr[k] + off;
    }
    if (has_parent(c, b)) mat_mul(s->local, c->st[c->parent[b]].local, s->local);
}

static void update_bone(const solve_ctx *c, uint32_t b, int solve_ik);

static void solve_ik_chain(const solve_ctx *c, uint32_t b) {
    bone_state_t *st = c->st;
    const uint32_t l0 = c->ik_link_off[b], n = c->ik_link_off[b + 1] - l0;
    const uint32_t target = (uint32_t)c->ik_target[b];
    uint32_t limit = c->ik_loop[b] < 0 || c->ik_loop[b] > 256 ? 256u : (uint32_t)c->ik_loop[b];   /* min(size_t(loop), 256) */
    float ik_pos[3], tgt_pos[3], err[3];
    for (uint32_t i = 0; i < n; ++i) st[c->ik_link_bone[l0 + i]].ik_rot = q_identity();
    memcpy(ik_pos, st[b].local + 12, 12);
    for (uint32_t i = 0; i < n; ++i) update_bone(c, (uint32_t)c->ik_link_bone[l0 + n - i - 1], 1);
    update_bone(c, target, 1);
    memcpy(tgt_pos, st[target].local + 12, 12);
    for (int k = 0; k < 3; ++k) err[k] = ik_pos[k] - tgt_pos[k];
    if (v3_dot(err, err) < (float)MMDX_EPS_D) return;
    const uint32_t ikt = limit / 2;
    for (uint32_t i = 0; i < limit; ++i) {
        for (uint32_t j = 0; j < n; ++j) {
            /* per-link constants of the Poser ctor (:63-90) */
            const int limited = c->ik_link_limited[l0 + j] != 0;
            float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            int order = ORDER_YZX, fix = FIX_NONE;
            if (limited) {
                const float *a = c->ik_link_lo + 3 * (size_t)(l0 + j), *z = c->ik_link_hi + 3 * (size_t)(l0 + j);
                for (int k = 0; k < 3; ++k) { lo[k] = z[k] < a[k] ? z[k] : a[k]; hi[k] = a[k] < z[k] ? z[k] : a[k]; }
                const double half_pi = 3.141592653589793238462643383279502884 * 0.5f;
                if (lo[0] > -half_pi && hi[0] < half_pi) order = ORDER_ZXY;
                else if (lo[1] > -half_pi && hi[1] < half_pi) order = ORDER_XYZ;
                const int zx = fabsf(lo[0]) < 1e-7f && fabsf(hi[0]) < 1e-7f;
                const int zy = fabsf(lo[1]) < 1e-7f && fabsf(hi[1]) < 1e-7f;
                const int zz = fabsf(lo[2]) < 1e-7f && fabsf(hi[2]) < 1e-7f;
                if (zx && zy && zz) fix = FIX_ALL;
                else if (zy && zz) fix = FIX_X;
                else if (zx && zz) fix = FIX_Y;
                else if (zx && zy) fix = FIX_Z;
            }
            if (fix == FIX_ALL) continue;
            const uint32_t lb = (uint32_t)c->ik_link_bone[l0 + j];
            bone_state_t *li = &st[lb];
            float tdir[3], idir[3], axis[3];
            for (int k = 0; k < 3; ++k) { tdir[k] = li->local[12 + k] - tgt_pos[k]; idir[k] = li->local[12 + k] - ik_pos[k]; }
            v3_normalize(tdir);
            v3_normalize(idir);
            axis[0] = tdir[1] * idir[2] - tdir[2] * idir[1];
            axis[1] = tdir[2] * idir[0] - tdir[0] * idir[2];
            axis[2] = tdir[0] * idir[1] - tdir[1] * idir[0];
            for (int k = 0; k < 3; ++k) if (fabsf(axis[k]) < 1e-7f) axis[k] = 1e-7f;
            float loc[16];
            if (has_parent(c, lb)) memcpy(loc, st[c->parent[lb]].local, 64); else mat_identity(loc);
            if (limited && fix != FIX_NONE && i < ikt) {
                const float d = v3_dot(axis, loc + 4 * (fix - FIX_X));
                axis[0] = axis[1] = axis[2] = 0.0f;
                axis[fix - FIX_X] = d >= 0.0f ? 1.0f : -1.0f;
            } else {                                           /* rotate(axis, loc.Transpose()) then Normalize */
                float r[3];
                for (int x = 0; x < 3; ++x) r[x] = axis[0] * loc[4 * x + 0] + axis[1] * loc[4 * x + 1] + axis[2] * loc[4 * x + 2];
                memcpy(axis, r, 12);
                v3_normalize(axis);
            }
            float dot = v3_dot(tdir, idir);
            dot = dot < -1.0f ? -1.0f : dot;                   /* clamp = min(max(x, lo), hi), math.inl:45-47 */
            dot = 1.0f < dot ? 1.0f : dot;
            const float ac = f_acos(dot), cap = c->ik_angle[b] * (j + 1);
            const float angle = cap < ac ? cap : ac;
            li->ik_rot = q_mul(axis_to_quat(axis, angle), li->ik_rot);
            if (limited) {
                quat_t lr = q_mul(li->ik_rot, li->pre_ik_rot);
                float e[3];
                quat_to_euler(order, lr, e);
                limit_euler(e, lo, hi, i < ikt);
                lr = euler_to_quat(order, e);
                li->ik_rot = q_mul(lr, q_inverse(li->pre_ik_rot));
            }
            for (uint32_t k = 0; k <= j; ++k) {
                const uint32_t bb = (uint32_t)c->ik_link_bone[l0 + j - k];
                st[bb].total_rot = q_mul(st[bb].ik_rot, st[bb].pre_ik_rot);
                place(c, bb);
            }
            update_bone(c, target, 1);
            memcpy(tgt_pos, st[target].local + 12, 12);
        }
        for (int k = 0; k < 3; ++k) err[k] = ik_pos[k] - tgt_pos[k];
        if (v3_dot(err, err) < (float)MMDX_EPS_D) return;
#ifdef MMDX_IK_CYCLE_STATS      /* diagnostic build only (tools/archive/probes/ik_cycle_probe.py): when does the chain's state start to repeat? */
        if (limit >= 200) {
            static unsigned long long hist[6];
            static int first[2][5];
            unsigned long long h = 1469598103934665603ull;
            for (uint32_t q = 0; q < n; ++q) {
                const unsigned char *pb = (const unsigned char *)&st[c->ik_link_bone[l0 + q]].ik_rot;
                for (size_t z = 0; z < sizeof(quat_t); ++z) { h ^= pb[z]; h *= 1099511628211ull; }
            }
            if (i == 0) { for (int a = 0; a < 2; ++a) for (int p2 = 0; p2 < 5; ++p2) first[a][p2] = -1; }
            const int half = i >= ikt;
            for (int per = 1; per <= 4; ++per)
                if (i >= (uint32_t)per && (half ? i - per >= ikt : 1) && hist[(i - per) % 6] == h && first[half][per] < 0) first[half][per] = (int)i;
            hist[i % 6] = h;
            if (i + 1 == limit)
                fprintf(stderr, "ikcycle half1 p1 %d p2 %d p3 %d p4 %d | half2 p1 %d p2 %d p3 %d p4 %d\n", first[0][1], first[0][2], first[0][3],
                        first[0][4], first[1][1], first[1][2], first[1][3], first[1][4]);
        }
#endif
    }
}

static void update_bone(const solve_ctx *c, uint32_t b, int solve_ik) {
    bone_state_t *s = &c->st[b];
    const float *t = c->poses + 8 * (size_t)b;
    const quat_t rot = {t[4], t[5], t[6], t[7]};
    const uint16_t f = c->flags ? c->flags[b] : 0;
    s->total_rot = q_mul(s->morph_rot, rot);
    for (int k = 0; k < 3; ++k) s->total_tr[k] = s->morph_tr[k] + t[k];
    if ((f & 0x0300) && c->append_parent[b] >= 0 && (uint64_t)c->append_parent[b] < c->nb) {
        const bone_state_t *ap = &c->st[c->append_parent[b]];
        if (f & 0x0100) s->total_rot = q_mul(s->total_rot, q_slerp_from_identity(ap->total_rot, c->append_ratio[b]));
        if (f & 0x0200) for (int k = 0; k < 3; ++k) s->total_tr[k] = s->total_tr[k] + c->append_ratio[b] * ap->total_tr[k];
    }
    if (c->is_ik_link[b]) {
        s->pre_ik_rot = s->total_rot;
        s->total_rot = q_mul(s->ik_rot, s->total_rot);
    }
    place(c, b);
    if (solve_ik && (f & 0x0020)) solve_ik_chain(c, b);
}

/* Bone morphs: Poser::UpdateMorphTransform restricted to its effect on the bones
 * (L/motion/poser_impl.inl:328-339 group recursion, :347-354 bone morphs). */
typedef struct {
    uint32_t nm;
    const int32_t *type;
    const uint32_t *off, *index;
    const float *value, *rotation;
    bone_state_t *st;
} bmorph_ctx;

static void apply_bone_morph(const bmorph_ctx *c, uint32_t m, float rate) {
    if (rate < (float)MMDX_EPS_D) return;
    if (c->type[m] == 0) {
        for (uint32_t e = c->off[m]; e < c->off[m + 1]; ++e) apply_bone_morph(c, c->index[e], c->value[3 * (size_t)e] * rate);
    } else if (c->type[m] == 2) {
        for (uint32_t e = c->off[m]; e < c->off[m + 1]; ++e) {
            bone_state_t *s = &c->st[c->index[e]];
            for (int k = 0; k < 3; ++k) s->morph_tr[k] = s->morph_tr[k] + c->value[3 * (size_t)e + k] * rate;
            quat_t r = q_identity();
            if (c->rotation) { r.i = c->rotation[4 * (size_t)e]; r.j = c->rotation[4 * (size_t)e + 1];
                               r.k = c->rotation[4 * (size_t)e + 2]; r.e = c->rotation[4 * (size_t)e + 3]; }
            s->morph_rot = q_mul(s->morph_rot, q_slerp_from_identity(r, rate));
        }
    }
}

/* Everything as flat arrays (NULL allowed where no bone uses it): append_parent / append_ratio [NB];
 * ik_target [NB], ik_loop [NB], ik_angle [NB], ik_link_off [NB+1], ik_link_bone / ik_link_limited [L],
 * ik_link_lo / ik_link_hi [L][3]; morph table (nm, type, off, index, value[E][3], rotation[E][4] or NULL) with
 * rates[nm] or nm == 0.  scratch = NB * (sizeof(bone_state_t) + 5) bytes.  Returns -1 when an index is out of range. */
/* Matrix4x4<T>::Inverse(), L/util/math_impl.inl:822-897: Gauss-Jordan on [M | I] with scaled partial pivoting
 * (row i's scale = its largest |element|; a zero row or a zero last pivot gives the ZERO matrix), forward
 * elimination, then the upper triangle is cleared column by column and every row divided by its pivot. */
void mmdx_oracle_matrix_inverse(const float *in, float *out) {
    float s[4][8], scale[4];
    int row[4] = {0, 1, 2, 3};                              /* the reference swaps row pointers */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { s[i][j] = in[4 * i + j]; s[i][j + 4] = i == j ? 1.0f : 0.0f; }
    for (int i = 0; i < 4; ++i) {
        scale[i] = fabsf(s[i][0]);
        for (int j = 1; j < 4; ++j) { const float x = fabsf(s[i][j]); if (x > scale[i]) scale[i] = x; }
        if (scale[i] == 0) { memset(out, 0, 64); return; }
    }
    for (int i = 0; i < 4; ++i) {
        int pivot = i;
        float best = fabsf(s[row[i]][i] / scale[i]);
        for (int p = i + 1; p < 4; ++p) {
            const float x = fabsf(s[row[p]][i] / scale[p]);
            if (x > best) { best = x; pivot = p; }
        }
        if (pivot != i) {
            const int r = row[i]; row[i] = row[pivot]; row[pivot] = r;
            const float c = scale[i]; scale[i] = scale[pivot]; scale[pivot] = c;
        }
        for (int j = i + 1; j < 4; ++j) {
            const float m = s[row[j]][i] / s[row[i]][i];
            s[row[j]][i] = 0.0f;
            for (int jj = i + 1; jj < 8; ++jj) s[row[j]][jj] -= m * s[row[i]][jj];
        }
    }
    if (s[row[3]][3] == 0) { memset(out, 0, 64); return; }
    for (int i = 1; i < 4; ++i)
        for (int j = 0; j < i; ++j) {
            const float m = s[row[j]][i] / s[row[i]][i];
            for (int jj = j + 1; jj < 8; ++jj) s[row[j]][jj] -= m * s[row[i]][jj];
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = s[row[i]][j + 4] / s[row[i]][i];
}

/* What BulletPhysicsReactor::React leaves in the poser between the two bone lists
 * (L/../mmd-bullet/mmd-bullet_impl.inl:312-326): Synchronize (:34-40) -- the body's transform becomes the
 * bone's skinning matrix -- for every moved bone, then Fix (:42-56) for the strict ones, in list order. */
static void physics_fix(const solve_ctx *c, uint32_t b, float *skin /* [16] of bone b, in / out */) {
    bone_state_t *s = &c->st[b];
    float g[16], pl[16], inv[16];
    mat_identity(g);                                        /* global_offset_matrix_inv_: + rest position */
    g[12] = c->rest[3 * (size_t)b]; g[13] = c->rest[3 * (size_t)b + 1]; g[14] = c->rest[3 * (size_t)b + 2];
    mat_mul(g, skin, s->local);
    if (has_parent(c, b)) {
        memcpy(pl, c->st[c->parent[b]].local, 64);
        mmdx_oracle_matrix_inverse(pl, inv);
        mat_mul(s->local, inv, s->local);
    }
    for (int k = 0; k < 3; ++k) {
        const float off = has_parent(c, b) ? c->rest[3 * (size_t)b + k] - c->rest[3 * (size_t)c->parent[b] + k]
                                           : c->rest[3 * (size_t)b + k];
        s->local[12 + k] = s->total_tr[k] + off;
    }
    if (has_parent(c, b)) mat_mul(s->local, pl, s->local);
    mat_identity(g);
    g[12] = -c->rest[3 * (size_t)b]; g[13] = -c->rest[3 * (size_t)b + 1]; g[14] = -c->rest[3 * (size_t)b + 2];
    mat_mul(g, s->local, skin);
}

static int bone_solve_impl(uint32_t nb, const float *rest, const int64_t *parent, const int32_t *level,
                           const uint16_t *flags, const int64_t *append_parent, const float *append_ratio,
                           const int64_t *ik_target, const int32_t *ik_loop, const float *ik_angle,
                           const uint32_t *ik_link_off, const int64_t *ik_link_bone,
                           const uint8_t *ik_link_limited, const float *ik_link_lo, const float *ik_link_hi,
                           uint32_t nm, const int32_t *morph_type, const uint32_t *morph_off,
                           const uint32_t *morph_index, const float *morph_value, const float *morph_rotation,
                           const float *rates, const float *poses, float *out, void *scratch,
                           uint32_t n_over, const int64_t *over_bone, const uint8_t *over_strict,
                           const float *over_skin, float *pre_out);

int mmdx_oracle_bone_solve_full(uint32_t nb, const float *rest, const int64_t *parent, const int32_t *level,
                                const uint16_t *flags, const int64_t *append_parent, const float *append_ratio,
                                const int64_t *ik_target, const int32_t *ik_loop, const float *ik_angle,
                                const uint32_t *ik_link_off, const int64_t *ik_link_bone,
                                const uint8_t *ik_link_limited, const float *ik_link_lo, const float *ik_link_hi,
                                uint32_t nm, const int32_t *morph_type, const uint32_t *morph_off,
                                const uint32_t *morph_index, const float *morph_value, const float *morph_rotation,
                                const float *rates,
                                const float *poses, float *out, void *scratch) {
    return bone_solve_impl(nb, rest, parent, level, flags, append_parent, append_ratio, ik_target, ik_loop, ik_angle,
                           ik_link_off, ik_link_bone, ik_link_limited, ik_link_lo, ik_link_hi, nm, morph_type, morph_off,
                           morph_index, morph_value, morph_rotation, rates, poses, out, scratch, 0, NULL, NULL, NULL, NULL);
}

/* The same frame with the physics reactor's writes between the two bone lists: over_skin [n_over][16] replace the
 * skinning matrices of over_bone[] after the pre-physics list (out must have been... it is written here), Fix()
 * for over_strict[k] != 0.  pre_out (may be NULL, [NB][16]) receives the palette as it stands after the
 * pre-physics list (rows of post-physics bones: identity-derived values of a fresh poser are NOT modelled; they
 * are left untouched). */
int mmdx_oracle_bone_solve_physics(uint32_t nb, const float *rest, const int64_t *parent, const int32_t *level,
                                   const uint16_t *flags, const int64_t *append_parent, const float *append_ratio,
                                   const int64_t *ik_target, const int32_t *ik_loop, const float *ik_angle,
                                   const uint32_t *ik_link_off, const int64_t *ik_link_bone,
                                   const uint8_t *ik_link_limited, const float *ik_link_lo, const float *ik_link_hi,
                                   uint32_t nm, const int32_t *morph_type, const uint32_t *morph_off,
                                   const uint32_t *morph_index, const float *morph_value, const float *morph_rotation,
                                   const float *rates, const float *poses, float *out, void *scratch,
                                   uint32_t n_over, const int64_t *over_bone, const uint8_t *over_strict,
                                   const float *over_skin, float *pre_out) {
    return bone_solve_impl(nb, rest, parent, level, flags, append_parent, append_ratio, ik_target, ik_loop, ik_angle,
                           ik_link_off, ik_link_bone, ik_link_limited, ik_link_lo, ik_link_hi, nm, morph_type, morph_off,
                           morph_index, morph_value, morph_rotation, rates, poses, out, scratch, n_over, over_bone,
                           over_strict, over_skin, pre_out);
}

static int bone_solve_impl(uint32_t nb, const float *rest, const int64_t *parent, const int32_t *level,
                           const uint16_t *flags, const int64_t *append_parent, const float *append_ratio,
                           const int64_t *ik_target, const int32_t *ik_loop, const float *ik_angle,
                           const uint32_t *ik_link_off, const int64_t *ik_link_bone,
                           const uint8_t *ik_link_limited, const float *ik_link_lo, const float *ik_link_hi,
                           uint32_t nm, const int32_t *morph_type, const uint32_t *morph_off,
                           const uint32_t *morph_index, const float *morph_value, const float *morph_rotation,
                           const float *rates, const float *poses, float *out, void *scratch,
                           uint32_t n_over, const int64_t *over_bone, const uint8_t *over_strict,
                           const float *over_skin, float *pre_out) {
    bone_state_t *st = (bone_state_t *)scratch;
    uint32_t *order = (uint32_t *)(st + nb);
    uint8_t *is_link = (uint8_t *)(order + nb);
    uint32_t n = 0, n_pre = 0;
    memset(is_link, 0, nb);
    for (uint32_t b = 0; b < nb; ++b) {
        if (!(flags && (flags[b] & 0x0020))) continue;
        /* a link or target that is itself an IK bone is legal: update_bone recurses like UpdateBoneTransform does
         * (poser_impl.inl:196-206); cycles (endless recursion upstream) are the caller's to avoid */
        if (ik_target[b] < 0 || (uint64_t)ik_target[b] >= nb) return -1;
        for (uint32_t l = ik_link_off[b]; l < ik_link_off[b + 1]; ++l) {
            if (ik_link_bone[l] < 0 || (uint64_t)ik_link_bone[l] >= nb) return -1;
            is_link[ik_link_bone[l]] = 1;
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t first = n;
        for (uint32_t b = 0; b < nb; ++b)
            if ((int)((flags ? flags[b] : 0) >> 12 & 1) == pass) order[n++] = b;
        for (uint32_t i = first + 1; i < n; ++i) {
            const uint32_t b = order[i];
            const uint64_t lb = (uint64_t)(int64_t)(level ? level[b] : 0);
            uint32_t j = i;
            while (j > first) {
                const uint32_t cb = order[j - 1];
                const uint64_t lc = (uint64_t)(int64_t)(level ? level[cb] : 0);
                if (lc < lb || (lc == lb && cb < b)) break;
                order[j] = cb;
                --j;
            }
            order[j] = b;
        }
        if (pass == 0) n_pre = n;
    }
    for (uint32_t b = 0; b < nb; ++b) {
        st[b].total_rot = st[b].ik_rot = st[b].pre_ik_rot = st[b].morph_rot = q_identity();
        st[b].total_tr[0] = st[b].total_tr[1] = st[b].total_tr[2] = 0.0f;
        st[b].morph_tr[0] = st[b].morph_tr[1] = st[b].morph_tr[2] = 0.0f;
        mat_identity(st[b].local);
    }
    if (nm) {
        const bmorph_ctx mc = {nm, morph_type, morph_off, morph_index, morph_value, morph_rotation, st};
        for (uint32_t m = 0; m < nm; ++m) apply_bone_morph(&mc, m, rates[m]);
    }
    const solve_ctx c = {nb, rest, parent, append_parent, ik_target, ik_link_bone, flags, append_ratio, ik_angle,
                         ik_link_lo, ik_link_hi, ik_loop, ik_link_off, ik_link_limited, is_link, poses, st};
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t s0 = pass ? n_pre : 0, s1 = pass ? nb : n_pre;
        for (uint32_t s = s0; s < s1; ++s) update_bone(&c, order[s], 1);
        for (uint32_t s = s0; s < s1; ++s) {                /* UpdateBoneSkinningMatrix of this list */
            const uint32_t b = order[s];
            float g[16];
            mat_identity(g);
            g[12] = -rest[3 * (size_t)b]; g[13] = -rest[3 * (size_t)b + 1]; g[14] = -rest[3 * (size_t)b + 2];
            mat_mul(g, st[b].local, out + 16 * (size_t)b);
        }
        if (pass == 0) {
            if (pre_out) memcpy(pre_out, out, (size_t)nb * 64);
            for (uint32_t k = 0; k < n_over; ++k) {
                if (over_bone[k] < 0 || (uint64_t)over_bone[k] >= nb) return -1;
                memcpy(out + 16 * (size_t)over_bone[k], over_skin + 16 * (size_t)k, 64);          /* Synchronize */
            }
            for (uint32_t k = 0; k < n_over; ++k)
                if (over_strict && over_strict[k]) physics_fix(&c, (uint32_t)over_bone[k], out + 16 * (size_t)over_bone[k]);
        }
    }
    return 0;
}

/* ---- cpu_baseline timing helpers (kind "port"; seconds, single thread) ---------------------- */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Crowd step: one shared morph pass, then per instance one skinning pass into out scratch. */
double mmdx_oracle_time_crowd(uint32_t nv, uint32_t nb, uint32_t nm, const float *pos,
                              const float *nrm, const int32_t *type, const int64_t *ids,
                              const float *w, const int32_t *morph_type, const uint32_t *morph_off,
                              const uint32_t *morph_index, const float *morph_value,
                              const float *rates, uint32_t instances,
                              const float *palettes /*[instances][NB][16]*/, float *vimg,
                              float *out_pos, float *out_nrm) {
    double t0 = now_s();
    mmdx_oracle_morph(nv, nm, morph_type, morph_off, morph_index, morph_value, rates, vimg);
    for (uint32_t i = 0; i < instances; ++i)
        mmdx_oracle_skin(nv, pos, nrm, vimg, type, ids, w, palettes + (size_t)i * nb * 16, out_pos,
                         out_nrm);
    return now_s() - t0;
}
