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
 * Matrix convention: row vector, row-major float[16], translation in elements 12..14
 *   (L/util/math.inl:383-395).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
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
