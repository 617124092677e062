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
