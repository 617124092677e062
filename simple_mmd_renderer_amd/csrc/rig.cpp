// rig.cpp -- host-side compilation of bone tracks and of the skeleton (no HIP in this file, so the
// sanitizer drivers can link it).  The device kernels are in rig_kernels.hip, the C ABI in rig_api.cpp.
//
// Reference semantics followed (L/ = 3rd_party/libmmd/include/mmd/):
//   control points: byte * (1.0f/127.0f), bytes [0],[4],[8],[12] of each 16-byte block
//                                                      L/reader/vmd_reader_impl.inl:31-60
//   Bezier<float,32>::SetC stores the points times 3, a curve is linear when x == y for both points;
//   otherwise 32 samples at i/31 are taken by a 32-step float bisection on the x polynomial
//                                                      L/util/math_impl.inl:1393-1428
//   evaluation order of the bone solve: pre-physics bones then post-physics bones, each sorted by
//   (transform level as size_t, index)                 L/motion/poser_impl.inl:99-109, :500-510
//   local offset = rest - parent's rest (or rest), global offset = translate(-rest)
//                                                      L/motion/poser_impl.inl:36-45
//
// Platform note on the bisection's `abs(m-x)` (math_impl.inl:1417): the call is unqualified, so which
// overload it finds depends on the headers seen before <mmd/mmd.hxx>.  The viewer includes sokol and
// imgui first (main.cpp:10-22; they pull in <math.h>/<stdlib.h>), which makes ::abs(float) visible --
// the floating-point test below.  A translation unit that includes mmd.hxx first binds ::abs(int) under
// g++ and every curve degenerates to a constant; oracle/ref_harness.cpp therefore mirrors the viewer's
// include order.
#include "rig.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

namespace mmdx {

bool presample_curve(int8_t x0, int8_t y0, int8_t x1, int8_t y1, float out[kCurveSamples]) {
    const float r = 1.0f / 127.0f;
    const float c0x = (float(int(x0)) * r) * 3.0f, c0y = (float(int(y0)) * r) * 3.0f;
    const float c1x = (float(int(x1)) * r) * 3.0f, c1y = (float(int(y1)) * r) * 3.0f;
    if (c0x == c0y && c1x == c1y) return false;
    for (uint32_t i = 0; i < kCurveSamples; ++i) {
        const float x = float(i) / float(kCurveSamples - 1);
        float lo = 0.0f, hi = 1.0f, m, lm = 0.0f, rm;
        for (int it = 0; it < 32; ++it) {
            lm = (lo + hi) * 0.5f;
            rm = 1.0f - lm;
            m = lm * (rm * (rm * c0x + lm * c1x) + lm * lm);
            if (std::fabs(m - x) < 1e-7f) break;
            if (m > x) hi = lm; else lo = lm;
        }
        rm = 1.0f - lm;
        out[i] = lm * (rm * (rm * c0y + lm * c1y) + lm * lm);
    }
    return true;
}

void build_bone_motion(const std::vector<std::string> &track_names, const std::vector<uint32_t> &track_off,
                       const mmdx_vmd_bone_key *keys, uint32_t n_bones, const char *const *bone_names,
                       BoneMotionHost &out) {
    std::map<std::string, uint32_t> track_of;
    for (uint32_t t = 0; t < track_names.size(); ++t) track_of.emplace(track_names[t], t);
    std::map<uint32_t, uint32_t> curve_of;   // packed control bytes -> table id
    out = BoneMotionHost();
    out.nb = n_bones;
    out.key_off.push_back(0);
    float table[kCurveSamples];
    for (uint32_t b = 0; b < n_bones; ++b) {
        auto it = bone_names[b] ? track_of.find(bone_names[b]) : track_of.end();
        if (it != track_of.end()) {
            ++out.n_mapped;
            for (uint32_t k = track_off[it->second]; k < track_off[it->second + 1]; ++k) {
                const mmdx_vmd_bone_key &key = keys[k];
                out.key_frame.push_back(key.frame);
                out.key_tr.insert(out.key_tr.end(), {key.translation[0], key.translation[1], key.translation[2], 0.0f});
                out.key_rot.insert(out.key_rot.end(), key.rotation, key.rotation + 4);
                for (int ch = 0; ch < 4; ++ch) {
                    const int8_t *c = key.interpolation + 16 * ch;
                    uint32_t packed;
                    const int8_t pts[4] = {c[0], c[4], c[8], c[12]};
                    std::memcpy(&packed, pts, 4);
                    auto cit = curve_of.find(packed);
                    if (cit == curve_of.end()) {
                        uint32_t id = kLinearCurve;
                        if (presample_curve(pts[0], pts[1], pts[2], pts[3], table)) {
                            id = uint32_t(out.lut.size() / kCurveSamples);
                            out.lut.insert(out.lut.end(), table, table + kCurveSamples);
                        }
                        cit = curve_of.emplace(packed, id).first;
                    }
                    out.key_curve.push_back(cit->second);
                }
            }
        }
        out.key_off.push_back(uint32_t(out.key_frame.size()));
    }
}

void solve_event_sets(const SkeletonPlan &plan, uint32_t bone, std::vector<uint32_t> &reads, std::vector<uint32_t> &writes) {
    reads.clear();
    writes.clear();
    // one bone's transform (rig_kernels.hip transform_at): reads the parent's local matrix and, for an append
    // bone, the append parent's totals; writes every field of its own state
    auto transform = [&](uint32_t x) {
        const BoneRec &r = plan.bones[x];
        if (r.parent >= 0) reads.push_back(uint32_t(r.parent));
        if (r.bits & (kBoneAppendRot | kBoneAppendTr)) reads.push_back(uint32_t(r.append_parent));
        writes.push_back(x);
    };
    // UpdateBoneTransform of `x`: its transform and, for an IK bone, the CCD loop, which re-transforms its links and
    // target -- recursively where one of those is itself an IK bone (build_skeleton bounds the depth, no cycles)
    std::function<void(uint32_t)> update = [&](uint32_t x) {
        transform(x);
        if (plan.bones[x].bits & kBoneHasIk) {
            const IkRec &ik = plan.iks[plan.bones[x].ik];
            for (uint32_t j = 0; j < ik.nlinks; ++j) update(plan.links[ik.link0 + j].bone);
            update(ik.target);
        }
    };
    update(bone);
}

namespace {

// Cut the evaluation sequence into rounds.  An event depends on every earlier event that wrote a bone it
// reads or writes, or read a bone it writes; list scheduling over that graph, with two rules that keep the
// expensive events together: bone evaluations without IK go first whenever one is ready, and IK solves are
// issued only when nothing else is -- so independent chains (left / right leg ...) share a round instead of
// each paying a round of its own.  The two lists (pre- / post-physics) are scheduled separately: the
// skinning matrices of the first are written out in between.
void schedule_rounds(SkeletonPlan &pl) {
    pl.windows = 0;
    if (pl.n_fast) {
        const size_t per_window = size_t(kSolveInstances) * (pl.fast_slots * kSerialStateFloats + 1) * sizeof(float);
        pl.windows = uint32_t(std::min<size_t>({size_t(kSolveSlots), size_t(pl.n_fast), std::max<size_t>(kSolveLdsBudget / per_window, 1)}));
    }
    std::vector<uint32_t> reads, writes;
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const uint32_t s0 = pass ? pl.n_pre : 0, n = pass ? pl.nb - pl.n_pre : pl.n_pre;
        std::vector<std::vector<uint32_t>> succ(n);
        std::vector<uint32_t> indeg(n, 0);
        std::vector<int64_t> last_writer(pl.nb, -1);
        std::vector<std::vector<uint32_t>> readers(pl.nb);      // since the last write
        std::vector<uint32_t> deps;
        for (uint32_t e = 0; e < n; ++e) {
            solve_event_sets(pl, pl.order[s0 + e], reads, writes);
            deps.clear();
            for (uint32_t x : reads) if (last_writer[x] >= 0) deps.push_back(uint32_t(last_writer[x]));
            for (uint32_t x : writes) {
                if (last_writer[x] >= 0) deps.push_back(uint32_t(last_writer[x]));
                deps.insert(deps.end(), readers[x].begin(), readers[x].end());
            }
            std::sort(deps.begin(), deps.end());
            deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
            for (uint32_t dep : deps) {
                if (dep == e) continue;
                succ[dep].push_back(e);
                ++indeg[e];
            }
            for (uint32_t x : writes) { last_writer[x] = e; readers[x].clear(); }
            for (uint32_t x : reads) if (last_writer[x] != int64_t(e)) readers[x].push_back(e);
        }
        // ready lists, each kept in sequence order: 0 = no IK, 1 = IK on the HBM state, 2 = IK on an LDS window
        std::vector<uint32_t> ready[3], round;
        auto kind = [&](uint32_t e) -> int {
            const BoneRec &r = pl.bones[pl.order[s0 + e]];
            return (r.bits & kBoneHasIk) ? (pl.iks[r.ik].fast ? 2 : 1) : 0;
        };
        for (uint32_t e = 0; e < n; ++e) if (!indeg[e]) ready[kind(e)].push_back(e);
        auto take = [&](int k, size_t limit) {
            auto &q = ready[k];
            const size_t c = std::min(q.size(), limit);
            round.insert(round.end(), q.begin(), q.begin() + c);
            q.erase(q.begin(), q.begin() + c);
        };
        // MMDX_SOLVE_SEQUENTIAL=1: one event per round in sequence order (debugging aid: A/B against the schedule)
        const char *seq_env = std::getenv("MMDX_SOLVE_SEQUENTIAL");
        if (seq_env && seq_env[0] == '1') {
            for (uint32_t e = 0; e < n; ++e) {
                pl.rounds.push_back({uint32_t(pl.events.size()), 1u});
                pl.events.push_back(pl.order[s0 + e]);
            }
            if (!pass) pl.n_rounds_pre = uint32_t(pl.rounds.size());
            continue;
        }
        uint32_t done = 0;
        while (done < n) {
            round.clear();
            if (!ready[0].empty()) {
                take(0, kSolveSlots);
            } else {
                take(2, pl.windows);                          // the window owners sit in the first slots
                take(1, kSolveSlots - round.size());
            }
            pl.rounds.push_back({uint32_t(pl.events.size()), uint32_t(round.size())});
            for (uint32_t e : round) pl.events.push_back(pl.order[s0 + e]);
            done += uint32_t(round.size());
            for (uint32_t e : round)
                for (uint32_t nx : succ[e])
                    if (--indeg[nx] == 0) {
                        auto &q = ready[kind(nx)];
                        q.insert(std::upper_bound(q.begin(), q.end(), nx), nx);
                    }
        }
        if (!pass) pl.n_rounds_pre = uint32_t(pl.rounds.size());
    }
    pl.round_coop.assign(pl.rounds.size(), 0);
    for (size_t r = 0; r < pl.rounds.size(); ++r) {
        const RoundRec &rr = pl.rounds[r];
        bool all_fast = rr.count > 0;
        for (uint32_t e = 0; e < rr.count && all_fast; ++e) {
            const BoneRec &rec = pl.bones[pl.events[rr.first + e]];
            all_fast = (rec.bits & kBoneHasIk) && pl.iks[rec.ik].fast && pl.iks[rec.ik].nlinks <= kMaxFastLinks;
        }
        pl.round_coop[r] = all_fast ? uint8_t(rr.count) : uint8_t(0);
    }
}

}  // namespace

mmdx_status build_skeleton(const mmdx_skeleton_desc &d, SkeletonPlan &out, std::string &err) {
    out = SkeletonPlan();
    const uint32_t nb = d.n_bones;
    auto bad = [&](mmdx_status st, const std::string &what) { err = what; return st; };
    if (nb && (!d.rest_position || !d.parent)) return bad(MMDX_ERR_INVALID_ARGUMENT, "rest_position / parent is NULL");
    out.nb = nb;
    std::vector<uint32_t> pre, post;
    if (d.create_flags & MMDX_SKELETON_PHYSICS_SEAM) out.serial = true;   // per-bone state must survive between two calls
    auto flag = [&](uint32_t b) -> uint16_t { return d.flags ? d.flags[b] : uint16_t(0); };
    for (uint32_t b = 0; b < nb; ++b) {
        const uint16_t f = flag(b);
        if (f & (MMDX_BONE_HAS_IK | MMDX_BONE_APPEND_ROTATE | MMDX_BONE_APPEND_TRANSLATE)) out.serial = true;
        (f & MMDX_BONE_POST_PHYSICS ? post : pre).push_back(b);
    }
    auto level = [&](uint32_t b) { return d.transform_level ? uint32_t(d.transform_level[b]) : 0u; };
    auto by_level = [&](uint32_t a, uint32_t b) { return level(a) != level(b) ? level(a) < level(b) : a < b; };
    std::sort(pre.begin(), pre.end(), by_level);
    std::sort(post.begin(), post.end(), by_level);
    out.n_pre = uint32_t(pre.size());
    out.n_post = uint32_t(post.size());
    out.order = pre;
    out.order.insert(out.order.end(), post.begin(), post.end());
    std::vector<uint32_t> seq(nb);
    for (uint32_t i = 0; i < nb; ++i) seq[out.order[i]] = i;

    auto parent_of = [&](uint32_t b) -> int64_t {
        const int32_t p = d.parent[b];
        return (p >= 0 && uint32_t(p) < nb) ? int64_t(p) : -1;
    };
    for (uint32_t b = 0; b < nb; ++b)
        if (parent_of(b) == int64_t(b)) return bad(MMDX_ERR_INVALID_ARGUMENT, "bone " + std::to_string(b) + " is its own parent");
    out.local_offset.resize(size_t(nb) * 4, 0.0f);
    out.neg_rest.resize(size_t(nb) * 4, 0.0f);
    for (uint32_t b = 0; b < nb; ++b) {
        const float *pos = d.rest_position + 3 * size_t(b);
        const int64_t p = parent_of(b);
        for (int k = 0; k < 3; ++k) {
            out.local_offset[4 * size_t(b) + k] = p >= 0 ? pos[k] - d.rest_position[3 * size_t(p) + k] : pos[k];
            out.neg_rest[4 * size_t(b) + k] = -pos[k];
        }
    }
    // ---- bone morphs: flatten the group recursion into applications in the reference's order
    // (top-level morph index ascending, groups expanded depth-first in place, entries in file order) --------
    if (d.n_morphs) {
        if (!d.morph_type || !d.morph_offset) return bad(MMDX_ERR_INVALID_ARGUMENT, "n_morphs > 0 but morph_type / morph_offset is NULL");
        const uint32_t nm = d.n_morphs;
        out.nm = nm;
        for (uint32_t m = 0; m < nm; ++m)
            if (d.morph_offset[m + 1] < d.morph_offset[m]) return bad(MMDX_ERR_INVALID_ARGUMENT, "morph_offset is not ascending");
        if (d.morph_offset[nm] && (!d.morph_index || !d.morph_value)) return bad(MMDX_ERR_INVALID_ARGUMENT, "morph_index / morph_value is NULL");
        std::vector<float> chain;                            // sub-rates on the current recursion path
        std::vector<uint8_t> on_path(nm, 0);
        std::string rec_err;
        mmdx_status rec_st = MMDX_OK;
        // returns false on error
        std::function<bool(uint32_t, uint32_t)> visit = [&](uint32_t top, uint32_t m) -> bool {
            if (on_path[m]) { rec_st = MMDX_ERR_UNSUPPORTED; rec_err = "group morph " + std::to_string(m) + " contains itself"; return false; }
            const int32_t type = d.morph_type[m];
            if (type != 0 && type != 2) return true;
            on_path[m] = 1;
            for (uint32_t e = d.morph_offset[m]; e < d.morph_offset[m + 1]; ++e) {
                const uint32_t idx = d.morph_index[e];
                if (type == 0) {
                    if (idx >= nm) { rec_st = MMDX_ERR_BAD_INDEX; rec_err = "group morph " + std::to_string(m) + " refers to morph " + std::to_string(idx); return false; }
                    chain.push_back(d.morph_value[3 * size_t(e)]);
                    const bool ok = visit(top, idx);
                    chain.pop_back();
                    if (!ok) return false;
                } else {
                    if (idx >= nb) { rec_st = MMDX_ERR_BAD_INDEX; rec_err = "bone morph " + std::to_string(m) + " refers to bone " + std::to_string(idx); return false; }
                    BoneMorphApp a = BoneMorphApp();
                    a.bone = idx; a.top = top;
                    a.chain_off = uint32_t(out.app_chain.size()); a.chain_len = uint32_t(chain.size());
                    out.app_chain.insert(out.app_chain.end(), chain.begin(), chain.end());
                    for (int k = 0; k < 3; ++k) a.tr[k] = d.morph_value[3 * size_t(e) + k];
                    if (d.morph_rotation) for (int k = 0; k < 4; ++k) a.rot[k] = d.morph_rotation[4 * size_t(e) + k];
                    else a.rot[3] = 1.0f;
                    out.apps.push_back(a);
                    if (out.apps.size() > (size_t(1) << 22)) { rec_st = MMDX_ERR_UNSUPPORTED; rec_err = "bone morphs expand to more than 4M applications"; return false; }
                }
            }
            on_path[m] = 0;
            return true;
        };
        for (uint32_t m = 0; m < nm; ++m)
            if (!visit(m, m)) return bad(rec_st, rec_err);
    }

    if (!out.serial) {
        // Parent chains.  A parent that comes LATER in the evaluation sequence still holds the identity
        // its local matrix was reset to (PrePhysicsPosing, poser_impl.inl:371): the chain then starts with
        // kIdentityParent and that product is carried out like any other.
        out.chain_off.push_back(0);
        std::vector<uint32_t> rev;
        for (uint32_t b = 0; b < nb; ++b) {
            rev.clear();
            uint32_t c = b;
            for (;;) {
                rev.push_back(c);
                const int64_t p = parent_of(c);
                if (p < 0) break;
                if (seq[uint32_t(p)] > seq[c]) { rev.push_back(kIdentityParent); break; }
                c = uint32_t(p);
            }
            if (out.chain.size() + rev.size() > (size_t(1) << 26))
                return bad(MMDX_ERR_UNSUPPORTED, "parent chains too long for the parallel bone solve");
            out.max_chain = std::max(out.max_chain, uint32_t(rev.size()));
            out.chain.insert(out.chain.end(), rev.rbegin(), rev.rend());
            out.chain_off.push_back(uint32_t(out.chain.size()));
        }
        return MMDX_OK;
    }

    // ---- ordered solver tables (Poser ctor, L/motion/poser_impl.inl:29-97) -----------------------
    out.bones.resize(nb);
    for (uint32_t b = 0; b < nb; ++b) {
        BoneRec &r = out.bones[b];
        r = BoneRec();
        for (int k = 0; k < 3; ++k) { r.local_offset[k] = out.local_offset[4 * size_t(b) + k]; r.neg_rest[k] = out.neg_rest[4 * size_t(b) + k]; }
        r.parent = int32_t(parent_of(b));
        r.append_parent = -1;
        const uint16_t f = flag(b);
        if (f & (MMDX_BONE_APPEND_ROTATE | MMDX_BONE_APPEND_TRANSLATE)) {
            if (!d.append_parent || !d.append_ratio)
                return bad(MMDX_ERR_INVALID_ARGUMENT, "append bones present but append_parent / append_ratio is NULL");
            const int32_t ap = d.append_parent[b];
            if (ap >= 0 && uint32_t(ap) < nb) {            // otherwise has_append_ stays false
                r.append_parent = ap;
                r.append_ratio = d.append_ratio[b];
                if (f & MMDX_BONE_APPEND_ROTATE) r.bits |= kBoneAppendRot;
                if (f & MMDX_BONE_APPEND_TRANSLATE) r.bits |= kBoneAppendTr;
                ++out.n_append;
            }
        }
    }
    for (uint32_t b = 0; b < nb; ++b) {
        if (!(flag(b) & MMDX_BONE_HAS_IK)) continue;
        if (!d.ik_target || !d.ik_loop_count || !d.ik_angle_limit || !d.ik_link_offset)
            return bad(MMDX_ERR_INVALID_ARGUMENT, "IK bones present but an ik_* array is NULL");
        const uint32_t l0 = d.ik_link_offset[b], l1 = d.ik_link_offset[b + 1];
        if (l1 < l0 || (l1 > l0 && (!d.ik_link_bone || !d.ik_link_limited)))
            return bad(MMDX_ERR_INVALID_ARGUMENT, "ik_link_offset is not ascending or the link arrays are NULL");
        const int32_t tgt = d.ik_target[b];
        if (tgt < 0 || uint32_t(tgt) >= nb) return bad(MMDX_ERR_BAD_INDEX, "IK target of bone " + std::to_string(b) + " is out of range");
        IkRec ik = IkRec();
        ik.target = uint32_t(tgt);
        const int32_t loop = d.ik_loop_count[b];
        ik.loop = (loop < 0 || loop > 256) ? 256u : uint32_t(loop);
        ik.angle_limit = d.ik_angle_limit[b];
        ik.link0 = uint32_t(out.links.size());
        ik.nlinks = l1 - l0;
        for (uint32_t l = l0; l < l1; ++l) {
            const int32_t lb = d.ik_link_bone[l];
            if (lb < 0 || uint32_t(lb) >= nb) return bad(MMDX_ERR_BAD_INDEX, "IK link of bone " + std::to_string(b) + " is out of range");
            LinkRec lr = LinkRec();
            lr.bone = uint32_t(lb);
            lr.limited = d.ik_link_limited[l] ? 1u : 0u;
            lr.order = kOrderYZX;
            lr.fix = kFixNone;
            if (lr.limited) {
                if (!d.ik_link_lo || !d.ik_link_hi) return bad(MMDX_ERR_INVALID_ARGUMENT, "limited IK links present but ik_link_lo / ik_link_hi is NULL");
                const float *a = d.ik_link_lo + 3 * size_t(l), *z = d.ik_link_hi + 3 * size_t(l);
                for (int k = 0; k < 3; ++k) { lr.lo[k] = std::min(a[k], z[k]); lr.hi[k] = std::max(a[k], z[k]); }
                const double half_pi = 3.141592653589793238462643383279502884 * 0.5f;
                if (lr.lo[0] > -half_pi && lr.hi[0] < half_pi) lr.order = kOrderZXY;
                else if (lr.lo[1] > -half_pi && lr.hi[1] < half_pi) lr.order = kOrderXYZ;
                auto zero = [&](int k) { return std::fabs(lr.lo[k]) < 1e-7f && std::fabs(lr.hi[k]) < 1e-7f; };
                if (zero(0) && zero(1) && zero(2)) lr.fix = kFixAll;
                else if (zero(1) && zero(2)) lr.fix = kFixX;
                else if (zero(0) && zero(2)) lr.fix = kFixY;
                else if (zero(0) && zero(1)) lr.fix = kFixZ;
            }
            out.links.push_back(lr);
            out.bones[uint32_t(lb)].bits |= kBoneIsIkLink;
        }
        out.bones[b].bits |= kBoneHasIk;
        out.bones[b].ik = uint32_t(out.iks.size());
        out.iks.push_back(ik);
    }
    // Nested IK: a link or target that is itself an IK bone is solved inside the outer solve, as the reference's
    // recursion does (UpdateBoneTransform calls itself for links and target, poser_impl.inl:196-206).  The device code
    // unrolls that recursion kMaxIkDepth deep; a cycle (endless recursion upstream) or a deeper nest is rejected.
    {
        std::vector<uint32_t> depth(nb, 0);                 // 0 = not computed
        std::vector<uint8_t> on_path(nb, 0);
        // returns the nesting depth of x's solve (0: not an IK bone), -1 for a cycle, -2 for a nest deeper than the device
        // code unrolls; `level` = solves already on the stack, so the recursion itself never goes deeper than that limit
        // (a file may chain a million IK bones)
        std::function<int64_t(uint32_t, uint32_t)> nest = [&](uint32_t x, uint32_t level) -> int64_t {
            if (!(out.bones[x].bits & kBoneHasIk)) return 0;
            if (on_path[x]) return -1;
            if (level >= kMaxIkDepth) return -2;
            if (depth[x]) return depth[x];
            on_path[x] = 1;
            IkRec &ik = out.iks[out.bones[x].ik];
            int64_t deepest = 0;
            for (uint32_t j = 0; j <= ik.nlinks; ++j) {
                const uint32_t y = j < ik.nlinks ? out.links[ik.link0 + j].bone : ik.target;
                const int64_t dd = nest(y, level + 1);
                if (dd < 0) { on_path[x] = 0; return dd; }
                if (dd > 0) { ik.nested = 1; out.nested_ik = true; }
                deepest = std::max(deepest, dd);
            }
            on_path[x] = 0;
            depth[x] = uint32_t(deepest + 1);
            return depth[x];
        };
        for (uint32_t b = 0; b < nb; ++b) {
            const int64_t dd = nest(b, 0);
            if (dd == -1) return bad(MMDX_ERR_UNSUPPORTED, "IK bone " + std::to_string(b) + " is (indirectly) a link or target of its own "
                                                           "solve: the reference recurses without end");
            if (dd == -2 || dd > int64_t(kMaxIkDepth))
                return bad(MMDX_ERR_UNSUPPORTED, "IK solves nested more than " + std::to_string(kMaxIkDepth) + " deep at bone " + std::to_string(b));
        }
    }
    // Which chains can run on the LDS window: link j's parent is link j+1, the target hangs off link 0,
    // all bones distinct, no append bone inside (appends read bones outside the window), no nested IK.
    for (uint32_t b = 0; b < nb; ++b) {
        if (!(out.bones[b].bits & kBoneHasIk)) continue;
        IkRec &ik = out.iks[out.bones[b].ik];
        const LinkRec *lk = out.links.data() + ik.link0;
        const uint32_t n = ik.nlinks;
        bool ok = !ik.nested && n >= 1 && n <= kMaxFastLinks && ik.target != b && out.bones[ik.target].parent == int32_t(lk[0].bone) &&
                  !(out.bones[ik.target].bits & (kBoneAppendRot | kBoneAppendTr));
        for (uint32_t j = 0; ok && j < n; ++j) {
            if (out.bones[lk[j].bone].bits & (kBoneAppendRot | kBoneAppendTr)) ok = false;
            if (lk[j].bone == ik.target || lk[j].bone == b) ok = false;
            if (j + 1 < n && out.bones[lk[j].bone].parent != int32_t(lk[j + 1].bone)) ok = false;
            for (uint32_t k = j + 1; ok && k < n; ++k) if (lk[k].bone == lk[j].bone) ok = false;
        }
        ik.outside_parent = -1;
        if (ok) {
            const int32_t op = out.bones[lk[n - 1].bone].parent;
            for (uint32_t j = 0; j < n; ++j) if (op == int32_t(lk[j].bone)) ok = false;
            if (op == int32_t(ik.target)) ok = false;
            ik.outside_parent = op;
        }
        ik.fast = ok ? 1u : 0u;
        if (ok) {
            ++out.n_fast;
            out.fast_slots = std::max(out.fast_slots, n + 2);
        }
    }
    out.n_ik = uint32_t(out.iks.size());
    out.n_links = uint32_t(out.links.size());
    schedule_rounds(out);
    return MMDX_OK;
}

}  // namespace mmdx
