// Host "model compile" (csrc/plan.cpp) under AddressSanitizer + UBSan on the CPU: random models incl.
// group morphs, ragged tiles, invalid indices (must be rejected, not read).  Built and run by
// tests/test_sanitizers.py; GPU sanitizers are not available on the pool.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../simple_mmd_renderer_amd/csrc/plan.hpp"
#include "../simple_mmd_renderer_amd/csrc/rig.hpp"
#include "../simple_mmd_renderer_amd/csrc/vmd.hpp"

// The round schedule of the ordered solver must show every event exactly the state the serial sequence
// shows it: replay both with a version number per bone (= the event that last wrote it) and compare what
// each event sees; events sharing a round must not touch each other's bones at all.
static int check_rounds(const mmdx::SkeletonPlan &sp) {
    const uint32_t nb = sp.nb;
    std::vector<uint32_t> reads, writes;
    std::vector<std::vector<int64_t>> seen(nb);
    std::vector<int64_t> version(nb, -1);
    auto look = [&](uint32_t b) {
        std::vector<int64_t> v;
        mmdx::solve_event_sets(sp, b, reads, writes);
        for (uint32_t x : reads) v.push_back(version[x]);
        for (uint32_t x : writes) v.push_back(version[x]);
        return v;
    };
    for (uint32_t s = 0; s < nb; ++s) {
        const uint32_t b = sp.order[s];
        seen[b] = look(b);
        for (uint32_t x : writes) version[x] = b;
    }
    std::fill(version.begin(), version.end(), -1);
    std::vector<uint8_t> ran(nb, 0), pre(nb, 0), touched(nb);
    for (uint32_t s = 0; s < sp.n_pre; ++s) pre[sp.order[s]] = 1;
    if (sp.n_rounds_pre > sp.rounds.size()) return 20;
    size_t total = 0;
    for (size_t r = 0; r < sp.rounds.size(); ++r) {
        const mmdx::RoundRec rr = sp.rounds[r];
        if (rr.count == 0 || rr.count > mmdx::kSolveSlots || rr.first != total || rr.first + rr.count > sp.events.size()) return 21;
        total += rr.count;
        std::fill(touched.begin(), touched.end(), uint8_t(0));     // 1 = read, 2 = written by an event of this round
        for (uint32_t k = 0; k < rr.count; ++k) {
            const uint32_t b = sp.events[rr.first + k];
            if (b >= nb || ran[b] || pre[b] != (r < sp.n_rounds_pre ? 1 : 0)) return 22;
            ran[b] = 1;
            if (look(b) != seen[b]) return 23;
            const mmdx::BoneRec &rec = sp.bones[b];
            if ((rec.bits & mmdx::kBoneHasIk) && sp.iks[rec.ik].fast && k >= sp.windows) return 24;
            for (uint32_t x : writes) { if (touched[x]) return 25; }
            for (uint32_t x : reads) { if (touched[x] == 2) return 25; }
            for (uint32_t x : writes) touched[x] = 2;
            for (uint32_t x : reads) if (!touched[x]) touched[x] = 1;
        }
        for (uint32_t k = 0; k < rr.count; ++k) {
            mmdx::solve_event_sets(sp, sp.events[rr.first + k], reads, writes);
            for (uint32_t x : writes) version[x] = sp.events[rr.first + k];
        }
    }
    if (total != nb || sp.events.size() != nb) return 26;
    return 0;
}

int main() {
    std::mt19937 rng(1234);
    int built = 0, rejected = 0;
    for (int it = 0; it < 60; ++it) {
        const uint32_t nv = 1 + rng() % 3000, nb = 1 + rng() % 300, nm = rng() % 8;
        std::vector<float> pos(nv * 3, 1.f), nrm(nv * 3, 0.5f), uv(nv * 2, 0.25f), w(nv * 4);
        std::vector<int32_t> type(nv), ids(nv * 4), parent(nb, -1), mtype(nm);
        std::vector<uint32_t> moff(nm + 1, 0), midx;
        std::vector<float> mval;
        for (uint32_t b = 1; b < nb; ++b) parent[b] = int32_t(rng() % b);
        const bool poison = it % 5 == 4;
        for (uint32_t v = 0; v < nv; ++v) {
            type[v] = int32_t(rng() % 5);
            for (int k = 0; k < 4; ++k) { ids[4 * v + k] = int32_t(rng() % nb); w[4 * v + k] = float(rng() % 1000) / 999.f; }
        }
        if (poison) ids[4 * (rng() % nv)] = int32_t(nb + rng() % 100);   // may hit a zero-weight slot: either way no OOB
        for (uint32_t m = 0; m < nm; ++m) {
            mtype[m] = (m > 0 && rng() % 3 == 0) ? MMDX_MORPH_GROUP : (rng() % 6 == 0 ? MMDX_MORPH_UV : MMDX_MORPH_VERTEX);
            const uint32_t k = rng() % 50;
            for (uint32_t j = 0; j < k; ++j) {
                midx.push_back(mtype[m] == MMDX_MORPH_GROUP ? rng() % m : rng() % nv);
                mval.push_back(0.5f); mval.push_back(-0.25f); mval.push_back(0.125f);
            }
            moff[m + 1] = uint32_t(midx.size());
        }
        mmdx_model_desc d{};
        d.struct_size = sizeof(d);
        d.flags = (it % 2 ? MMDX_CREATE_NORMALIZE : 0) | (it % 3 == 0 ? MMDX_CREATE_F16_POSITIONS : 0);
        d.n_vertices = nv; d.n_bones = nb; d.n_morphs = nm;
        d.positions = pos.data(); d.normals = nrm.data(); d.uvs = uv.data();
        d.skin_type = type.data(); d.bone_ids = ids.data(); d.bone_weights = w.data();
        d.bone_parent = parent.data(); d.morph_type = mtype.data(); d.morph_offset = moff.data();
        d.morph_index = midx.data(); d.morph_value = mval.data();
        mmdx::Plan plan;
        std::string err;
        const mmdx_status st = mmdx::build_plan(d, plan, err);
        if (st != MMDX_OK) { ++rejected; continue; }
        ++built;
        std::vector<float> rates(nm, 0.7f), ws(plan.ns + 1);
        mmdx::flatten_slot_weights(plan, rates.data(), ws.data());
        // every table entry must name a slot <= ns and every tile-local bone index must be in range
        for (size_t e = 0; e < plan.ne; ++e) {
            uint32_t slot;
            if (plan.f16) slot = plan.entries16[4 * e + 3];
            else std::memcpy(&slot, &plan.entries[4 * e + 3], 4);
            if (slot > plan.ns) { std::printf("bad slot\n"); return 2; }
        }
        for (const auto &t : plan.tiles)
            if (t.n1 + t.n2 > t.nv || t.bone_off + t.nbt > plan.bone_list.size()) { std::printf("bad tile\n"); return 2; }
    }
    std::printf("built=%d rejected=%d\n", built, rejected);
    if (built <= 30) return 3;

    // PMX parser under the sanitizers: a minimal valid file, then byte-level fuzzing of it (every
    // outcome must be "parsed" or "rejected", never an out-of-bounds read)
    std::vector<uint8_t> f;
    auto put = [&](const void *p, size_t n) { f.insert(f.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto f32 = [&](float v) { put(&v, 4); };
    auto text = [&](const char *s) { i32(int32_t(std::strlen(s))); put(s, std::strlen(s)); };
    put("PMX ", 4); f32(2.0f);
    const uint8_t globals[9] = {8, 1, 0, 1, 1, 1, 1, 1, 1};
    put(globals, 9);
    text("m"); text("m"); text(""); text("");
    i32(3);
    for (int v = 0; v < 3; ++v) {
        for (int k = 0; k < 8; ++k) f32(float(v + k));
        const uint8_t t = uint8_t(v == 2 ? 2 : v);
        put(&t, 1);
        const uint8_t b[4] = {0, 1, 0, 1};
        if (t == 0) put(b, 1);
        else if (t == 1) { put(b, 2); f32(0.5f); }
        else { put(b, 4); for (int k = 0; k < 4; ++k) f32(0.25f); }
        f32(1.0f);
    }
    i32(3); const uint8_t tri[3] = {0, 1, 2}; put(tri, 3);
    i32(0); i32(0);                                  // textures, materials
    i32(2);                                          // bones
    for (int b = 0; b < 2; ++b) {
        text("b"); text("b"); f32(0); f32(float(b)); f32(0);
        const uint8_t parent = b ? 0 : 255; put(&parent, 1);
        i32(0); const uint16_t flags = 0x001f; put(&flags, 2);
        const uint8_t child = 1; put(&child, 1);
    }
    i32(1);                                          // one vertex morph
    text("mm"); text("mm"); const uint8_t pt[2] = {1, 1}; put(pt, 2); i32(1);
    const uint8_t vi = 2; put(&vi, 1); f32(0.1f); f32(0.2f); f32(0.3f);
    mmdx_pmx_t pmx = nullptr;
    if (mmdx_pmx_parse(f.data(), f.size(), &pmx) != MMDX_OK) { std::printf("valid pmx rejected: %s\n", mmdx_last_error_string()); return 4; }
    mmdx_pmx_destroy(pmx);
    int ok = 0, bad = 0;
    for (int it = 0; it < 4000; ++it) {
        std::vector<uint8_t> g = f;
        const int nmut = 1 + int(rng() % 4);
        for (int k = 0; k < nmut; ++k) g[rng() % g.size()] = uint8_t(rng());
        if (it % 7 == 0) g.resize(rng() % g.size());
        pmx = nullptr;
        if (mmdx_pmx_parse(g.data(), g.size(), &pmx) == MMDX_OK) { ++ok; mmdx_pmx_destroy(pmx); } else ++bad;
    }
    std::printf("pmx fuzz: parsed=%d rejected=%d\n", ok, bad);

    // PMD parser: a minimal valid file (2 vertices, 3 bones incl. an IK bone with two IK records, base morph
    // + one morph), then byte-level fuzzing; whatever parses must also survive the skeleton compile
    {
        std::vector<uint8_t> f;
        auto put = [&](const void *p, size_t n) { f.insert(f.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
        auto zeros = [&](size_t n) { f.insert(f.end(), n, 0); };
        const float one = 1.0f;
        put("Pmd", 3); put(&one, 4); zeros(276);
        const uint32_t nvv = 2; put(&nvv, 4);
        for (uint32_t i = 0; i < nvv; ++i) { float v[8] = {float(i), 1, 2, 0, 1, 0, 0.5f, 0.5f}; put(v, 32); const int16_t id[2] = {0, 1}; put(id, 4); const uint8_t w[2] = {60, 0}; put(w, 2); }
        const uint32_t nidx = 0, nmat = 0; put(&nidx, 4); put(&nmat, 4);
        const uint16_t nbb = 3; put(&nbb, 2);
        for (uint16_t b = 0; b < nbb; ++b) { uint8_t rec[39] = {0}; rec[0] = uint8_t('a' + b); const int16_t par = int16_t(b) - 1; std::memcpy(rec + 20, &par, 2); rec[24] = b == 2 ? 2 : 0; put(rec, 39); }
        const uint16_t nikk = 2; put(&nikk, 2);
        for (uint16_t k = 0; k < nikk; ++k) { const int16_t ib = 2, tg = 1; put(&ib, 2); put(&tg, 2); const uint8_t len = 1; put(&len, 1); const uint16_t it = 3; put(&it, 2); put(&one, 4); const uint16_t ch = k; put(&ch, 2); }
        const uint16_t nmm = 2; put(&nmm, 2);
        for (uint16_t k = 0; k < nmm; ++k) { uint8_t hd[25] = {0}; hd[0] = 'm'; const uint32_t cnt = 1; std::memcpy(hd + 20, &cnt, 4); hd[24] = uint8_t(k); put(hd, 25); const uint32_t idx = 0; put(&idx, 4); float o[3] = {0.1f, 0.2f, 0.3f}; put(o, 12); }
        zeros(1 + 1 + 4);
        mmdx_pmx_t h = nullptr;
        if (mmdx_pmd_parse(f.data(), f.size(), &h) != MMDX_OK) { std::printf("valid pmd rejected: %s\n", mmdx_last_error_string()); return 14; }
        mmdx_pmx_info pi{}; pi.struct_size = sizeof(pi);
        mmdx_pmx_get_info(h, &pi);
        if (pi.n_vertices != 2 || pi.n_bones != 4 || pi.n_morphs != 2) { std::printf("pmd counts %u %u %u\n", pi.n_vertices, pi.n_bones, pi.n_morphs); return 15; }
        mmdx_pmx_destroy(h);
        int pok = 0, pbad = 0;
        for (int it = 0; it < 4000; ++it) {
            std::vector<uint8_t> g = f;
            const int nmut = 1 + int(rng() % 4);
            for (int k = 0; k < nmut; ++k) g[rng() % g.size()] = uint8_t(rng());
            if (it % 7 == 0) g.resize(rng() % g.size());
            h = nullptr;
            if (mmdx_pmd_parse(g.data(), g.size(), &h) == MMDX_OK) {
                ++pok;
                mmdx_skeleton_desc sd{};
                mmdx_pmx_get_skeleton_desc(h, &sd);
                mmdx::SkeletonPlan sp;
                std::string err;
                (void)mmdx::build_skeleton(sd, sp, err);
                mmdx_pmx_destroy(h);
            } else ++pbad;
        }
        std::printf("pmd fuzz: parsed=%d rejected=%d\n", pok, pbad);
    }

    // VMD parser: a small valid motion, then the same byte-level fuzzing
    std::vector<uint8_t> m(50, 0);
    std::memcpy(m.data(), "Vocaloid Motion Data 0002", 25);
    auto mput = [&](const void *p, size_t n) { m.insert(m.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    const uint32_t nbone = 2, nmorph = 5;
    mput(&nbone, 4);
    for (uint32_t b = 0; b < nbone; ++b) {
        uint8_t rec[111] = {0};
        std::memcpy(rec, "bone", 4);
        const uint32_t fr = b * 10; std::memcpy(rec + 15, &fr, 4);
        mput(rec, sizeof(rec));
    }
    mput(&nmorph, 4);
    for (uint32_t k = 0; k < nmorph; ++k) {
        uint8_t rec[23] = {0};
        std::memcpy(rec, k % 2 ? "a" : "b", 1);
        const uint32_t fr = k * 3; const float w = 0.25f * float(k);
        std::memcpy(rec + 15, &fr, 4); std::memcpy(rec + 19, &w, 4);
        mput(rec, sizeof(rec));
    }
    mmdx_vmd_t vmd = nullptr;
    if (mmdx_vmd_parse(m.data(), m.size(), &vmd) != MMDX_OK) { std::printf("valid vmd rejected: %s\n", mmdx_last_error_string()); return 5; }
    const char *names[3] = {"a", "zz", "b"};
    mmdx_morph_motion_t mm = nullptr;
    if (mmdx_vmd_bind_morphs(vmd, 3, names, &mm) != MMDX_OK) return 6;
    uint32_t nmm = 0, mapped = 0, nkeys = 0;
    mmdx_morph_motion_get_info(mm, &nmm, &mapped, &nkeys);
    if (nmm != 3 || mapped != 2 || nkeys != 5) { std::printf("bind: %u %u %u\n", nmm, mapped, nkeys); return 7; }
    mmdx_morph_motion_destroy(mm);
    mmdx_vmd_destroy(vmd);
    ok = bad = 0;
    for (int it = 0; it < 4000; ++it) {
        std::vector<uint8_t> g = m;
        const int nmut = 1 + int(rng() % 4);
        for (int k = 0; k < nmut; ++k) g[rng() % g.size()] = uint8_t(rng());
        if (it % 7 == 0) g.resize(rng() % g.size());
        vmd = nullptr;
        if (mmdx_vmd_parse(g.data(), g.size(), &vmd) == MMDX_OK) {
            ++ok;
            // bone tracks of whatever parsed: curve presampling + binding (rig.cpp)
            const mmdx::VmdBoneTracks t = mmdx::vmd_bone_tracks(vmd);
            const char *bn[3] = {"bone", nullptr, "x"};
            mmdx::BoneMotionHost bm;
            mmdx::build_bone_motion(*t.names, *t.off, t.keys, 3, bn, bm);
            if (bm.key_off.size() != 4 || bm.key_curve.size() != bm.key_frame.size() * 4) return 8;
            mmdx_vmd_destroy(vmd);
        } else ++bad;
    }
    std::printf("vmd fuzz: parsed=%d rejected=%d\n", ok, bad);

    // skeleton compile (rig.cpp): random hierarchies incl. forward / out-of-range / self parents
    int sk_ok = 0, sk_bad = 0, sk_rounds = 0, sk_events = 0;
    for (int it = 0; it < 900; ++it) {
        const uint32_t nb = rng() % 70;
        std::vector<float> rest(size_t(nb) * 3 + 1, 1.0f);
        std::vector<int32_t> par(nb + 1), lvl(nb + 1);
        std::vector<uint16_t> fl(nb + 1);
        for (uint32_t b = 0; b < nb; ++b) {
            par[b] = int32_t(rng() % (nb + 4)) - 2;
            lvl[b] = int32_t(rng() % 4) - 1;
            fl[b] = rng() % 5 == 0 ? 0x1000 : 0;
        }
        mmdx_skeleton_desc d{};
        d.struct_size = sizeof(d); d.n_bones = nb;
        d.rest_position = rest.data(); d.parent = par.data();
        d.transform_level = it % 3 ? lvl.data() : nullptr; d.flags = it % 2 ? fl.data() : nullptr;
        d.create_flags = it % 5 == 0 ? MMDX_SKELETON_PHYSICS_SEAM : 0;   // forces the ordered solver's tables for any rig
        mmdx::SkeletonPlan sp;
        std::string err;
        // IK / append tables, with some indices out of range on purpose (targets / links may be IK bones themselves:
        // nested solves up to 3 deep are scheduled, cycles and deeper nests rejected)
        std::vector<int32_t> app(nb + 1), tgt(nb + 1), loop(nb + 1), lbone;
        std::vector<float> ratio(nb + 1, 0.5f), ang(nb + 1, 1.0f), lim;
        std::vector<uint32_t> loff(nb + 2, 0);
        std::vector<uint8_t> limited;
        if (it % 4 == 3) {
            for (uint32_t b = 0; b < nb; ++b) {
                if (rng() % 6 == 0) fl[b] |= uint16_t(0x0100 << (rng() % 2));
                if (rng() % 9 == 0) fl[b] |= 0x0020;
                app[b] = int32_t(rng() % (nb + 3)) - 1;
                tgt[b] = int32_t(rng() % (nb + 1));
                loop[b] = int32_t(rng() % 400) - 50;
                const uint32_t nl = (fl[b] & 0x0020) ? rng() % 4 : 0;
                for (uint32_t l = 0; l < nl; ++l) {
                    lbone.push_back(int32_t(rng() % (nb + 1)));
                    limited.push_back(uint8_t(rng() % 2));
                    for (int k = 0; k < 6; ++k) lim.push_back(float(int(rng() % 7) - 3));
                }
                loff[b + 1] = uint32_t(lbone.size());
            }
            d.flags = fl.data();
            d.append_parent = app.data(); d.append_ratio = ratio.data();
            d.ik_target = tgt.data(); d.ik_loop_count = loop.data(); d.ik_angle_limit = ang.data();
            d.ik_link_offset = loff.data(); d.ik_link_bone = lbone.data(); d.ik_link_limited = limited.data();
            d.ik_link_lo = lim.data(); d.ik_link_hi = lim.data() + 3;
        }
        if (mmdx::build_skeleton(d, sp, err) == MMDX_OK) {
            if (sp.serial) {
                if (int rc = check_rounds(sp)) return rc;
                sk_rounds += int(sp.rounds.size());
                sk_events += int(sp.events.size());
                for (const auto &r : sp.bones)
                    if (r.parent >= int32_t(nb) || ((r.bits & 3u) && uint32_t(r.append_parent) >= nb)) return 11;
                for (const auto &k : sp.iks) if (k.target >= nb || k.loop > 256 || k.link0 + k.nlinks > sp.links.size()) return 12;
                for (const auto &l : sp.links) if (l.bone >= nb) return 13;
                ++sk_ok;
                continue;
            }
            ++sk_ok;
            if (sp.chain_off.size() != size_t(nb) + 1 || sp.chain_off.back() != sp.chain.size()) return 9;
            for (uint32_t c : sp.chain) if (c != mmdx::kIdentityParent && c >= nb) return 10;
        } else ++sk_bad;
    }
    std::printf("skeleton: compiled=%d rejected=%d; ordered solver: %d events in %d rounds\n", sk_ok, sk_bad, sk_events, sk_rounds);
    {   // 200 000 IK bones, each the target of the previous one: the nesting check must reject it without recursing
        // 200 000 frames deep; the same chain closed into a ring is a cycle
        const uint32_t nb = 200000;
        std::vector<float> rest(size_t(nb) * 3, 0.5f), ang(nb, 1.0f);
        std::vector<int32_t> par(nb, -1), tgt(nb), loop(nb, 1), lbone;
        std::vector<uint16_t> fl(nb, 0x0020);
        std::vector<uint32_t> loff(nb + 1, 0);
        std::vector<uint8_t> limited(1, 0);
        lbone.push_back(0);
        for (uint32_t b = 0; b < nb; ++b) tgt[b] = int32_t(b + 1 < nb ? b + 1 : b);
        fl[nb - 1] = 0;                                         // the last one is a plain bone
        mmdx_skeleton_desc d{};
        d.struct_size = sizeof(d); d.n_bones = nb; d.rest_position = rest.data(); d.parent = par.data(); d.flags = fl.data();
        d.ik_target = tgt.data(); d.ik_loop_count = loop.data(); d.ik_angle_limit = ang.data();
        d.ik_link_offset = loff.data(); d.ik_link_bone = lbone.data(); d.ik_link_limited = limited.data();
        mmdx::SkeletonPlan sp;
        std::string err;
        if (mmdx::build_skeleton(d, sp, err) != MMDX_ERR_UNSUPPORTED) return 14;
        fl[nb - 1] = 0x0020; tgt[nb - 1] = 0;                   // a ring
        if (mmdx::build_skeleton(d, sp, err) != MMDX_ERR_UNSUPPORTED) return 15;
        std::printf("deep / cyclic IK nests rejected: %s\n", err.c_str());
    }
    return 0;
}

// vmd.cpp hands device buffers back to api.cpp (HIP); nothing to release in this host-only build
#include "../simple_mmd_renderer_amd/csrc/vmd.hpp"
void mmdx::morph_motion_release_device(mmdx::MorphMotionDevice &) {}
