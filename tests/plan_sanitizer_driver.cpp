// Host "model compile" (csrc/plan.cpp) under AddressSanitizer + UBSan on the CPU: random models incl.
// group morphs, ragged tiles, invalid indices (must be rejected, not read).  Built and run by
// tests/test_sanitizers.py; GPU sanitizers are not available on the pool.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../simple_mmd_renderer_amd/csrc/plan.hpp"

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
    return built > 30 ? 0 : 3;
}
