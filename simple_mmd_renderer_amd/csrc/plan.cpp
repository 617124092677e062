// plan.cpp -- see plan.hpp.  Host-only; no HIP.
#include "plan.hpp"

#include <algorithm>
#include <cstring>

namespace mmdx {

namespace {

constexpr double kEpsD = 1e-7;  // L/util/math.inl:24 (a double literal in the reference)

inline float lerp_lo() { return float(kEpsD); }         // l <  this -> first operand only
inline float lerp_hi() { return float(1.0 - kEpsD); }   // l >  this -> second operand only

struct SlotBuilder {
    const mmdx_model_desc &d;
    Plan &p;
    std::vector<uint32_t> slot_morph;  // slot -> vertex morph it applies
    std::vector<float> chain;          // current group chain
    std::string *err;
    uint64_t entry_budget = 0;

    mmdx_status visit(uint32_t m, uint32_t top, uint32_t depth) {
        if (depth > kMaxGroupDepth) {
            *err = "group morph nesting deeper than " + std::to_string(kMaxGroupDepth) +
                   " (cycle?) at morph " + std::to_string(m);
            return MMDX_ERR_UNSUPPORTED;
        }
        const uint32_t b = d.morph_offset[m], e = d.morph_offset[m + 1];
        if (d.morph_type[m] == MMDX_MORPH_VERTEX) {
            entry_budget += uint64_t(e - b);
            if (entry_budget > 0x7fffffffull || p.slot_top.size() >= (1u << 24)) {
                *err = "group morph expansion is too large";
                return MMDX_ERR_UNSUPPORTED;
            }
            p.slot_top.push_back(top);
            p.chain_rate.insert(p.chain_rate.end(), chain.begin(), chain.end());
            p.chain_off.push_back(uint32_t(p.chain_rate.size()));
            slot_morph.push_back(m);
        } else if (d.morph_type[m] == MMDX_MORPH_GROUP) {
            for (uint32_t j = b; j < e; ++j) {
                const uint32_t sub = d.morph_index[j];
                if (sub >= d.n_morphs) {
                    *err = "group morph " + std::to_string(m) + " refers to morph " +
                           std::to_string(sub) + " >= n_morphs";
                    return MMDX_ERR_BAD_INDEX;
                }
                chain.push_back(d.morph_value[3 * size_t(j)]);
                mmdx_status st = visit(sub, top, depth + 1);
                chain.pop_back();
                if (st != MMDX_OK) return st;
            }
        }  // bone / uv / material morphs never touch vertices (poser_impl.inl:347-358)
        return MMDX_OK;
    }
};

}  // namespace

uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return uint16_t(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // nan / inf
    if (x >= 0x477ff000u) return uint16_t(sign | 0x7c00u);  // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return uint16_t(sign);             // <= 2^-25 -> 0 (ties to even)
    int32_t exp = int32_t(x >> 23) - 127;
    uint32_t man = (x & 0x7fffffu) | 0x800000u;
    uint32_t shift, half;
    if (exp < -14) {  // subnormal half
        shift = uint32_t(13 + (-14 - exp));
        half = 0;
    } else {
        shift = 13;
        half = uint32_t(exp + 15) << 10;
        man &= 0x7fffffu;
    }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (q & 1u))) ++q;
    return uint16_t(sign | (half + q));  // mantissa carry correctly bumps the exponent
}

void flatten_slot_weights(const Plan &plan, const float *rates, float *out) {
    for (uint32_t s = 0; s < plan.ns; ++s) {
        float r = rates[plan.slot_top[s]];
        bool skip = double(r) < kEpsD;
        for (uint32_t c = plan.chain_off[s]; !skip && c < plan.chain_off[s + 1]; ++c) {
            r = plan.chain_rate[c] * r;
            skip = double(r) < kEpsD;
        }
        out[s] = skip ? 0.0f : r;
    }
}

mmdx_status build_plan(const mmdx_model_desc &d, Plan &p, std::string &err) {
    if (d.struct_size != sizeof(mmdx_model_desc)) {
        err = "mmdx_model_desc.struct_size mismatch";
        return MMDX_ERR_INVALID_ARGUMENT;
    }
    const uint32_t nv = d.n_vertices, nb = d.n_bones, nm = d.n_morphs;
    if (nv == 0 || nb == 0) {
        err = "model needs at least one vertex and one bone";
        return MMDX_ERR_INVALID_ARGUMENT;
    }
    if (!d.positions || !d.normals || !d.skin_type || !d.bone_ids || !d.bone_weights) {
        err = "positions, normals, skin_type, bone_ids and bone_weights are required";
        return MMDX_ERR_INVALID_ARGUMENT;
    }
    if (nm && (!d.morph_type || !d.morph_offset)) {
        err = "morph_type and morph_offset are required when n_morphs > 0";
        return MMDX_ERR_INVALID_ARGUMENT;
    }
    if (uint64_t(nv) * 3 >= (1ull << 32)) {
        err = "too many vertices";
        return MMDX_ERR_UNSUPPORTED;
    }
    p = Plan();
    p.nv = nv; p.nb = nb; p.nm = nm; p.flags = d.flags;
    p.f16 = (d.flags & MMDX_CREATE_F16_POSITIONS) != 0;

    // ---- morph table validation ------------------------------------------------------------
    for (uint32_t m = 0; m < nm; ++m) {
        if (d.morph_offset[m] > d.morph_offset[m + 1]) {
            err = "morph_offset is not monotonic at morph " + std::to_string(m);
            return MMDX_ERR_INVALID_ARGUMENT;
        }
        if (d.morph_offset[m + 1] > d.morph_offset[m] && (!d.morph_index || !d.morph_value)) {
            err = "morph_index / morph_value missing";
            return MMDX_ERR_INVALID_ARGUMENT;
        }
        if (d.morph_type[m] == MMDX_MORPH_VERTEX)
            for (uint32_t j = d.morph_offset[m]; j < d.morph_offset[m + 1]; ++j)
                if (d.morph_index[j] >= nv) {
                    err = "vertex morph " + std::to_string(m) + " refers to vertex " +
                          std::to_string(d.morph_index[j]) + " >= n_vertices";
                    return MMDX_ERR_BAD_INDEX;
                }
    }

    // ---- skin: class + optional Model::Normalize (model_impl.inl:406-452) --------------------
    p.cls.resize(nv);
    p.ids.assign(size_t(nv) * 4, 0);
    p.wts.assign(size_t(nv) * 4, 0.0f);
    const bool normalize = (d.flags & MMDX_CREATE_NORMALIZE) != 0;
    auto valid = [nb](int32_t b) { return b >= 0 && uint32_t(b) < nb; };
    auto parent_of = [&](int32_t b) -> int64_t { return d.bone_parent ? d.bone_parent[b] : -1; };
    for (uint32_t i = 0; i < nv; ++i) {
        const int32_t *id = d.bone_ids + 4 * size_t(i);
        const float *w = d.bone_weights + 4 * size_t(i);
        int32_t *oid = p.ids.data() + 4 * size_t(i);
        float *ow = p.wts.data() + 4 * size_t(i);
        int32_t t = d.skin_type[i];
        int32_t b0 = id[0], b1 = id[1];
        if (normalize && (t == MMDX_SKIN_BDEF2 || t == MMDX_SKIN_SDEF)) {
            bool retag = true;
            if (t == MMDX_SKIN_SDEF)
                retag = !(valid(b0) && valid(b1) && (parent_of(b0) == b1 || parent_of(b1) == b0));
            if (retag) {
                if (w[0] == 0.0f) { t = MMDX_SKIN_BDEF1; b0 = b1; }
                else if (w[0] == 1.0f) { t = MMDX_SKIN_BDEF1; }
                else { t = MMDX_SKIN_BDEF2; }
            }
        }
        auto bad = [&](int k, int32_t b) {
            err = "vertex " + std::to_string(i) + ": bone id " + std::to_string(b) + " (slot " +
                  std::to_string(k) + ") is out of range and its weight is not zero";
            return MMDX_ERR_BAD_INDEX;
        };
        if (t == MMDX_SKIN_BDEF1) {
            if (!valid(b0)) return bad(0, b0);
            p.cls[i] = 0; oid[0] = b0; ++p.n1;
        } else if (t == MMDX_SKIN_BDEF4) {
            int32_t first_valid = -1;
            for (int k = 0; k < 4; ++k)
                if (valid(id[k])) { first_valid = id[k]; break; }
            for (int k = 0; k < 4; ++k) {
                int32_t b = id[k];
                if (!valid(b)) {
                    // PMX "no bone" (-1, or 255/65535 after libmmd's zero extension,
                    // L/util/dwarf_impl.inl:90-95) with weight 0: the reference reads out of
                    // bounds and multiplies by 0; we substitute a valid bone (documented divergence)
                    if (w[k] != 0.0f || first_valid < 0) return bad(k, b);
                    b = first_valid;
                }
                oid[k] = b; ow[k] = w[k];
            }
            p.cls[i] = 2; ++p.n4;
        } else {  // BDEF2, SDEF (evaluated as BDEF2) and unknown tags (reference `default:`)
            const float l = w[0];
            const bool need_b1 = !(l > lerp_hi());  // Lerp(S[b1], S[b0])[l]: S[b1] unless l > hi
            const bool need_b0 = !(l < lerp_lo());
            if (!valid(b0)) { if (need_b0 || !valid(b1)) return bad(0, b0); b0 = b1; }
            if (!valid(b1)) { if (need_b1) return bad(1, b1); b1 = b0; }
            p.cls[i] = 1; oid[0] = b0; oid[1] = b1; ow[0] = l; ++p.n2;
        }
    }

    // ---- morph slots in the reference's traversal order; entries per vertex ----------------------
    p.chain_off.push_back(0);
    SlotBuilder sb{d, p, {}, {}, &err};
    for (uint32_t m = 0; m < nm; ++m) {
        mmdx_status st = sb.visit(m, m, 0);
        if (st != MMDX_OK) return st;
    }
    p.ns = uint32_t(p.slot_top.size());
    if (p.f16 && p.ns > 65534) {
        err = "f16 morph entries need <= 65534 slots";
        return MMDX_ERR_UNSUPPORTED;
    }
    std::vector<uint32_t> cnt(nv, 0);           // CSR row length of every original vertex
    for (uint32_t s = 0; s < p.ns; ++s) {
        const uint32_t m = sb.slot_morph[s];
        for (uint32_t j = d.morph_offset[m]; j < d.morph_offset[m + 1]; ++j) ++cnt[d.morph_index[j]];
    }

    // ---- tiles: sort by (class, row length), tile-local bone lists, per-class streams -----------
    p.ntiles = (nv + kTileVerts - 1) / kTileVerts;
    p.tiles.resize(p.ntiles);
    if (p.f16) p.spos16.resize(size_t(nv) * 4); else p.spos.resize(size_t(nv) * 3);
    p.snrm.resize(size_t(nv) * 3);
    p.suv.assign(size_t(nv) * 2, 0.0f);
    p.perm.resize(nv);
    p.skin1.reserve(p.n1);
    p.skin2_ids.reserve(p.n2); p.skin2_w.reserve(p.n2);
    p.skin4_ids.reserve(size_t(p.n4) * 4); p.skin4_w.reserve(size_t(p.n4) * 4);
    std::vector<uint32_t> gs_of(nv);            // original vertex -> sorted global slot
    std::vector<int32_t> lut(nb, -1);           // global bone -> tile-local index
    std::vector<uint32_t> tile_bones, order;
    for (uint32_t t = 0; t < p.ntiles; ++t) {
        TileHdr &h = p.tiles[t];
        std::memset(&h, 0, sizeof(h));
        h.v0 = t * kTileVerts;
        h.nv = std::min(kTileVerts, nv - h.v0);
        tile_bones.clear();
        for (uint32_t l = 0; l < h.nv; ++l) {
            const uint32_t v = h.v0 + l;
            const int nids = p.cls[v] == 0 ? 1 : (p.cls[v] == 1 ? 2 : 4);
            for (int k = 0; k < nids; ++k) {
                const int32_t b = p.ids[4 * size_t(v) + k];
                if (lut[b] < 0) { lut[b] = 0; tile_bones.push_back(uint32_t(b)); }
            }
        }
        std::sort(tile_bones.begin(), tile_bones.end());
        if (tile_bones.size() > 65535) {
            err = "more than 65535 distinct bones in one vertex tile";
            return MMDX_ERR_UNSUPPORTED;
        }
        h.nbt = uint32_t(tile_bones.size());
        h.bone_off = uint32_t(p.bone_list.size());
        for (uint32_t k = 0; k < h.nbt; ++k) lut[tile_bones[k]] = int32_t(k);
        p.bone_list.insert(p.bone_list.end(), tile_bones.begin(), tile_bones.end());
        p.max_tile_bones = std::max(p.max_tile_bones, h.nbt);
        h.skin1_off = uint32_t(p.skin1.size());
        h.skin2_off = uint32_t(p.skin2_w.size());
        h.skin4_off = uint32_t(p.skin4_w.size() / 4);
        // class first (wave-uniform code paths), then by morph row length (wave-uniform gather trip
        // counts: the 64 rows of a wave-slot are padded to the longest), then original order
        order.resize(h.nv);
        for (uint32_t l = 0; l < h.nv; ++l) order[l] = l;
        // The length order zig-zags over the classes -- BDEF1 longest first, BDEF2 shortest first, BDEF4 longest
        // first -- so that the wave-slots straddling a class boundary hold rows of similar length on both sides
        // (short | short, long | long): 8-10 % fewer padded entries on the benchmark models than longest-first
        // throughout.
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const uint32_t va = h.v0 + a, vb = h.v0 + b;
            if (p.cls[va] != p.cls[vb]) return p.cls[va] < p.cls[vb];
            if (cnt[va] != cnt[vb]) return p.cls[va] == 1 ? cnt[va] < cnt[vb] : cnt[va] > cnt[vb];
            return a < b;
        });
        for (uint32_t s = 0; s < h.nv; ++s) {
            const uint32_t l = order[s], v = h.v0 + l, gs = h.v0 + s;
            const int c = p.cls[v];
            gs_of[v] = gs;
            p.perm[gs] = uint16_t(l);
            const int32_t *id = p.ids.data() + 4 * size_t(v);
            const float *w = p.wts.data() + 4 * size_t(v);
            if (c == 0) {
                p.skin1.push_back(uint16_t(lut[id[0]]));
                ++h.n1;
            } else if (c == 1) {
                p.skin2_ids.push_back(uint32_t(lut[id[0]]) | (uint32_t(lut[id[1]]) << 16));
                p.skin2_w.push_back(w[0]);
                ++h.n2;
            } else {
                for (int k = 0; k < 4; ++k) {
                    p.skin4_ids.push_back(uint16_t(lut[id[k]]));
                    p.skin4_w.push_back(w[k]);
                }
            }
            for (int k = 0; k < 3; ++k) {
                const float x = d.positions[3 * size_t(v) + k];
                if (p.f16) p.spos16[4 * size_t(gs) + k] = f32_to_f16_rne(x);
                else p.spos[3 * size_t(gs) + k] = x;
                p.snrm[3 * size_t(gs) + k] = d.normals[3 * size_t(v) + k];
            }
            if (p.f16) p.spos16[4 * size_t(gs) + 3] = 0;
            if (d.uvs) {
                p.suv[2 * size_t(gs)] = d.uvs[2 * size_t(v)];
                p.suv[2 * size_t(gs) + 1] = d.uvs[2 * size_t(v) + 1];
            }
        }
        for (uint32_t k = 0; k < h.nbt; ++k) lut[tile_bones[k]] = -1;
    }

    // ---- morph gather table: sliced ELL -----------------------------------------------------------
    // One slice per wave-slot (64 consecutive sorted slots).  Entry j of the 64 rows is stored
    // contiguously (slice_base + j*64 + lane): one coalesced 1 KiB (f32) load per wave and j, all j
    // independent of each other.  Rows are padded to the slice's longest row with the dummy entry
    // {0,0,0, slot = NS}; slot NS always carries weight 0, so a pad adds (+0)*0 = +0 to a sum that is
    // never -0: bit-exact no-op.  Per-row entry order = the reference's accumulation order.
    const uint32_t nslices = p.ntiles * kSlicesPerTile;
    p.ell.assign(size_t(nslices) * 2, 0);
    uint64_t total = 0;
    uint32_t tile_entries = 0;
    for (uint32_t sl = 0; sl < nslices; ++sl) {
        const uint32_t g0 = sl * 64, g1 = std::min(g0 + 64, nv);
        uint32_t len = 0;
        for (uint32_t g = g0; g < g1; ++g) {
            const uint32_t t = g / kTileVerts;
            len = std::max(len, cnt[p.tiles[t].v0 + p.perm[g]]);
        }
        p.ell[2 * size_t(sl)] = uint32_t(total);
        p.ell[2 * size_t(sl) + 1] = len;
        total += uint64_t(len) * 64;
        tile_entries += len * 64;
        if ((sl + 1) % kSlicesPerTile == 0) {
            p.max_tile_entries = std::max(p.max_tile_entries, tile_entries);
            tile_entries = 0;
        }
        if (total >= (1ull << 28)) {      // the kernels address entries with 32-bit byte offsets (16 B each)
            err = "morph gather table too large";
            return MMDX_ERR_UNSUPPORTED;
        }
    }
    p.ne = uint32_t(total);
    p.ne_real = 0;
    if (p.f16) {
        p.entries16.assign(size_t(p.ne) * 4, 0);
        for (size_t e = 0; e < p.ne; ++e) p.entries16[4 * e + 3] = uint16_t(p.ns);
    } else {
        p.entries.assign(size_t(p.ne) * 4, 0.0f);
        const uint32_t dummy = p.ns;
        for (size_t e = 0; e < p.ne; ++e) std::memcpy(&p.entries[4 * e + 3], &dummy, 4);
    }
    std::vector<uint32_t> cursor(nv, 0);        // next row position j of every sorted slot
    for (uint32_t s = 0; s < p.ns; ++s) {
        const uint32_t m = sb.slot_morph[s];
        for (uint32_t j = d.morph_offset[m]; j < d.morph_offset[m + 1]; ++j) {
            const uint32_t g = gs_of[d.morph_index[j]];
            const size_t at = size_t(p.ell[2 * size_t(g >> 6)]) + size_t(cursor[g]++) * 64 + (g & 63);
            ++p.ne_real;
            const float *o = d.morph_value + 3 * size_t(j);
            for (int c = 0; c < 3; ++c) {
                uint32_t bits;
                std::memcpy(&bits, o + c, 4);
                // inf / NaN, or a finite f32 that rounds to inf as binary16
                if ((bits & 0x7f800000u) == 0x7f800000u ||
                    (p.f16 && (f32_to_f16_rne(o[c]) & 0x7c00u) == 0x7c00u))
                    p.finite_offsets = false;
            }
            if (p.f16) {
                uint16_t *e = p.entries16.data() + 4 * at;
                e[0] = f32_to_f16_rne(o[0]); e[1] = f32_to_f16_rne(o[1]); e[2] = f32_to_f16_rne(o[2]);
                e[3] = uint16_t(s);
            } else {
                float *e = p.entries.data() + 4 * at;
                e[0] = o[0]; e[1] = o[1]; e[2] = o[2];
                std::memcpy(e + 3, &s, 4);
            }
        }
    }

    // A model without any vertex-morph slot never runs a morph pass on the device.  The reference
    // still evaluates `coordinate + vertex_images_[i]` with a +0 image (poser_impl.inl:407), which
    // turns a -0 coordinate into +0; bake that in so the no-morph kernel variant stays bit-exact.
    if (p.ns == 0) {
        for (float &x : p.spos) {
            uint32_t bits;
            std::memcpy(&bits, &x, 4);
            if (bits == 0x80000000u) x = 0.0f;  // (-0) + (+0) = +0; every other value is unchanged
        }
        for (size_t i = 0; i < p.spos16.size(); ++i)
            if (p.spos16[i] == 0x8000u) p.spos16[i] = 0;
    }
    return MMDX_OK;
}

}  // namespace mmdx
