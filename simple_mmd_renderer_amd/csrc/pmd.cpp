// pmd.cpp -- PMD 1.0 (the older MMD model format) -> the same flat model the PMX parser produces, so that
// mmdx_pmx_get_model_desc / _get_skeleton_desc / _get_arrays / _get_name serve it unchanged (SURVEY.md section
// 8f-1, "a cheap follow-on").  From-scratch host code (C++17, no HIP).
//
// Reference semantics followed (L/ = 3rd_party/libmmd/include/mmd/): PmdReader::ReadModel,
// L/reader/pmd_reader_impl.inl:16-566, record layouts L/reader/interprete/pmd_types.inl:17-125.
//   vertices   38-byte records; every vertex is BDEF2 with weight = byte * 0.01f, bone ids int16
//              sign-extended                                         pmd_reader_impl.inl:35-51
//   bones      39-byte records; parent == own index means none; type 2 (IK) or membership in the IK
//              list sets has-IK (transform level 1); type 5 = append-rotate from bone `ik_number`,
//              ratio 1, level 2; type 9 = append-rotate from `child_id`, ratio ik_number * 0.01f
//                                                                    :187-262
//   IK list    sorted by the first chain bone; the first record of an IK bone configures that bone,
//              every further record of the same bone APPENDS a copy of it ("[IK]" + name, parent =
//              the original) carrying that chain; angle limit = file value * 4; links named 左ひざ /
//              右ひざ get the knee limit x in [-pi, -0.5 deg]               :176-186, :264-327
//   levels     every original bone is raised to the highest level among its ancestors   :331-352
//   morphs     all vertex morphs; indices of the non-base morphs go through the base ("system",
//              category 0) morph's vertex list                           :354-392
//   the reader ends with Model::Normalize (requested through MMDX_CREATE_NORMALIZE)      :557
// Display lists, English names, toon textures, rigid bodies and joints follow and are not read.
//
// Reference defect worth knowing: on Linux libmmd's Shift-JIS conversion prepends a byte-order mark
// (L/util/dwarf_impl.inl:221-230), so its knee-name test never matches there and the limits are never set;
// this loader compares the decoded names (the evident intent), as csrc/vmd.cpp does for motion tracks.
#include <iconv.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <set>

#include "../../include/mmdx.h"
#include "error.hpp"
#include "pmx.hpp"

std::string mmdx::sjis_field_to_utf8(const uint8_t *p, size_t field_size) {
    size_t n = 0;
    while (n < field_size && p[n]) ++n;
    std::string s(reinterpret_cast<const char *>(p), n);
    if (s.empty()) return s;
    iconv_t cd = iconv_open("UTF-8", "SHIFT-JIS");
    if (cd == iconv_t(-1)) return s;
    std::vector<char> in(s.begin(), s.end()), out(s.size() * 4 + 4);
    char *pi = in.data(), *po = out.data();
    size_t ni = in.size(), no = out.size();
    const size_t r = iconv(cd, &pi, &ni, &po, &no);
    iconv_close(cd);
    if (r == size_t(-1)) return s;
    return std::string(out.data(), out.size() - no);
}

namespace {

struct PmdError {
    std::string what;
};

class Reader {
public:
    Reader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    size_t pos() const { return at_; }
    size_t left() const { return n_ - at_; }
    const uint8_t *need(size_t k, const char *what) {
        if (k > n_ - at_) throw PmdError{std::string("PMD: file ends inside ") + what};
        const uint8_t *q = p_ + at_;
        at_ += k;
        return q;
    }
    template <typename T>
    T get(const char *what) {
        T v;
        std::memcpy(&v, need(sizeof(T), what), sizeof(T));
        return v;
    }

private:
    const uint8_t *p_;
    size_t n_, at_ = 0;
};

template <typename T>
T rd(const uint8_t *p) {
    T v;
    std::memcpy(&v, p, sizeof(T));
    return v;
}

struct RawIk {
    int32_t bone, target;
    uint32_t loop;
    float angle;
    std::vector<uint32_t> chain;
};

void parse_pmd(mmdx_pmx_s &m, const uint8_t *data, size_t size) {
    Reader c(data, size);
    const uint8_t *h = c.need(283, "the header");
    if (std::memcmp(h, "Pmd", 3) != 0 || rd<float>(h + 3) != 1.0f) throw PmdError{"PMD: not a PMD 1.0 file"};
    m.name = mmdx::sjis_field_to_utf8(h + 7, 20);

    // ---- vertices ---------------------------------------------------------------------------------
    const uint32_t nv = c.get<uint32_t>("the vertex count");
    const uint8_t *v = c.need(size_t(nv) * 38, "the vertices");
    m.positions.resize(size_t(nv) * 3); m.normals.resize(size_t(nv) * 3); m.uvs.resize(size_t(nv) * 2);
    m.skin_type.assign(nv, MMDX_SKIN_BDEF2);
    m.bone_ids.assign(size_t(nv) * 4, 0); m.bone_weights.assign(size_t(nv) * 4, 0.f);
    m.sdef.assign(size_t(nv) * 9, 0.f); m.edge_scale.resize(nv);
    for (uint32_t i = 0; i < nv; ++i, v += 38) {
        std::memcpy(&m.positions[3 * size_t(i)], v, 12);
        std::memcpy(&m.normals[3 * size_t(i)], v + 12, 12);
        std::memcpy(&m.uvs[2 * size_t(i)], v + 24, 8);
        m.bone_ids[4 * size_t(i)] = rd<int16_t>(v + 32);
        m.bone_ids[4 * size_t(i) + 1] = rd<int16_t>(v + 34);
        m.bone_weights[4 * size_t(i)] = float(int(v[36])) * 0.01f;
        m.edge_scale[i] = v[37] > 0 ? 0.0f : 1.0f;
    }

    // ---- triangles, materials ----------------------------------------------------------------------
    const uint32_t nidx = c.get<uint32_t>("the index count");
    const uint8_t *ix = c.need(size_t(nidx) * 2, "the indices");
    m.triangles.resize(nidx / 3 * 3);
    for (size_t i = 0; i < m.triangles.size(); ++i) m.triangles[i] = rd<uint16_t>(ix + 2 * i);
    const uint32_t nmat = c.get<uint32_t>("the material count");
    const uint8_t *mt = c.need(size_t(nmat) * 70, "the materials");
    for (uint32_t i = 0; i < nmat; ++i, mt += 70) {
        m.material_index_count.push_back(rd<uint32_t>(mt + 46) / 3 * 3);
        m.material_names.push_back("material" + std::to_string(i));
        m.textures.push_back(mmdx::sjis_field_to_utf8(mt + 50, 20));
    }

    // ---- bones + IK list ------------------------------------------------------------------------------
    const uint32_t nb0 = c.get<uint16_t>("the bone count");
    const uint8_t *bn = c.need(size_t(nb0) * 39, "the bones");
    const uint32_t nik = c.get<uint16_t>("the IK count");
    std::vector<RawIk> iks(nik);
    std::set<int32_t> ik_bones;
    for (RawIk &k : iks) {
        const uint8_t *p = c.need(11, "an IK record");
        k.bone = rd<int16_t>(p); k.target = rd<int16_t>(p + 2);
        const uint32_t len = p[4];
        k.loop = rd<uint16_t>(p + 5); k.angle = rd<float>(p + 7);
        const uint8_t *ch = c.need(size_t(len) * 2, "an IK chain");
        for (uint32_t j = 0; j < len; ++j) k.chain.push_back(rd<uint16_t>(ch + 2 * j));
        ik_bones.insert(k.bone);
    }
    // the reference sorts with std::sort on the first chain bone; a stable sort gives the same order
    // whenever those keys are distinct (ties are unspecified upstream)
    std::stable_sort(iks.begin(), iks.end(), [](const RawIk &a, const RawIk &b) {
        return (a.chain.empty() ? 0u : a.chain[0]) < (b.chain.empty() ? 0u : b.chain[0]);
    });

    struct Bone {
        std::string name;
        float pos[3];
        int32_t parent, level, append_parent;
        float append_ratio;
        uint16_t flags;
    };
    std::vector<Bone> bones(nb0);
    for (uint32_t i = 0; i < nb0; ++i, bn += 39) {
        Bone &b = bones[i];
        b.name = mmdx::sjis_field_to_utf8(bn, 20);
        const int32_t parent = rd<int16_t>(bn + 20), child = rd<int16_t>(bn + 22), ik_number = rd<int16_t>(bn + 25);
        const uint8_t type = bn[24];
        std::memcpy(b.pos, bn + 27, 12);
        b.parent = (parent != int32_t(i)) ? parent : -1;       // out-of-range values mean "none" downstream
        b.level = 0; b.append_parent = -1; b.append_ratio = 0.f; b.flags = 0x0002 | 0x0010 | 0x0001;
        const bool has_ik = type == 2 || ik_bones.count(int32_t(i)) > 0;
        if (has_ik) b.flags |= kPmxBoneHasIk | 0x0004;
        if (type == 1) b.flags |= 0x0004;
        if (type != 6 && type != 7 && type != 9) b.flags |= 0x0008;
        if (type == 5) {
            b.flags |= kPmxBoneAppendRotate; b.append_parent = ik_number; b.append_ratio = 1.0f; b.level = 2;
        } else if (type == 9) {
            b.flags |= kPmxBoneAppendRotate; b.append_parent = child; b.append_ratio = float(ik_number) * 0.01f;
        }
        if (type == 8) b.flags |= 0x0400;
        if (has_ik) b.level = 1;
    }
    // IK records -> bones; a second record of the same bone becomes an appended copy of that bone
    struct IkOut {
        int32_t target;
        uint32_t loop;
        float angle;
        std::vector<uint32_t> chain;
        bool set = false;
    };
    std::vector<IkOut> ik_of(nb0);
    for (uint32_t i = 0; i < nb0; ++i) {
        if (!ik_bones.count(int32_t(i))) continue;
        uint32_t seen = 0;
        for (const RawIk &k : iks) {
            if (k.bone != int32_t(i)) continue;
            uint32_t dst = i;
            if (seen++) {
                Bone copy = bones[i];
                copy.name = "[IK]" + bones[i].name;
                copy.parent = int32_t(i);
                copy.flags = uint16_t((copy.flags & ~uint16_t(0x0009)) | kPmxBoneHasIk);
                bones.push_back(copy);
                ik_of.emplace_back();
                dst = uint32_t(bones.size() - 1);
            }
            ik_of[dst].target = k.target; ik_of[dst].loop = k.loop; ik_of[dst].angle = k.angle * 4.0f;
            ik_of[dst].chain = k.chain; ik_of[dst].set = true;
        }
    }
    // transform levels: every ORIGINAL bone takes the highest level among its ancestors
    const uint32_t nb = uint32_t(bones.size());
    for (uint32_t pass = 0; pass < nb0; ++pass) {
        bool stable = true;
        for (uint32_t j = 0; j < nb0; ++j) {
            int32_t level = bones[j].level;
            uint32_t steps = 0;
            for (int32_t p = bones[j].parent; p >= 0 && uint32_t(p) < nb0 && steps <= nb; p = bones[p].parent, ++steps)
                if (level < bones[p].level) { level = bones[p].level; stable = false; }
            if (steps > nb) throw PmdError{"PMD: the bone parents form a cycle"};
            bones[j].level = level;
        }
        if (stable) break;
    }
    const float pi_f = float(3.141592653589793238462643383279502884);
    m.bone_pos.resize(size_t(nb) * 3); m.bone_parent.resize(nb); m.bone_level.resize(nb); m.bone_flags.resize(nb);
    m.append_parent.assign(nb, -1); m.append_ratio.assign(nb, 0.f);
    m.ik_target.assign(nb, -1); m.ik_loop.assign(nb, 0); m.ik_angle.assign(nb, 0.f);
    m.ik_link_off.assign(size_t(nb) + 1, 0);
    for (uint32_t i = 0; i < nb; ++i) {
        const Bone &b = bones[i];
        m.bone_names.push_back(b.name);
        std::memcpy(&m.bone_pos[3 * size_t(i)], b.pos, 12);
        m.bone_parent[i] = (b.parent >= 0 && uint32_t(b.parent) < nb) ? b.parent : -1;
        m.bone_level[i] = b.level; m.bone_flags[i] = b.flags;
        m.append_parent[i] = b.append_parent; m.append_ratio[i] = b.append_ratio;
        if (b.flags & kPmxBoneHasIk) {
            // a type-2 bone without an IK record keeps libmmd's zero-initialised target (bone 0), no links
            m.ik_target[i] = ik_of[i].set ? ik_of[i].target : 0;
            m.ik_loop[i] = ik_of[i].set ? int32_t(ik_of[i].loop) : 0;
            m.ik_angle[i] = ik_of[i].set ? ik_of[i].angle : 0.f;
            for (uint32_t lb : ik_of[i].chain) {
                m.ik_link_bone.push_back(int32_t(lb));
                const bool knee = lb < nb && (bones[lb].name == "\xE5\xB7\xA6\xE3\x81\xB2\xE3\x81\x96" ||   // 左ひざ
                                              bones[lb].name == "\xE5\x8F\xB3\xE3\x81\xB2\xE3\x81\x96");    // 右ひざ
                m.ik_link_limited.push_back(knee ? 1 : 0);
                const float lo[3] = {knee ? -pi_f : 0.f, 0.f, 0.f};
                const float hi[3] = {knee ? -0.5f / 180.0f * pi_f : 0.f, 0.f, 0.f};
                m.ik_link_lo.insert(m.ik_link_lo.end(), lo, lo + 3);
                m.ik_link_hi.insert(m.ik_link_hi.end(), hi, hi + 3);
            }
        }
        m.ik_link_off[size_t(i) + 1] = uint32_t(m.ik_link_bone.size());
    }

    // ---- morphs ---------------------------------------------------------------------------------------
    const uint32_t nm = c.get<uint16_t>("the morph count");
    m.morph_offset.push_back(0);
    int64_t base = -1;
    for (uint32_t i = 0; i < nm; ++i) {
        const uint8_t *p = c.need(25, "a morph header");
        m.morph_names.push_back(mmdx::sjis_field_to_utf8(p, 20));
        const uint32_t cnt = rd<uint32_t>(p + 20);
        const uint8_t category = p[24];
        if (category == 0) base = i;                           // the last "system" morph wins, as upstream
        m.morph_type.push_back(MMDX_MORPH_VERTEX);
        m.morph_panel.push_back(category);
        const uint8_t *e = c.need(size_t(cnt) * 16, "the morph entries");
        for (uint32_t j = 0; j < cnt; ++j, e += 16) {
            m.morph_index.push_back(rd<uint32_t>(e));
            float off[3];
            std::memcpy(off, e + 4, 12);
            m.morph_value.insert(m.morph_value.end(), off, off + 3);
            const float rot[4] = {0.f, 0.f, 0.f, 1.f};
            m.morph_rotation.insert(m.morph_rotation.end(), rot, rot + 4);
        }
        m.morph_offset.push_back(uint32_t(m.morph_index.size()));
    }
    if (base >= 0) {
        const uint32_t b0 = m.morph_offset[size_t(base)], bcount = m.morph_offset[size_t(base) + 1] - b0;
        const std::vector<uint32_t> base_index(m.morph_index.begin() + b0, m.morph_index.begin() + b0 + bcount);
        for (uint32_t i = 0; i < nm; ++i) {
            if (int64_t(i) == base) continue;
            for (uint32_t e = m.morph_offset[i]; e < m.morph_offset[i + 1]; ++e) {
                if (m.morph_index[e] >= bcount)
                    throw PmdError{"PMD: morph " + std::to_string(i) + " refers to entry " + std::to_string(m.morph_index[e]) +
                                   " of a base morph with " + std::to_string(bcount) + " entries"};
                m.morph_index[e] = base_index[m.morph_index[e]];
            }
        }
    }

    m.info.n_vertices = nv; m.info.n_indices = uint32_t(m.triangles.size());
    m.info.n_textures = 0; m.info.n_materials = nmat;
    m.info.n_bones = nb; m.info.n_morphs = nm;
    m.info.n_morph_entries = uint32_t(m.morph_index.size());
    m.info.extra_uv = 0; m.info.utf8 = 0;
    m.info.index_width[0] = 2; m.info.index_width[3] = 2;
    m.info.bytes_consumed = c.pos();
}

}  // namespace

extern "C" {

mmdx_status mmdx_pmd_parse(const void *data, size_t size, mmdx_pmx_t *out) {
    if (!data || !out) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "data / out is NULL");
    *out = nullptr;
    mmdx_pmx_s *m = new (std::nothrow) mmdx_pmx_s;
    if (!m) return mmdx::fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->info.struct_size = sizeof(mmdx_pmx_info);
    try {
        parse_pmd(*m, static_cast<const uint8_t *>(data), size);
    } catch (const PmdError &e) {
        delete m;
        return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, e.what);
    } catch (const std::bad_alloc &) {
        delete m;
        return mmdx::fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while parsing");
    }
    *out = m;
    return MMDX_OK;
}

mmdx_status mmdx_pmd_load_file(const char *path, mmdx_pmx_t *out) {
    if (!path || !out) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "path / out is NULL");
    *out = nullptr;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t got;
    try {
        while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    } catch (const std::bad_alloc &) {
        std::fclose(f);
        return mmdx::fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while reading the file");
    }
    std::fclose(f);
    return mmdx_pmd_parse(buf.data(), buf.size(), out);
}

}  // extern "C"
