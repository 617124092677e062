// pmx.cpp -- PMX 2.0 model file -> the flat model description mmdx_model_create() consumes.
// From-scratch host code (C++17, no HIP); the first "next" row after the hot path.
//
// Covers what the reference's loader produces for the deformation path, with the same observable
// semantics (reference file:line, L/ = 3rd_party/libmmd/include/mmd/):
//   header / globals / index sizes           L/reader/pmx_reader_impl.inl:21-43
//   vertex block (deform types, SDEF params) L/reader/pmx_reader_impl.inl:50-102
//   triangles, textures, materials           L/reader/pmx_reader_impl.inl:104-190 (kept: index counts)
//   bones (rest position, parent, flags)     L/reader/pmx_reader_impl.inl:192-264
//   morphs (group/vertex/bone/uv/material)   L/reader/pmx_reader_impl.inl:266-357
//   index widths: 1- and 2-byte indices are ZERO-extended, 4-byte ones sign-extended
//                                            L/util/dwarf_impl.inl:84-104
//   text: int32 byte length + UTF-16LE or UTF-8 payload      L/util/dwarf_impl.inl:117-130
// Display frames, rigid bodies and joints follow the morphs in the file and are not read (the path
// does not use them).  Model::Normalize (L/model/model_impl.inl:406-452), which the reference's reader
// calls at the end (pmx_reader_impl.inl:441), is applied by mmdx_model_create when the description
// carries MMDX_CREATE_NORMALIZE -- mmdx_pmx_get_model_desc sets that flag.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mmdx.h"
#include "error.hpp"
#include "pmx.hpp"

namespace {

struct ParseError {
    std::string what;
};

class Cursor {
public:
    Cursor(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    size_t pos() const { return at_; }
    void need(size_t k, const char *what) const {
        if (k > n_ - at_) throw ParseError{std::string("PMX: file ends inside ") + what};
    }
    template <typename T>
    T get(const char *what) {
        need(sizeof(T), what);
        T v;
        std::memcpy(&v, p_ + at_, sizeof(T));
        at_ += sizeof(T);
        return v;
    }
    void floats(float *dst, size_t count, const char *what) {
        need(count * 4, what);
        std::memcpy(dst, p_ + at_, count * 4);
        at_ += count * 4;
    }
    void skip(size_t k, const char *what) {
        need(k, what);
        at_ += k;
    }
    // PMX index field of `width` bytes with libmmd's extension rule
    int32_t index(uint32_t width, const char *what) {
        switch (width) {
        case 1: return int32_t(get<uint8_t>(what));
        case 2: return int32_t(get<uint16_t>(what));
        case 4: return get<int32_t>(what);
        default: throw ParseError{"PMX: index width must be 1, 2 or 4"};
        }
    }
    std::string text(bool utf8, const char *what) {
        const int32_t len = get<int32_t>(what);
        if (len < 0) throw ParseError{std::string("PMX: negative text length in ") + what};
        need(size_t(len), what);
        std::string out;
        if (utf8) {
            out.assign(reinterpret_cast<const char *>(p_ + at_), size_t(len));
        } else {  // UTF-16LE -> UTF-8
            for (int32_t i = 0; i + 1 < len; i += 2) {
                uint32_t c = uint32_t(p_[at_ + i]) | (uint32_t(p_[at_ + i + 1]) << 8);
                if (c >= 0xD800 && c < 0xDC00 && i + 3 < len) {
                    const uint32_t lo = uint32_t(p_[at_ + i + 2]) | (uint32_t(p_[at_ + i + 3]) << 8);
                    if (lo >= 0xDC00 && lo < 0xE000) {
                        c = 0x10000 + ((c - 0xD800) << 10) + (lo - 0xDC00);
                        i += 2;
                    }
                }
                if (c < 0x80) out.push_back(char(c));
                else if (c < 0x800) { out.push_back(char(0xC0 | (c >> 6))); out.push_back(char(0x80 | (c & 0x3F))); }
                else if (c < 0x10000) {
                    out.push_back(char(0xE0 | (c >> 12))); out.push_back(char(0x80 | ((c >> 6) & 0x3F)));
                    out.push_back(char(0x80 | (c & 0x3F)));
                } else {
                    out.push_back(char(0xF0 | (c >> 18))); out.push_back(char(0x80 | ((c >> 12) & 0x3F)));
                    out.push_back(char(0x80 | ((c >> 6) & 0x3F))); out.push_back(char(0x80 | (c & 0x3F)));
                }
            }
        }
        at_ += size_t(len);
        return out;
    }

private:
    const uint8_t *p_;
    size_t n_, at_ = 0;
};

enum : uint16_t {
    kBoneChildUseId = 0x0001, kBoneHasIk = 0x0020, kBoneAppendRotate = 0x0100,
    kBoneAppendTranslate = 0x0200, kBoneAxisFixed = 0x0400, kBoneLocalAxis = 0x0800,
    kBoneReceiveTransform = 0x2000
};

}  // namespace


namespace {

void parse(mmdx_pmx_s &m, const uint8_t *data, size_t size) {
    Cursor c(data, size);
    char magic[4];
    c.need(4, "the header");
    for (char &ch : magic) ch = char(c.get<uint8_t>("the header"));
    const float version = c.get<float>("the header");
    const uint8_t nglobals = c.get<uint8_t>("the header");
    if (std::memcmp(magic, "PMX ", 4) != 0 || version != 2.0f || nglobals != 8)
        throw ParseError{"PMX: not a PMX 2.0 file (magic / version / globals count)"};
    const bool utf8 = c.get<uint8_t>("the globals") > 0;
    const uint32_t extra_uv = c.get<uint8_t>("the globals");
    const uint32_t w_vertex = c.get<uint8_t>("the globals"), w_texture = c.get<uint8_t>("the globals"),
                   w_material = c.get<uint8_t>("the globals"), w_bone = c.get<uint8_t>("the globals"),
                   w_morph = c.get<uint8_t>("the globals"), w_rigid = c.get<uint8_t>("the globals");
    if (extra_uv > 4) throw ParseError{"PMX: more than 4 additional UV sets"};
    for (uint32_t w : {w_vertex, w_texture, w_material, w_bone, w_morph, w_rigid})
        if (w != 1 && w != 2 && w != 4) throw ParseError{"PMX: index width must be 1, 2 or 4"};
    m.info.utf8 = utf8; m.info.extra_uv = extra_uv;
    m.info.index_width[0] = uint8_t(w_vertex); m.info.index_width[1] = uint8_t(w_texture);
    m.info.index_width[2] = uint8_t(w_material); m.info.index_width[3] = uint8_t(w_bone);
    m.info.index_width[4] = uint8_t(w_morph); m.info.index_width[5] = uint8_t(w_rigid);
    m.name = c.text(utf8, "the model name");
    m.name_en = c.text(utf8, "the model name");
    c.text(utf8, "the model comment");
    c.text(utf8, "the model comment");

    // ---- vertices ---------------------------------------------------------------------------
    const int32_t nv = c.get<int32_t>("the vertex count");
    if (nv < 0) throw ParseError{"PMX: negative vertex count"};
    c.need(size_t(nv) * 38, "the vertex block");     // 32 + type + >= 1 index + edge: cheap sanity bound
    m.positions.resize(size_t(nv) * 3); m.normals.resize(size_t(nv) * 3); m.uvs.resize(size_t(nv) * 2);
    m.skin_type.resize(nv); m.bone_ids.assign(size_t(nv) * 4, 0); m.bone_weights.assign(size_t(nv) * 4, 0.f);
    m.sdef.assign(size_t(nv) * 9, 0.f); m.edge_scale.resize(nv);
    for (int32_t i = 0; i < nv; ++i) {
        c.floats(&m.positions[3 * size_t(i)], 3, "a vertex");
        c.floats(&m.normals[3 * size_t(i)], 3, "a vertex");
        c.floats(&m.uvs[2 * size_t(i)], 2, "a vertex");
        c.skip(size_t(extra_uv) * 16, "a vertex");
        const int8_t t = c.get<int8_t>("a vertex");
        int32_t *id = &m.bone_ids[4 * size_t(i)];
        float *w = &m.bone_weights[4 * size_t(i)];
        m.skin_type[i] = t;
        switch (t) {
        case MMDX_SKIN_BDEF1:
            id[0] = c.index(w_bone, "a vertex");
            break;
        case MMDX_SKIN_BDEF2:
            id[0] = c.index(w_bone, "a vertex"); id[1] = c.index(w_bone, "a vertex");
            w[0] = c.get<float>("a vertex");
            break;
        case MMDX_SKIN_BDEF4:
            for (int k = 0; k < 4; ++k) id[k] = c.index(w_bone, "a vertex");
            c.floats(w, 4, "a vertex");
            break;
        case MMDX_SKIN_SDEF:
            id[0] = c.index(w_bone, "a vertex"); id[1] = c.index(w_bone, "a vertex");
            w[0] = c.get<float>("a vertex");
            c.floats(&m.sdef[9 * size_t(i)], 9, "a vertex");
            break;
        default:
            throw ParseError{"PMX: vertex " + std::to_string(i) + " has deform type " + std::to_string(int(t)) +
                             " (PMX 2.0 knows BDEF1/2/4 and SDEF)"};
        }
        m.edge_scale[i] = c.get<float>("a vertex");
    }

    // ---- triangles ----------------------------------------------------------------------------
    const int32_t nidx = c.get<int32_t>("the index count");
    if (nidx < 0) throw ParseError{"PMX: negative index count"};
    c.need(size_t(nidx / 3) * 3 * w_vertex, "the index block");
    m.triangles.resize(size_t(nidx / 3) * 3);
    for (uint32_t &t : m.triangles) t = uint32_t(c.index(w_vertex, "the index block"));

    // ---- textures, materials --------------------------------------------------------------------
    const int32_t ntex = c.get<int32_t>("the texture count");
    if (ntex < 0) throw ParseError{"PMX: negative texture count"};
    for (int32_t i = 0; i < ntex; ++i) m.textures.push_back(c.text(utf8, "a texture path"));
    const int32_t nmat = c.get<int32_t>("the material count");
    if (nmat < 0) throw ParseError{"PMX: negative material count"};
    for (int32_t i = 0; i < nmat; ++i) {
        m.material_names.push_back(c.text(utf8, "a material"));
        c.text(utf8, "a material");
        c.skip(65, "a material");                    // colours, flags, edge (packed, 65 bytes)
        c.index(w_texture, "a material");
        c.index(w_texture, "a material");
        c.skip(1, "a material");                     // sphere mode
        const bool shared_toon = c.get<uint8_t>("a material") > 0;
        if (shared_toon) c.skip(1, "a material"); else c.index(w_texture, "a material");
        c.text(utf8, "a material");                  // memo
        const int32_t cnt = c.get<int32_t>("a material");
        m.material_index_count.push_back(cnt < 0 ? 0u : uint32_t(cnt));
    }

    // ---- bones ----------------------------------------------------------------------------------
    const int32_t nb = c.get<int32_t>("the bone count");
    if (nb < 0) throw ParseError{"PMX: negative bone count"};
    c.need(size_t(nb) * (8 + 12 + w_bone + 4 + 2 + w_bone), "the bone block");   // smallest possible bone record
    m.bone_pos.resize(size_t(nb) * 3); m.bone_parent.resize(nb); m.bone_level.resize(nb); m.bone_flags.resize(nb);
    m.append_parent.assign(nb, -1); m.append_ratio.assign(nb, 0.f);
    m.ik_target.assign(nb, -1); m.ik_loop.assign(nb, 0); m.ik_angle.assign(nb, 0.f);
    m.ik_link_off.assign(size_t(nb) + 1, 0);
    for (int32_t i = 0; i < nb; ++i) {
        m.bone_names.push_back(c.text(utf8, "a bone"));
        c.text(utf8, "a bone");
        c.floats(&m.bone_pos[3 * size_t(i)], 3, "a bone");
        const int32_t parent = c.index(w_bone, "a bone");
        m.bone_parent[i] = (parent >= 0 && parent < nb) ? parent : -1;   // anything else means "none"
        m.bone_level[i] = c.get<int32_t>("a bone");
        const uint16_t flags = c.get<uint16_t>("a bone");
        m.bone_flags[i] = flags;
        if (flags & kBoneChildUseId) c.index(w_bone, "a bone"); else c.skip(12, "a bone");
        if (flags & (kBoneAppendRotate | kBoneAppendTranslate)) {
            m.append_parent[i] = c.index(w_bone, "a bone");   // range-checked by mmdx_skeleton_create ("outside = none")
            m.append_ratio[i] = c.get<float>("a bone");
        }
        if (flags & kBoneAxisFixed) c.skip(12, "a bone");
        if (flags & kBoneLocalAxis) c.skip(24, "a bone");
        if (flags & kBoneReceiveTransform) c.skip(4, "a bone");
        if (flags & kBoneHasIk) {
            m.ik_target[i] = c.index(w_bone, "a bone");
            m.ik_loop[i] = c.get<int32_t>("a bone");
            m.ik_angle[i] = c.get<float>("a bone");
            const int32_t links = c.get<int32_t>("a bone");
            if (links < 0) throw ParseError{"PMX: negative IK link count"};
            c.need(size_t(links) * (w_bone + 1), "the IK links");
            for (int32_t l = 0; l < links; ++l) {
                m.ik_link_bone.push_back(c.index(w_bone, "an IK link"));
                const bool limited = c.get<int8_t>("an IK link") != 0;
                m.ik_link_limited.push_back(limited ? 1 : 0);
                float lim[6] = {0, 0, 0, 0, 0, 0};
                if (limited) c.floats(lim, 6, "an IK link");
                m.ik_link_lo.insert(m.ik_link_lo.end(), lim, lim + 3);
                m.ik_link_hi.insert(m.ik_link_hi.end(), lim + 3, lim + 6);
            }
        }
        m.ik_link_off[size_t(i) + 1] = uint32_t(m.ik_link_bone.size());
    }

    // ---- morphs ---------------------------------------------------------------------------------
    const int32_t nm = c.get<int32_t>("the morph count");
    if (nm < 0) throw ParseError{"PMX: negative morph count"};
    m.morph_offset.push_back(0);
    for (int32_t i = 0; i < nm; ++i) {
        m.morph_names.push_back(c.text(utf8, "a morph"));
        c.text(utf8, "a morph");
        m.morph_panel.push_back(c.get<uint8_t>("a morph"));
        const uint8_t type = c.get<uint8_t>("a morph");
        const int32_t count = c.get<int32_t>("a morph");
        if (count < 0) throw ParseError{"PMX: negative morph offset count"};
        m.morph_type.push_back(type);
        for (int32_t j = 0; j < count; ++j) {
            float v[3] = {0.f, 0.f, 0.f}, rot[4] = {0.f, 0.f, 0.f, 1.f};
            int32_t idx = 0;
            if (type == MMDX_MORPH_GROUP) {
                idx = c.index(w_morph, "a group morph");
                v[0] = c.get<float>("a group morph");
            } else if (type == MMDX_MORPH_VERTEX) {
                idx = c.index(w_vertex, "a vertex morph");
                c.floats(v, 3, "a vertex morph");
            } else if (type == MMDX_MORPH_BONE) {
                idx = c.index(w_bone, "a bone morph");
                c.floats(v, 3, "a bone morph");      // translation
                c.floats(rot, 4, "a bone morph");    // rotation quaternion: consumed by the bone solve only
            } else if (type >= MMDX_MORPH_UV && type <= 7) {
                idx = c.index(w_vertex, "a uv morph");
                c.floats(v, 3, "a uv morph");
                c.skip(4, "a uv morph");
            } else if (type == MMDX_MORPH_MATERIAL) {
                idx = c.index(w_material, "a material morph");
                c.skip(113, "a material morph");
            } else {
                throw ParseError{"PMX: morph " + std::to_string(i) + " has unknown type " + std::to_string(int(type))};
            }
            m.morph_index.push_back(uint32_t(idx));
            m.morph_value.insert(m.morph_value.end(), v, v + 3);
            m.morph_rotation.insert(m.morph_rotation.end(), rot, rot + 4);
        }
        m.morph_offset.push_back(uint32_t(m.morph_index.size()));
    }

    m.info.n_vertices = uint32_t(nv); m.info.n_indices = uint32_t(m.triangles.size());
    m.info.n_textures = uint32_t(ntex); m.info.n_materials = uint32_t(nmat);
    m.info.n_bones = uint32_t(nb); m.info.n_morphs = uint32_t(nm);
    m.info.n_morph_entries = uint32_t(m.morph_index.size());
    m.info.bytes_consumed = c.pos();
}

mmdx_status pmx_fail(mmdx_status st, const std::string &msg) { return mmdx::fail(st, msg); }

}  // namespace

extern "C" {

mmdx_status mmdx_pmx_parse(const void *data, size_t size, mmdx_pmx_t *out) {
    if (!data || !out) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "data / out is NULL");
    *out = nullptr;
    mmdx_pmx_s *m = new (std::nothrow) mmdx_pmx_s;
    if (!m) return pmx_fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->info.struct_size = sizeof(mmdx_pmx_info);
    try {
        parse(*m, static_cast<const uint8_t *>(data), size);
    } catch (const ParseError &e) {
        delete m;
        return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, e.what);
    } catch (const std::bad_alloc &) {
        delete m;
        return pmx_fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while parsing");
    }
    *out = m;
    return MMDX_OK;
}

mmdx_status mmdx_pmx_load_file(const char *path, mmdx_pmx_t *out) {
    if (!path || !out) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "path / out is NULL");
    *out = nullptr;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t got;
    try {
        while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    } catch (const std::bad_alloc &) {
        std::fclose(f);
        return pmx_fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while reading the file");
    }
    std::fclose(f);
    return mmdx_pmx_parse(buf.data(), buf.size(), out);
}

void mmdx_pmx_destroy(mmdx_pmx_t pmx) { delete pmx; }

mmdx_status mmdx_pmx_get_info(mmdx_pmx_t pmx, mmdx_pmx_info *info) {
    if (!pmx || !info || info->struct_size != sizeof(mmdx_pmx_info))
        return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or mmdx_pmx_info.struct_size mismatch");
    *info = pmx->info;
    return MMDX_OK;
}

mmdx_status mmdx_pmx_get_model_desc(mmdx_pmx_t pmx, mmdx_model_desc *d) {
    if (!pmx || !d) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(d, 0, sizeof(*d));
    d->struct_size = sizeof(*d);
    d->flags = MMDX_CREATE_NORMALIZE;   // the reference's reader ends with model.Normalize()
    d->n_vertices = pmx->info.n_vertices; d->n_bones = pmx->info.n_bones; d->n_morphs = pmx->info.n_morphs;
    d->positions = pmx->positions.data(); d->normals = pmx->normals.data(); d->uvs = pmx->uvs.data();
    d->skin_type = pmx->skin_type.data(); d->bone_ids = pmx->bone_ids.data();
    d->bone_weights = pmx->bone_weights.data(); d->sdef_params = pmx->sdef.data();
    d->bone_parent = pmx->bone_parent.data();
    d->morph_type = pmx->morph_type.data(); d->morph_offset = pmx->morph_offset.data();
    d->morph_index = pmx->morph_index.data(); d->morph_value = pmx->morph_value.data();
    return MMDX_OK;
}

mmdx_status mmdx_pmx_get_skeleton_desc(mmdx_pmx_t pmx, mmdx_skeleton_desc *d) {
    if (!pmx || !d) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(d, 0, sizeof(*d));
    d->struct_size = sizeof(*d);
    d->n_bones = pmx->info.n_bones;
    d->rest_position = pmx->bone_pos.data(); d->parent = pmx->bone_parent.data();
    d->transform_level = pmx->bone_level.data(); d->flags = pmx->bone_flags.data();
    d->append_parent = pmx->append_parent.data(); d->append_ratio = pmx->append_ratio.data();
    d->ik_target = pmx->ik_target.data(); d->ik_loop_count = pmx->ik_loop.data();
    d->ik_angle_limit = pmx->ik_angle.data(); d->ik_link_offset = pmx->ik_link_off.data();
    d->ik_link_bone = pmx->ik_link_bone.data(); d->ik_link_limited = pmx->ik_link_limited.data();
    d->ik_link_lo = pmx->ik_link_lo.data(); d->ik_link_hi = pmx->ik_link_hi.data();
    d->n_morphs = pmx->info.n_morphs;
    d->morph_type = pmx->morph_type.data(); d->morph_offset = pmx->morph_offset.data();
    d->morph_index = pmx->morph_index.data(); d->morph_value = pmx->morph_value.data();
    d->morph_rotation = pmx->morph_rotation.data();
    return MMDX_OK;
}

mmdx_status mmdx_pmx_get_arrays(mmdx_pmx_t pmx, mmdx_pmx_arrays *a) {
    if (!pmx || !a || a->struct_size != sizeof(mmdx_pmx_arrays))
        return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or mmdx_pmx_arrays.struct_size mismatch");
    a->triangles = pmx->triangles.data();
    a->material_index_count = pmx->material_index_count.data();
    a->bone_rest_position = pmx->bone_pos.data();
    a->bone_parent = pmx->bone_parent.data();
    a->bone_transform_level = pmx->bone_level.data();
    a->bone_flags = pmx->bone_flags.data();
    a->morph_panel = pmx->morph_panel.data();
    a->edge_scale = pmx->edge_scale.data();
    return MMDX_OK;
}

mmdx_status mmdx_pmx_get_name(mmdx_pmx_t pmx, int32_t kind, uint32_t index, char *buf, size_t buf_size) {
    if (!pmx || !buf || !buf_size) return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    const std::vector<std::string> *v = nullptr;
    switch (kind) {
    case MMDX_PMX_NAME_MODEL: break;
    case MMDX_PMX_NAME_BONE: v = &pmx->bone_names; break;
    case MMDX_PMX_NAME_MORPH: v = &pmx->morph_names; break;
    case MMDX_PMX_NAME_MATERIAL: v = &pmx->material_names; break;
    case MMDX_PMX_NAME_TEXTURE: v = &pmx->textures; break;
    default: return pmx_fail(MMDX_ERR_INVALID_ARGUMENT, "unknown name kind");
    }
    const std::string *s = v ? (index < v->size() ? &(*v)[index] : nullptr) : &pmx->name;
    if (!s) return pmx_fail(MMDX_ERR_BAD_INDEX, "name index out of range");
    std::snprintf(buf, buf_size, "%s", s->c_str());
    return MMDX_OK;
}

}  // extern "C"
