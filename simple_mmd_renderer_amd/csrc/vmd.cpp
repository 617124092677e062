// vmd.cpp -- VMD motion file -> keyframe tracks; morph tracks bound to a model are evaluated on the
// GPU for many instances / frames at once.  Host code, C++17; the device
// kernel lives in kernels.hip (morph_track_eval_kernel).
//
// Reference semantics followed (L/ = 3rd_party/libmmd/include/mmd/):
//   file layout: 50-byte header (magic[30], model name[20]), u32 count + 111-byte bone records,
//   u32 count + 23-byte morph records                 L/reader/interprete/vmd_types.inl:17-37,
//                                                      L/reader/vmd_reader_impl.inl:9-79
//   names: Shift-JIS, up to the first NUL of a 15-byte field  L/util/dwarf_impl.inl:22-27
//   a later record for the same (name, frame) replaces the earlier one (std::map operator[])
//                                                      L/motion/motion_impl.inl:221-240
//   model morph <-> track association by equal name    MotionPlayer ctor, L/motion/poser_impl.inl:522-537
//   morph weight at a frame: clamp to first / last key, exact key hit, else linear blend
//   l*(1-t) + r*t with t = float(frame-left)/float(right-left)
//                                                      Motion::GetMorphPose, L/motion/motion_impl.inl:382-424
//   (the per-key weight interpolator is a default Bezier, i.e. linear: L/util/math_impl.inl:1350-1354)
// Bone keyframes (translation, rotation, four Bezier control-point sets) are parsed here, exposed raw,
// and compiled / evaluated on the device by rig.cpp + rig_kernels.hip (mmdx_vmd_bind_bones).
//
// Reference defect worth knowing: on Linux libmmd converts names with iconv_open("UTF-16", "SHIFT-JIS"),
// whose output starts with a byte-order mark (L/util/dwarf_impl.inl:221-230), so VMD track names never
// compare equal to PMX names there and MotionPlayer maps nothing.  This loader implements the evident
// intent: names are equal when their decoded text is equal.
#include <iconv.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/mmdx.h"
#include "error.hpp"
#include "vmd.hpp"

namespace {

constexpr size_t kHeaderBytes = 50, kBoneBytes = 111, kMorphBytes = 23;

std::string field_name(const uint8_t *p) {  // up to the first NUL of a 15-byte field
    size_t n = 0;
    while (n < 15 && p[n]) ++n;
    return std::string(reinterpret_cast<const char *>(p), n);
}

std::string sjis_to_utf8(const std::string &s) {
    if (s.empty()) return s;
    iconv_t cd = iconv_open("UTF-8", "SHIFT-JIS");
    if (cd == iconv_t(-1)) return s;
    std::vector<char> in(s.begin(), s.end()), out(s.size() * 4 + 4);
    char *pi = in.data(), *po = out.data();
    size_t ni = in.size(), no = out.size();
    const size_t r = iconv(cd, &pi, &ni, &po, &no);
    iconv_close(cd);
    if (r == size_t(-1)) return s;  // not valid Shift-JIS: keep the bytes
    return std::string(out.data(), out.size() - no);
}

template <typename T>
T rd(const uint8_t *p) {
    T v;
    std::memcpy(&v, p, sizeof(T));
    return v;
}

}  // namespace

struct mmdx_vmd_s {
    mmdx_vmd_info info{};
    std::string model_name;
    std::vector<std::string> bone_names, morph_names;     // UTF-8, track order = first appearance
    std::vector<uint32_t> bone_off, morph_off;            // [tracks+1]
    std::vector<mmdx_vmd_bone_key> bone_keys;             // sorted by frame inside a track
    std::vector<uint32_t> morph_frames;
    std::vector<float> morph_weights;
};

struct mmdx_morph_motion_s {
    uint32_t nm = 0, n_mapped = 0;
    std::vector<uint32_t> key_off, frames;                // [nm+1], [K]
    std::vector<float> weights;                           // [K]
    mmdx::MorphMotionDevice dev;                          // uploaded lazily by the first eval
};

const mmdx::MorphMotionHost mmdx::morph_motion_host(const mmdx_morph_motion_s *m) {
    return {m->nm, m->key_off.data(), m->frames.data(), m->weights.data(), uint32_t(m->frames.size())};
}
mmdx::MorphMotionDevice &mmdx::morph_motion_device(mmdx_morph_motion_s *m) { return m->dev; }
mmdx::VmdBoneTracks mmdx::vmd_bone_tracks(const mmdx_vmd_s *v) {
    return {&v->bone_names, &v->bone_off, v->bone_keys.data()};
}

extern "C" {

mmdx_status mmdx_vmd_parse(const void *data, size_t size, mmdx_vmd_t *out) {
    using mmdx::fail;
    if (!data || !out) return fail(MMDX_ERR_INVALID_ARGUMENT, "data / out is NULL");
    *out = nullptr;
    const uint8_t *p = static_cast<const uint8_t *>(data);
    if (size < kHeaderBytes + 4 || std::memcmp(p, "Vocaloid Motion Data 0002", 25) != 0)
        return fail(MMDX_ERR_INVALID_ARGUMENT, "VMD: not a 'Vocaloid Motion Data 0002' file");
    mmdx_vmd_s *v = new (std::nothrow) mmdx_vmd_s;
    if (!v) return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    try {
        {
            size_t n = 0;
            while (n < 20 && p[30 + n]) ++n;
            v->model_name = sjis_to_utf8(std::string(reinterpret_cast<const char *>(p + 30), n));
        }
        size_t at = kHeaderBytes;
        const uint32_t nbone = rd<uint32_t>(p + at);
        at += 4;
        if (uint64_t(nbone) * kBoneBytes > size - at) {
            delete v;
            return fail(MMDX_ERR_INVALID_ARGUMENT, "VMD: file ends inside the bone keyframes");
        }
        std::map<std::string, uint32_t> track_of;
        std::vector<std::map<uint32_t, mmdx_vmd_bone_key>> btracks;
        for (uint32_t i = 0; i < nbone; ++i, at += kBoneBytes) {
            const std::string name = field_name(p + at);
            auto it = track_of.find(name);
            if (it == track_of.end()) {
                it = track_of.emplace(name, uint32_t(btracks.size())).first;
                btracks.emplace_back();
                v->bone_names.push_back(sjis_to_utf8(name));
            }
            mmdx_vmd_bone_key k;
            k.frame = rd<uint32_t>(p + at + 15);
            std::memcpy(k.translation, p + at + 19, 12);
            std::memcpy(k.rotation, p + at + 31, 16);
            std::memcpy(k.interpolation, p + at + 47, 64);
            btracks[it->second][k.frame] = k;   // a later record replaces an earlier one
        }
        if (size - at < 4) {
            delete v;
            return fail(MMDX_ERR_INVALID_ARGUMENT, "VMD: file ends before the morph keyframe count");
        }
        const uint32_t nmorph = rd<uint32_t>(p + at);
        at += 4;
        if (uint64_t(nmorph) * kMorphBytes > size - at) {
            delete v;
            return fail(MMDX_ERR_INVALID_ARGUMENT, "VMD: file ends inside the morph keyframes");
        }
        track_of.clear();
        std::vector<std::map<uint32_t, float>> mtracks;
        for (uint32_t i = 0; i < nmorph; ++i, at += kMorphBytes) {
            const std::string name = field_name(p + at);
            auto it = track_of.find(name);
            if (it == track_of.end()) {
                it = track_of.emplace(name, uint32_t(mtracks.size())).first;
                mtracks.emplace_back();
                v->morph_names.push_back(sjis_to_utf8(name));
            }
            mtracks[it->second][rd<uint32_t>(p + at + 15)] = rd<float>(p + at + 19);
        }
        uint32_t max_frame = 0;
        v->bone_off.push_back(0);
        for (const auto &t : btracks) {
            for (const auto &kv : t) { v->bone_keys.push_back(kv.second); max_frame = std::max(max_frame, kv.first); }
            v->bone_off.push_back(uint32_t(v->bone_keys.size()));
        }
        v->morph_off.push_back(0);
        for (const auto &t : mtracks) {
            for (const auto &kv : t) {
                v->morph_frames.push_back(kv.first);
                v->morph_weights.push_back(kv.second);
                max_frame = std::max(max_frame, kv.first);
            }
            v->morph_off.push_back(uint32_t(v->morph_frames.size()));
        }
        v->info.struct_size = sizeof(mmdx_vmd_info);
        v->info.n_bone_records = nbone; v->info.n_morph_records = nmorph;
        v->info.n_bone_tracks = uint32_t(btracks.size()); v->info.n_morph_tracks = uint32_t(mtracks.size());
        v->info.n_bone_keys = uint32_t(v->bone_keys.size()); v->info.n_morph_keys = uint32_t(v->morph_frames.size());
        v->info.max_frame = max_frame;
        v->info.bytes_consumed = at;
    } catch (const std::bad_alloc &) {
        delete v;
        return fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed while parsing");
    }
    *out = v;
    return MMDX_OK;
}

mmdx_status mmdx_vmd_load_file(const char *path, mmdx_vmd_t *out) {
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
    return mmdx_vmd_parse(buf.data(), buf.size(), out);
}

void mmdx_vmd_destroy(mmdx_vmd_t vmd) { delete vmd; }

mmdx_status mmdx_vmd_get_info(mmdx_vmd_t vmd, mmdx_vmd_info *info) {
    if (!vmd || !info || info->struct_size != sizeof(mmdx_vmd_info))
        return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument or mmdx_vmd_info.struct_size mismatch");
    *info = vmd->info;
    return MMDX_OK;
}

mmdx_status mmdx_vmd_track_name(mmdx_vmd_t vmd, int32_t is_morph, uint32_t track, char *buf, size_t buf_size) {
    if (!vmd || !buf || !buf_size) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    const auto &names = is_morph ? vmd->morph_names : vmd->bone_names;
    if (track >= names.size()) return mmdx::fail(MMDX_ERR_BAD_INDEX, "track index out of range");
    std::snprintf(buf, buf_size, "%s", names[track].c_str());
    return MMDX_OK;
}

mmdx_status mmdx_vmd_bone_track(mmdx_vmd_t vmd, uint32_t track, const mmdx_vmd_bone_key **keys, uint32_t *n_keys) {
    if (!vmd || !keys || !n_keys) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (track >= vmd->info.n_bone_tracks) return mmdx::fail(MMDX_ERR_BAD_INDEX, "track index out of range");
    *keys = vmd->bone_keys.data() + vmd->bone_off[track];
    *n_keys = vmd->bone_off[track + 1] - vmd->bone_off[track];
    return MMDX_OK;
}

mmdx_status mmdx_vmd_morph_track(mmdx_vmd_t vmd, uint32_t track, const uint32_t **frames, const float **weights,
                                 uint32_t *n_keys) {
    if (!vmd || !frames || !weights || !n_keys) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (track >= vmd->info.n_morph_tracks) return mmdx::fail(MMDX_ERR_BAD_INDEX, "track index out of range");
    *frames = vmd->morph_frames.data() + vmd->morph_off[track];
    *weights = vmd->morph_weights.data() + vmd->morph_off[track];
    *n_keys = vmd->morph_off[track + 1] - vmd->morph_off[track];
    return MMDX_OK;
}

mmdx_status mmdx_vmd_bind_morphs(mmdx_vmd_t vmd, uint32_t n_morphs, const char *const *morph_names,
                                 mmdx_morph_motion_t *out) {
    if (!vmd || !out || (n_morphs && !morph_names)) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    mmdx_morph_motion_s *m = new (std::nothrow) mmdx_morph_motion_s;
    if (!m) return mmdx::fail(MMDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    std::map<std::string, uint32_t> track_of;
    for (uint32_t t = 0; t < vmd->morph_names.size(); ++t) track_of.emplace(vmd->morph_names[t], t);
    m->nm = n_morphs;
    m->key_off.push_back(0);
    for (uint32_t i = 0; i < n_morphs; ++i) {
        auto it = morph_names[i] ? track_of.find(morph_names[i]) : track_of.end();
        if (it != track_of.end()) {
            const uint32_t b = vmd->morph_off[it->second], e = vmd->morph_off[it->second + 1];
            m->frames.insert(m->frames.end(), vmd->morph_frames.begin() + b, vmd->morph_frames.begin() + e);
            m->weights.insert(m->weights.end(), vmd->morph_weights.begin() + b, vmd->morph_weights.begin() + e);
            ++m->n_mapped;
        }
        m->key_off.push_back(uint32_t(m->frames.size()));
    }
    *out = m;
    return MMDX_OK;
}

mmdx_status mmdx_morph_motion_get_info(mmdx_morph_motion_t mm, uint32_t *n_morphs, uint32_t *n_mapped, uint32_t *n_keys) {
    if (!mm) return mmdx::fail(MMDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_morphs) *n_morphs = mm->nm;
    if (n_mapped) *n_mapped = mm->n_mapped;
    if (n_keys) *n_keys = uint32_t(mm->frames.size());
    return MMDX_OK;
}

void mmdx_morph_motion_destroy(mmdx_morph_motion_t mm) {
    if (!mm) return;
    mmdx::morph_motion_release_device(mm->dev);
    delete mm;
}

}  // extern "C"
