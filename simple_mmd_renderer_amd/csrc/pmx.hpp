// pmx.hpp -- the parsed-model handle shared by the PMX 2.0 parser (pmx.cpp) and the PMD parser (pmd.cpp):
// both fill the same flat arrays, so every mmdx_pmx_get_* accessor serves either format.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mmdx.h"

struct mmdx_pmx_s {
    mmdx_pmx_info info{};
    std::string name, name_en;
    // vertex streams
    std::vector<float> positions, normals, uvs, bone_weights, sdef, edge_scale;
    std::vector<int32_t> skin_type, bone_ids;
    std::vector<uint32_t> triangles;
    std::vector<std::string> textures, material_names, bone_names, morph_names;
    std::vector<uint32_t> material_index_count;
    // bones
    std::vector<float> bone_pos;
    std::vector<int32_t> bone_parent, bone_level;
    std::vector<uint16_t> bone_flags;
    // append (inherit) and IK data of the bone block, for mmdx_skeleton_create
    std::vector<int32_t> append_parent, ik_target, ik_loop, ik_link_bone;
    std::vector<float> append_ratio, ik_angle, ik_link_lo, ik_link_hi;
    std::vector<uint32_t> ik_link_off;
    std::vector<uint8_t> ik_link_limited;
    // morphs
    std::vector<int32_t> morph_type;
    std::vector<uint8_t> morph_panel;
    std::vector<uint32_t> morph_offset, morph_index;
    std::vector<float> morph_value;
    std::vector<float> morph_rotation;     // [E][4]: bone-morph rotation, (0,0,0,1) for every other entry
};

enum : uint16_t { kPmxBoneHasIk = 0x0020, kPmxBoneAppendRotate = 0x0100 };

namespace mmdx {
// Shift-JIS bytes (up to the first NUL of a fixed-size field) -> UTF-8; invalid input is kept as is.
std::string sjis_field_to_utf8(const uint8_t *p, size_t field_size);
}  // namespace mmdx
