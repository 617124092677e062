// plan.hpp -- host-side "model compile": flat PMX-like description -> the HBM layout the gfx950
// kernels stream.  Pure C++17 (no HIP), so it is unit-testable on machines without a GPU.
//
// What the reference keeps as AoS / tagged unions (mmd::Model::VertexInfo, 64-byte SkinningOperator,
// 136-byte MorphData; L/model/model.inl:21-104, :334-517, :719-726) becomes:
//
//   * vertex TILES of kTileVerts consecutive original vertices.  Inside a tile the vertices are
//     stably sorted by deform class (BDEF1 | BDEF2-like | BDEF4) so that 64-lane wavefronts are
//     class-uniform except at the two class boundaries; `perm` maps a sorted slot back to the
//     original local index (outputs stay in original vertex order -- the index buffer refers to it,
//     main.cpp:781-787).
//   * per-class skin streams in minimal encodings (BDEF1: one u16; BDEF2: 2 x u16 + f32;
//     BDEF4: 4 x u16 + 4 x f32); bone ids are TILE-LOCAL indices into the tile's sorted list of
//     distinct bones, so a workgroup stages only the bones its tile uses into LDS.
//   * the morph-major scatter lists turned into a vertex-major gather table (sliced ELL: 64 rows per
//     slice, rows sorted by length inside a class so the padding is small) whose per-vertex entry
//     order is the reference's accumulation order (morph index ascending, groups expanded
//     depth-first in place, file order inside a morph): no atomics, bit-reproducible.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mmdx.h"

namespace mmdx {

#ifndef MMDX_TILE
#define MMDX_TILE 512
#endif
constexpr uint32_t kTileVerts = MMDX_TILE;  // vertices per tile = per workgroup
constexpr uint32_t kSlicesPerTile = kTileVerts / 64;  // morph-table slices (wave-slots) per tile
constexpr uint32_t kMaxGroupDepth = 64;   // group-morph nesting limit (cycles are rejected)

struct TileHdr {                          // 48 bytes, read through the scalar cache
    uint32_t v0;                          // first original vertex of the tile
    uint32_t nv;                          // vertices in the tile (<= kTileVerts)
    uint32_t n1, n2;                      // BDEF1 count, BDEF2-like count (BDEF4 = nv - n1 - n2)
    uint32_t nbt;                         // distinct bones used by the tile
    uint32_t skin1_off, skin2_off, skin4_off;  // element offsets into the per-class streams
    uint32_t bone_off;                    // offset into bone_list
    uint32_t pad[3];
};
static_assert(sizeof(TileHdr) == 48, "TileHdr layout");

struct Plan {
    uint32_t nv = 0, nb = 0, nm = 0, flags = 0;
    uint32_t ns = 0;          // slots (vertex-morph applications in traversal order)
    uint32_t ne = 0;          // morph-table entries incl. slice padding
    uint32_t ne_real = 0;     // ... without padding (= vertex-morph entries after group expansion)
    uint32_t ntiles = 0;
    uint32_t n1 = 0, n2 = 0, n4 = 0;
    uint32_t max_tile_bones = 0;
    uint32_t max_tile_entries = 0;   // largest tile's slice of the morph table, in entries incl. padding
    bool f16 = false;
    bool finite_offsets = true;   // no inf/NaN among the vertex-morph offsets (after f16 rounding)

    // post-Normalize skin in ORIGINAL order (class 0/1/2, ids, weights)
    std::vector<int32_t> cls, ids;
    std::vector<float> wts;

    // sorted static streams (index = v0 + sorted slot)
    std::vector<float> spos;        // [NV][3]   (f32 mode)
    std::vector<uint16_t> spos16;   // [NV][4]   (f16 mode: x,y,z,0 as binary16)
    std::vector<float> snrm;        // [NV][3]
    std::vector<float> suv;         // [NV][2]
    std::vector<uint16_t> perm;     // [NV] sorted slot -> original local index
    std::vector<uint16_t> skin1;    // [n1]      tile-local bone
    std::vector<uint32_t> skin2_ids;  // [n2]    lb0 | lb1 << 16
    std::vector<float> skin2_w;     // [n2]
    std::vector<uint16_t> skin4_ids;  // [n4][4]
    std::vector<float> skin4_w;     // [n4][4]
    std::vector<uint32_t> bone_list;  // tile-local -> global bone id
    std::vector<TileHdr> tiles;

    // morph gather: sliced ELL (see plan.cpp)
    std::vector<uint32_t> ell;      // [ntiles*kSlicesPerTile][2] = {first entry, rows' padded length}
    std::vector<float> entries;     // [NE][4]  off.xyz, slot (as uint32 bits)     (f32 mode)
    std::vector<uint16_t> entries16;  // [NE][4] off.xyz as binary16, slot as u16  (f16 mode)
    std::vector<uint32_t> slot_top;   // [NS] top-level morph whose rate starts the chain
    std::vector<uint32_t> chain_off;  // [NS+1]
    std::vector<float> chain_rate;    // sub-rates applied in order (outermost group first)
};

// Returns MMDX_OK or an error code with a message in `err`.
mmdx_status build_plan(const mmdx_model_desc &desc, Plan &plan, std::string &err);

// One frame's slot weights, following Poser::UpdateMorphTransform (L/motion/poser_impl.inl:328-339):
// w = rate[top]; skip if w < 1e-7 (double compare); for each nested group: w = sub_rate * w, skip test
// again.  A skipped slot gets 0.0f (the device skips w < 1e-7f).  `out` has ns entries.
void flatten_slot_weights(const Plan &plan, const float *rates, float *out);

uint16_t f32_to_f16_rne(float f);

}  // namespace mmdx
