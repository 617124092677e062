// glue_driver.cpp -- compiles host/libmmd_glue.hpp (the reference-side binding of INTEGRATION.md section 1) against the REAL
// libmmd and runs it without a GPU (build container only; tests/test_libmmd_glue.py):
//   1. a .pmx read by libmmd's own PmxReader -> glue::CreateMmdxModel(MMDX_CREATE_HOST_ONLY) must give the same compiled model
//      (counts, post-Normalize skin tags, bone ids, weights) as the same file through this repo's loader
//      (mmdx_pmx_load_file -> mmdx_pmx_get_model_desc -> mmdx_model_create);
//   2. glue::MorphRateMirror must reproduce what MotionPlayer::SeekFrame puts into Poser::morph_rates_: a second Poser fed
//      the mirrored rates through SetMorphPose deforms bit-identically to the one the MotionPlayer drove;
//   3. glue::PaletteTap::Read returns the palette both posers ended with.
// Include order as in main.cpp:10-23 (C headers, then mmd.hxx): see oracle/ref_harness.cpp on `abs`.
#include <math.h>
#include <stdlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <mmd/mmd.hxx>

#include "../simple_mmd_renderer_amd/host/libmmd_glue.hpp"

static int fail(const char *what) {
    std::fprintf(stderr, "glue_driver: FAILED: %s (%s)\n", what, mmdx_last_error_string());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 4) return fail("usage: glue_driver model.pmx motion_model.pmx motion.vmd");
    const std::string pmx_path = argv[1], motion_pmx_path = argv[2], vmd_path = argv[3];
    mmd::Model model;
    {
        mmd::FileReader file(std::wstring(pmx_path.begin(), pmx_path.end()));
        mmd::PmxReader(file).ReadModel(model);                                   // main.cpp:656-661
    }
    // ---- 1. the model through the glue vs through the bundled loader -------------------------------------------------
    mmdx_model_t via_glue = nullptr, via_loader = nullptr;
    if (mmdx::glue::CreateMmdxModel(model, MMDX_CREATE_HOST_ONLY, &via_glue) != MMDX_OK) return fail("CreateMmdxModel");
    mmdx_pmx_t pmx = nullptr;
    if (mmdx_pmx_load_file(pmx_path.c_str(), &pmx) != MMDX_OK) return fail("mmdx_pmx_load_file");
    mmdx_model_desc desc;
    if (mmdx_pmx_get_model_desc(pmx, &desc) != MMDX_OK) return fail("mmdx_pmx_get_model_desc");
    desc.flags |= MMDX_CREATE_HOST_ONLY;
    if (mmdx_model_create(&desc, &via_loader) != MMDX_OK) return fail("mmdx_model_create (loader path)");
    mmdx_model_info a, b;
    a.struct_size = b.struct_size = sizeof(mmdx_model_info);
    if (mmdx_model_get_info(via_glue, &a) != MMDX_OK || mmdx_model_get_info(via_loader, &b) != MMDX_OK) return fail("get_info");
    if (a.n_vertices != b.n_vertices || a.n_bones != b.n_bones || a.n_morphs != b.n_morphs || a.n_slots != b.n_slots ||
        a.n_entries != b.n_entries || a.n_entries_padded != b.n_entries_padded || a.n_tiles != b.n_tiles ||
        a.n_bdef1 != b.n_bdef1 || a.n_bdef2 != b.n_bdef2 || a.n_bdef4 != b.n_bdef4 || a.max_tile_bones != b.max_tile_bones)
        return fail("model info differs between the glue path and the loader path");
    const size_t nv = a.n_vertices, nb = a.n_bones, nm = a.n_morphs;
    std::vector<int32_t> ta(nv), tb(nv), ia(nv * 4), ib(nv * 4);
    std::vector<float> wa(nv * 4), wb(nv * 4);
    if (mmdx_model_get_skin(via_glue, ta.data(), ia.data(), wa.data()) != MMDX_OK ||
        mmdx_model_get_skin(via_loader, tb.data(), ib.data(), wb.data()) != MMDX_OK) return fail("get_skin");
    if (ta != tb || ia != ib || std::memcmp(wa.data(), wb.data(), nv * 16) != 0)
        return fail("post-Normalize skin differs between the glue path and the loader path");
    std::vector<uint32_t> oa(nv), ob(nv);
    mmdx_model_get_vertex_order(via_glue, oa.data(), nullptr);
    mmdx_model_get_vertex_order(via_loader, ob.data(), nullptr);
    if (oa != ob) return fail("engine vertex order differs");
    mmdx_model_destroy(via_glue);
    mmdx_model_destroy(via_loader);
    mmdx_pmx_destroy(pmx);
    std::printf("model: %zu vertices, %zu bones, %zu morphs, %u slots, %u entries; bdef1/2/4 %u/%u/%u -- glue == loader\n", nv, nb, nm,
                a.n_slots, a.n_entries, a.n_bdef1, a.n_bdef2, a.n_bdef4);

    // ---- 2. + 3. the morph-rate mirror and the palette tap against MotionPlayer ----------------------------------------
    // A second model whose names MotionPlayer can actually match on this platform: on Linux libmmd's Shift-JIS -> wide-string
    // conversion of the VMD's names keeps iconv's byte-order mark, so no name of a real .pmx ever equals one (a reference
    // defect, tests/test_vmd.py); the test writes this model's names with that mark so that the association is not empty.
    {
        model.Clear();
        mmd::FileReader file(std::wstring(motion_pmx_path.begin(), motion_pmx_path.end()));
        mmd::PmxReader(file).ReadModel(model);
    }
    const size_t nv2 = model.GetVertexNum(), nb2 = model.GetBoneNum(), nm2 = model.GetMorphNum();
    mmd::Motion motion;
    {
        mmd::FileReader file(std::wstring(vmd_path.begin(), vmd_path.end()));
        mmd::VmdReader(file).ReadMotion(motion);
    }
    mmd::Poser by_player(model), by_mirror(model);
    mmd::MotionPlayer player(motion, by_player);
    mmdx::glue::MorphRateMirror mirror(motion, model);
    std::vector<float> rates(nm2), pal_a(nb2 * 16), pal_b(nb2 * 16);
    size_t nonzero = 0;
    const size_t frames[] = {0, 7, 33, 60, 119, 500};
    for (size_t f : frames) {
        by_player.ResetPosing();                                                  // main.cpp:1788-1810
        player.SeekFrame(f);
        by_player.PrePhysicsPosing();
        by_player.PostPhysicsPosing();
        by_player.Deform();
        by_mirror.ResetPosing();
        mirror.Seek(f, rates.data());
        for (size_t i = 0; i < nm2; ++i) {
            by_mirror.SetMorphPose(i, mmd::Motion::MorphPose(rates[i]));
            nonzero += rates[i] != 0.f;
        }
        for (size_t i = 0; i < nb2; ++i) {                                         // the bone half of SeekFrame, public API
            const std::wstring &name = model.GetBone(i).GetName();
            if (motion.IsBoneRegistered(name)) by_mirror.SetBonePose(i, motion.GetBonePose(name, f));
        }
        by_mirror.PrePhysicsPosing();
        by_mirror.PostPhysicsPosing();
        by_mirror.Deform();
        if (std::memcmp(by_player.pose_image.coordinates.data(), by_mirror.pose_image.coordinates.data(), nv2 * 12) != 0 ||
            std::memcmp(by_player.pose_image.normals.data(), by_mirror.pose_image.normals.data(), nv2 * 12) != 0)
            return fail("a Poser fed the mirrored rates does not deform like the one MotionPlayer drove");
        mmdx::glue::PaletteTap::Read(by_player, nb2, pal_a.data());
        mmdx::glue::PaletteTap::Read(by_mirror, nb2, pal_b.data());
        if (std::memcmp(pal_a.data(), pal_b.data(), nb2 * 64) != 0) return fail("palette tap");
    }
    if (!nonzero) return fail("the motion drove no morph of this model: the mirror check would be vacuous");
    std::printf("mirror: %zu non-zero rates over %zu frames reproduce MotionPlayer::SeekFrame; palette tap consistent\n", nonzero,
                sizeof(frames) / sizeof(frames[0]));
    std::printf("GLUE OK\n");
    return 0;
}
