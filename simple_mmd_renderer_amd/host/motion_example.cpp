// motion_example.cpp -- the reference's whole per-frame sequence (main.cpp:1786-1825) from files, without
// libmmd: model + rig from a .pmx / .pmd, motion from a .vmd, every step on the GPU behind the C ABI.
//   g++ -std=c++17 -O2 motion_example.cpp -I../../include -L.. -lmmdx -Wl,-rpath,'$ORIGIN/..' -o motion_example
//   ./motion_example model.pmx motion.vmd [frames]
// Prints per-run checksums so tests can compare with the Python path over the same C ABI.
#include <cstdio>
#include <cstdlib>

#include "mmdx_poser.hpp"

static uint64_t checksum(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 3) { std::printf("usage: %s model.pmx|.pmd motion.vmd [frames]\n", argv[0]); return 64; }
    try {
        std::unique_ptr<mmdx::Poser> poser(mmdx::Poser::FromFile(argv[1]));
        mmdx::Motion motion(argv[2]);
        mmdx::MotionPlayer player(motion, *poser);
        const size_t frames = argc > 3 ? size_t(std::atoi(argv[3])) : motion.GetLength() + 1;
        std::vector<mmdx::Vertex> vertices;
        uint64_t h = 0;
        for (size_t frame = 0; frame < frames; ++frame) {
            poser->ResetPosing();                 // main.cpp:1788
            player.SeekFrame(frame);              // :1795
            poser->PrePhysicsPosing();            // :1801  (physics would React() here and overwrite its bones)
            poser->PostPhysicsPosing();           // :1810
            poser->Deform();                      // :1821
            poser->UpdateDeformedVertices(vertices);   // :1824
            h = h * 31 + checksum(vertices.data(), vertices.size() * sizeof(mmdx::Vertex)) +
                checksum(poser->pose_image.normals.data(), size_t(poser->vertex_count()) * 12);
        }
        std::printf("frames=%zu nv=%u nb=%u mapped_bones=%u checksum=%016llx\n", frames, poser->vertex_count(),
                    poser->bone_count(), player.mapped_bones(), (unsigned long long)h);
    } catch (const mmdx::Error &e) {
        std::printf("mmdx error %d: %s\n", int(e.status), e.what());
        return 1;
    }
    return 0;
}
