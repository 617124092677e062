// frame_loop_example.cpp -- the reference's per-frame sequence (main.cpp:1786-1825) with the two hot
// lines replaced by the GPU path, on a synthetic model.  Also the C++ host-side smoke test:
//   g++ -std=c++17 -O2 frame_loop_example.cpp -I../../include -L.. -lmmdx -Wl,-rpath,'$ORIGIN/..' -o frame_loop_example
// Prints a checksum of the last frame so tests can compare it with the Python/oracle path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "mmdx_poser.hpp"

static uint64_t checksum(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    const uint32_t nv = argc > 1 ? uint32_t(std::atoi(argv[1])) : 20000, nb = 150, nm = 30, k = 500;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 60;
    std::mt19937 rng(20001);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    mmdx::ModelData m;
    m.n_vertices = nv; m.n_bones = nb; m.n_morphs = nm;
    m.positions.resize(size_t(nv) * 3); m.normals.resize(size_t(nv) * 3); m.uvs.resize(size_t(nv) * 2);
    m.skin_type.resize(nv); m.bone_ids.assign(size_t(nv) * 4, 0); m.bone_weights.assign(size_t(nv) * 4, 0.f);
    m.bone_parent.assign(nb, -1);
    for (uint32_t b = 1; b < nb; ++b) m.bone_parent[b] = int32_t(rng() % b);
    for (uint32_t v = 0; v < nv; ++v) {
        for (int c = 0; c < 3; ++c) { m.positions[3 * v + c] = 20.f * u(rng) - 10.f; m.normals[3 * v + c] = 2.f * u(rng) - 1.f; }
        m.uvs[2 * v] = u(rng); m.uvs[2 * v + 1] = u(rng);
        const float r = u(rng);
        m.skin_type[v] = r < 0.2f ? MMDX_SKIN_BDEF1 : (r < 0.7f ? MMDX_SKIN_BDEF2 : (r < 0.95f ? MMDX_SKIN_BDEF4 : MMDX_SKIN_SDEF));
        const uint32_t lo = uint32_t(uint64_t(v) * (nb - 16) / nv);
        float sum = 0.f;
        for (int c = 0; c < 4; ++c) { m.bone_ids[4 * v + c] = int32_t(lo + rng() % 16); m.bone_weights[4 * v + c] = 0.01f + u(rng); sum += m.bone_weights[4 * v + c]; }
        if (m.skin_type[v] == MMDX_SKIN_BDEF4) for (int c = 0; c < 4; ++c) m.bone_weights[4 * v + c] /= sum;
    }
    m.morph_type.assign(nm, MMDX_MORPH_VERTEX);
    m.morph_offset.resize(nm + 1);
    for (uint32_t i = 0; i <= nm; ++i) m.morph_offset[i] = i * k;
    m.morph_index.resize(size_t(nm) * k); m.morph_value.resize(size_t(nm) * k * 3);
    for (size_t e = 0; e < m.morph_index.size(); ++e) {
        m.morph_index[e] = rng() % nv;
        for (int c = 0; c < 3; ++c) m.morph_value[3 * e + c] = u(rng) - 0.5f;
    }
    try {
        mmdx::Poser poser(m);
        std::vector<mmdx::Vertex> vertices;
        for (int f = 0; f < frames; ++f) {
            poser.ResetPosing();                                             // main.cpp:1788
            for (uint32_t i = 0; i < nm; ++i)                                 // MotionPlayer::SeekFrame
                poser.SetMorphPose(i, 0.5f + 0.5f * std::sin(6.2831853f * (f / 90.f + float(i) / nm)));
            for (uint32_t b = 0; b < nb; ++b) {                               // bone solve + physics (host)
                float *M = poser.SkinningMatrix(b);
                const float a = 0.5f * std::sin(6.2831853f * (f / 60.f + float(b) / nb)), c = std::cos(a), s = std::sin(a);
                const float R[16] = {c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0.1f * b / nb, 0.05f, 0, 1};
                std::memcpy(M, R, sizeof(R));
            }
            poser.Deform();                                                   // main.cpp:1821
            poser.UpdateDeformedVertices(vertices);                           // main.cpp:1824 (-> sg_update_buffer)
        }
        // the same last frame into page-locked storage (written by the kernel directly): identical bytes
        std::vector<mmdx::Vertex, mmdx::PinnedAllocator<mmdx::Vertex>> pinned;
        poser.UpdateDeformedVertices(pinned);
        if (std::memcmp(pinned.data(), vertices.data(), vertices.size() * sizeof(mmdx::Vertex)) != 0) {
            std::printf("MISMATCH between the page-locked and the pageable vertex buffer\n");
            return 2;
        }
        std::printf("frames=%d nv=%u pose_image=%016llx vertices=%016llx\n", frames, nv,
                    (unsigned long long)checksum(poser.pose_image.coordinates.data(), size_t(nv) * 12),
                    (unsigned long long)checksum(vertices.data(), vertices.size() * sizeof(mmdx::Vertex)));
        // consistency: the interleaved stream is pose_image * 0.1f + normals + uv
        for (uint32_t v = 0; v < nv; v += 997) {
            const float want = poser.pose_image.coordinates[v].x * 0.1f;
            if (std::memcmp(&want, &vertices[v].pos[0], 4) != 0 || vertices[v].uv[0] != m.uvs[2 * v]) {
                std::printf("MISMATCH at vertex %u\n", v);
                return 2;
            }
        }
    } catch (const mmdx::Error &e) {
        std::printf("mmdx error %d: %s\n", int(e.status), e.what());
        return 1;
    }
    return 0;
}
