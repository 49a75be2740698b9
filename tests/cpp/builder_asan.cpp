// Host-only driver for the AddressSanitizer / UBSan build of the BVH producer (tests/test_builder.py):
// triangles whose extents overflow the SAH cost, degenerate cost ties, tiny inputs.  Prints one line per case
// "<name> <status>"; any wild access aborts the run under ASan.
#include "../../raytracedshadows_amd/csrc/bvh_builder.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static int run(const char* name, const std::vector<float>& v, unsigned P, unsigned sahLimit = 1000000u) {
    std::vector<unsigned> idx(3 * P);
    for (unsigned i = 0; i < 3 * P; ++i) idx[i] = i;
    rts::BVHBuilder b;
    b.sahPrimLimit = sahLimit;
    b.threads = 2;
    const bool ok = b.build(v.data(), 3, idx.data(), P);
    std::printf("%s %d %zu\n", name, ok ? 0 : b.lastError, b.m_packedNodes.size());
    return ok ? 0 : b.lastError;
}

int main() {
    const float scales[] = { 1e18f, 1e19f, 1e20f, 1e30f, 3e38f };
    for (float s : scales) {
        for (unsigned P : { 2u, 3u, 17u, 5000u, 9000u }) {     // 9000 >= 2 * share threshold: the threaded path
            std::vector<float> v(9 * P);
            unsigned seed = 12345u + P;
            for (float& f : v) { seed = seed * 1664525u + 1013904223u; f = ((seed >> 8) / 16777216.0f - 0.5f) * s; }
            char name[64];
            std::snprintf(name, sizeof name, "scale%g_P%u", (double)s, P);
            run(name, v, P);
            run(name, v, P, 4);                                  // median-split branch at the top
        }
    }
    {   // every cost ties (identical triangles): a chain, no overflow
        std::vector<float> v;
        for (int i = 0; i < 300; ++i) { const float t[9] = { 0, 0, 0, 1, 0, 0, 0, 1, 0 }; v.insert(v.end(), t, t + 9); }
        run("ties", v, 300);
    }
    {   // one huge + many tiny: only the top ranges overflow
        std::vector<float> v;
        for (int i = 0; i < 64; ++i) { const float t[9] = { (float)i, 0, 0, i + 1.f, 0, 0, (float)i, 1, 0 }; v.insert(v.end(), t, t + 9); }
        const float big[9] = { -1e20f, -1e20f, -1e20f, 1e20f, 1e20f, -1e20f, 0, 1e20f, 1e20f };
        v.insert(v.end(), big, big + 9);
        run("one_huge", v, 65);
    }
    return 0;
}
