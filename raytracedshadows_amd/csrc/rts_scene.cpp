// Harness: synthesises the G-buffer targets the shadow kernel and the combine pass consume
// (Source/Shaders/Model.frag:35-39 writes normal and worldPosition - cameraPosition; targets created at
// Source/RayTracedShadows.cpp:378-387) and evaluates the combine pass (Source/Shaders/Combine.frag:18-37).
// Host versions, multi-threaded; the device version of the G-buffer pass is rts_primary.hip.
// This is input synthesis / presentation, not the path under test: both the GPU kernels and the CPU oracle are fed
// the buffers this file produces.
#include "../../include/rts_scene.h"
#include "rts_closest_hit.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

using namespace rts_harness;

namespace rts_harness {
// Pinhole camera as the reference sets it up (RayTracedShadows.cpp:238-242): vertical fov, lookAt, +Y up.
Camera makeCamera(const float eye[3], const float target[3], float fovy, uint32_t W, uint32_t H) {
    Camera c;
    c.eye = V3{ eye[0], eye[1], eye[2] };
    V3 f = sub(V3{ target[0], target[1], target[2] }, c.eye);
    float fl = std::sqrt(dot(f, f));
    c.fwd = fl > 0 ? mul(f, 1.0f / fl) : V3{ 0, 0, -1 };
    V3 r = cross(V3{ 0, 1, 0 }, c.fwd);
    float rl = std::sqrt(dot(r, r));
    c.right = rl > 0 ? mul(r, 1.0f / rl) : V3{ 1, 0, 0 };
    c.up = cross(c.fwd, c.right);
    c.tanHalf = std::tan(fovy * 0.5f);
    c.aspect = (float)W / (float)H;
    return c;
}
} // namespace rts_harness

extern "C" int rtsh_primary_gbuffer(const rts_vec4u* packed, size_t count, const float eye[3], const float target[3],
                                    float fovy, uint32_t W, uint32_t H, float* positions, float* normals,
                                    uint64_t* hit_count, int threads) {
    if (!packed || !eye || !target || !positions || W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    uint32_t P = 0;
    int s = rts_bvh_validate(packed, count, &P);
    if (s != RTS_OK) return s;
    const Camera cam = makeCamera(eye, target, fovy, W, H);
    const uint32_t* bvh = (const uint32_t*)packed;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    std::atomic<uint32_t> nextRow{ 0 };
    std::atomic<uint64_t> hits{ 0 };
    auto work = [&]() {
        uint64_t local = 0;
        for (;;) {
            uint32_t y = nextRow.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; ++x) {
                size_t i = ((size_t)y * W + x) * 4;
                shadePixel(bvh, cam, x, y, W, H, positions + i, normals ? normals + i : nullptr);
                local += positions[i + 3] != 0.0f;
            }
        }
        hits.fetch_add(local);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (hit_count) *hit_count = hits.load();
    return RTS_OK;
}

extern "C" int rtsh_primary_positions(const rts_vec4u* packed, size_t count, const float eye[3], const float target[3],
                                      float fovy, uint32_t W, uint32_t H, float* positions, uint64_t* hit_count,
                                      int threads) {
    return rtsh_primary_gbuffer(packed, count, eye, target, fovy, W, H, positions, nullptr, hit_count, threads);
}

namespace rts_harness {
// Shared by the host and device combine passes: validates and condenses the arguments.
int makeCombineParams(const rts_constants* k, const rts_light* light, bool havePositions, CombineParams* out) {
    if (!k || !out) return RTS_ERR_INVALID_ARG;
    if (light && light->type == RTS_LIGHT_POINT && !havePositions) return RTS_ERR_INVALID_ARG;
    CombineParams c;
    c.samples = (light && light->nsamples > 1) ? (float)light->nsamples : 1.0f;
    c.cam = V3{ k->cameraPosition[0], k->cameraPosition[1], k->cameraPosition[2] };
    V3 cd{ k->cameraDirection[0], k->cameraDirection[1], k->cameraDirection[2] };
    float cl = std::sqrt(dot(cd, cd));
    if (cl > 0) cd = mul(cd, 1.0f / cl);
    c.viewDir = cd;
    c.light = light ? V3{ light->xyz[0], light->xyz[1], light->xyz[2] }
                    : V3{ k->lightDirection[0], k->lightDirection[1], k->lightDirection[2] };
    c.pointLight = (light && light->type == RTS_LIGHT_POINT) ? 1u : 0u;
    *out = c;
    return RTS_OK;
}
} // namespace rts_harness

// Combine pass on the host (Combine.frag:18-37; per-pixel arithmetic in rts_closest_hit.h: combinePixel).
extern "C" int rtsh_combine(const rts_constants* k, const rts_light* light, const float* positions, const float* normals,
                            const uint8_t* mask, uint32_t W, uint32_t H, uint8_t* rgb) {
    if (!k || !normals || !mask || !rgb || W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    CombineParams c;
    int s = makeCombineParams(k, light, positions != nullptr, &c);
    if (s != RTS_OK) return s;
    const float zero[4] = { 0, 0, 0, 0 };
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const uint8_t q = combinePixel(c, positions ? positions + i * 4 : zero, normals + i * 4, mask[i]);
        rgb[i * 3] = rgb[i * 3 + 1] = rgb[i * 3 + 2] = q;
    }
    return RTS_OK;
}
