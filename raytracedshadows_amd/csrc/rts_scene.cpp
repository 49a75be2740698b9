// Harness: synthesises the G-buffer position target the shadow kernel consumes
// (Source/Shaders/Model.frag:35,39 writes worldPosition - cameraPosition into an RGBA32F target;
// Source/RayTracedShadows.cpp:385-387).  Here: closest hit of one pinhole ray per pixel centre
// through the same packed BVH, on the host, multi-threaded.
// This is input synthesis, not the path under test: both the GPU kernels and the CPU oracle are fed
// the buffer this file produces.
#include "../../include/rts_scene.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator+(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator*(V3 a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
inline V3 cross(V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }
inline float asF(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Closest hit along o + t*d, t in (0, inf).  Stackless walk over the miss links; boxes are culled
// against the best t so far.
float closestHit(const rts_vec4u* bvh, V3 o, V3 d) {
    const V3 inv{ 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    float best = INFINITY;
    uint32_t node = 0;
    while (node != 0xFFFFFFFFu) {
        const rts_vec4u& a = bvh[2 * (size_t)node];
        const rts_vec4u& b = bvh[2 * (size_t)node + 1];
        if (a.d != 0xFFFFFFFFu) {
            const rts_vec4u& t = bvh[a.d];
            V3 e0{ asF(a.a), asF(a.b), asF(a.c) }, e1{ asF(b.a), asF(b.b), asF(b.c) }, v0{ asF(t.a), asF(t.b), asF(t.c) };
            V3 s1 = cross(d, e1);
            float det = dot(s1, e0);
            if (det != 0.0f) {
                float invd = 1.0f / det;
                V3 dd = o - v0;
                float b1 = dot(dd, s1) * invd;
                V3 s2 = cross(dd, e0);
                float b2 = dot(d, s2) * invd;
                float tt = dot(e1, s2) * invd;
                if (b1 >= 0.0f && b2 >= 0.0f && b1 + b2 <= 1.0f && tt > 0.0f && tt < best) best = tt;
            }
        } else {
            float lo[3] = { asF(a.a), asF(a.b), asF(a.c) }, hi[3] = { asF(b.a), asF(b.b), asF(b.c) };
            float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
            float t0 = 0.0f, t1 = best;
            for (int k = 0; k < 3; ++k) {
                float f = (hi[k] - oo[k]) * ii[k], n = (lo[k] - oo[k]) * ii[k];
                float mx = f > n ? f : n, mn = f > n ? n : f;
                if (mx < t1) t1 = mx;     // NaN (0*inf) compares false: slab ignored
                if (mn > t0) t0 = mn;
            }
            if (t1 >= t0) { ++node; continue; }
        }
        node = b.d;
    }
    return best;
}

} // namespace

extern "C" int rtsh_primary_positions(const rts_vec4u* packed, size_t count, const float eye[3],
                                      const float target[3], float fovy, uint32_t W, uint32_t H,
                                      float* positions, uint64_t* hit_count, int threads) {
    if (!packed || !eye || !target || !positions || W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    uint32_t P = 0;
    int s = rts_bvh_validate(packed, count, &P);
    if (s != RTS_OK) return s;
    const V3 e{ eye[0], eye[1], eye[2] }, tg{ target[0], target[1], target[2] };
    const V3 fwd = normalize(tg - e);
    V3 right = cross(V3{ 0, 1, 0 }, fwd);
    if (dot(right, right) == 0.0f) right = V3{ 1, 0, 0 };
    right = normalize(right);
    const V3 up = cross(fwd, right);
    const float th = std::tan(fovy * 0.5f), aspect = (float)W / (float)H;

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
                float sx = (((float)x + 0.5f) / (float)W * 2.0f - 1.0f) * th * aspect;
                float sy = (1.0f - ((float)y + 0.5f) / (float)H * 2.0f) * th;
                V3 d = fwd + right * sx + up * sy;
                float t = closestHit(packed, e, d);
                float* o = positions + ((size_t)y * W + x) * 4;
                if (t < INFINITY) { V3 rel = d * t; o[0] = rel.x; o[1] = rel.y; o[2] = rel.z; o[3] = 1.0f; ++local; }
                else { o[0] = o[1] = o[2] = o[3] = 0.0f; }
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
