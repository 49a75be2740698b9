// Harness: closest hit of one primary ray through the packed BVH, shared verbatim by the
// host (rts_scene.cpp) and device (rts_primary.hip) G-buffer generators so that both produce the same bits.
// Stackless walk over the miss links; boxes are culled against the best t so far.  Input synthesis only -- this is
// not the path under test (the reference rasterises its G-buffer, Source/Shaders/Model.vert/.frag).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RTS_HD __host__ __device__ inline
#else
#define RTS_HD inline
#endif

namespace rts_harness {

struct V3 { float x, y, z; };
RTS_HD V3 sub(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
RTS_HD V3 add(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
RTS_HD V3 mul(V3 a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
RTS_HD V3 cross(V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
RTS_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTS_HD float asFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

struct Camera { V3 eye, fwd, right, up; float tanHalf, aspect; };

struct Hit { float t; uint32_t leaf; };   // leaf = node index of the hit triangle, 0xFFFFFFFF = none

RTS_HD V3 primaryDirection(const Camera& c, uint32_t x, uint32_t y, uint32_t W, uint32_t H) {
    float sx = (((float)x + 0.5f) / (float)W * 2.0f - 1.0f) * c.tanHalf * c.aspect;
    float sy = (1.0f - ((float)y + 0.5f) / (float)H * 2.0f) * c.tanHalf;
    return add(add(c.fwd, mul(c.right, sx)), mul(c.up, sy));
}

// One triangle against the best hit so far (a = {e0, tail index}, b = {e1, next}, t = {v0}): strictly nearer hits replace it,
// so of two triangles at the same distance the one met first in depth-first order stays.
RTS_HD void leafTest(const uint32_t* a, const uint32_t* b, const uint32_t* t, uint32_t node, V3 o, V3 d, Hit* best) {
    V3 e0{ asFloat(a[0]), asFloat(a[1]), asFloat(a[2]) }, e1{ asFloat(b[0]), asFloat(b[1]), asFloat(b[2]) };
    V3 v0{ asFloat(t[0]), asFloat(t[1]), asFloat(t[2]) };
    V3 s1 = cross(d, e1);
    float det = dot(s1, e0);
    if (det != 0.0f) {
        float invd = 1.0f / det;
        V3 dd = sub(o, v0);
        float b1 = dot(dd, s1) * invd;
        V3 s2 = cross(dd, e0);
        float b2 = dot(d, s2) * invd;
        float tt = dot(e1, s2) * invd;
        if (b1 >= 0.0f && b2 >= 0.0f && b1 + b2 <= 1.0f && tt > 0.0f && tt < best->t) { best->t = tt; best->leaf = node; }
    }
}

// Slab test of an inner node (a = {bboxMin, -}, b = {bboxMax, next}) against [0, best t].
RTS_HD bool boxTest(const uint32_t* a, const uint32_t* b, V3 o, V3 inv, float bestT) {
    float lo[3] = { asFloat(a[0]), asFloat(a[1]), asFloat(a[2]) }, hi[3] = { asFloat(b[0]), asFloat(b[1]), asFloat(b[2]) };
    float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
    float t0 = 0.0f, t1 = bestT;
    for (int k = 0; k < 3; ++k) {
        float f = (hi[k] - oo[k]) * ii[k], n = (lo[k] - oo[k]) * ii[k];
        float mx = f > n ? f : n, mn = f > n ? n : f;
        if (mx < t1) t1 = mx;      // NaN (0*inf) compares false: the slab is ignored
        if (mn > t0) t0 = mn;
    }
    return t1 >= t0;
}

RTS_HD Hit closestHit(const uint32_t* bvh, V3 o, V3 d) {
    const float inf = asFloat(0x7F800000u);
    const V3 inv{ 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    Hit best{ inf, 0xFFFFFFFFu };
    uint32_t node = 0;
    while (node != 0xFFFFFFFFu) {
        const uint32_t* a = bvh + (size_t)node * 8;
        const uint32_t* b = a + 4;
        if (a[3] != 0xFFFFFFFFu) {
            leafTest(a, b, bvh + (size_t)a[3] * 4, node, o, d, &best);
        } else if (boxTest(a, b, o, inv, best.t)) {
            ++node;
            continue;
        }
        node = b[3];
    }
    return best;
}

// G-buffer texel: camera-relative position (Model.frag:39) and the face normal turned towards the viewer
// (Model.frag:38 `gl_FrontFacing ? n : -n`; the harness has no vertex normals, so the geometric one is used).
RTS_HD void writeTexel(const uint32_t* bvh, V3 d, Hit h, float* position4, float* normal4) {
    if (h.leaf == 0xFFFFFFFFu) {
        position4[0] = position4[1] = position4[2] = position4[3] = 0.0f;       // clear value (background)
        if (normal4) normal4[0] = normal4[1] = normal4[2] = normal4[3] = 0.0f;
        return;
    }
    V3 rel = mul(d, h.t);
    position4[0] = rel.x; position4[1] = rel.y; position4[2] = rel.z; position4[3] = 1.0f;
    if (normal4) {
        const uint32_t* a = bvh + (size_t)h.leaf * 8;
        V3 e0{ asFloat(a[0]), asFloat(a[1]), asFloat(a[2]) }, e1{ asFloat(a[4]), asFloat(a[5]), asFloat(a[6]) };
        V3 n = cross(e0, e1);
        float len2 = dot(n, n);
        float s = len2 > 0.0f ? 1.0f / __builtin_sqrtf(len2) : 0.0f;
        if (dot(n, d) > 0.0f) s = -s;
        normal4[0] = n.x * s; normal4[1] = n.y * s; normal4[2] = n.z * s; normal4[3] = 0.0f;
    }
}

RTS_HD void shadePixel(const uint32_t* bvh, const Camera& c, uint32_t x, uint32_t y, uint32_t W, uint32_t H,
                       float* position4, float* normal4) {
    V3 d = primaryDirection(c, x, y, W, H);
    writeTexel(bvh, d, closestHit(bvh, c.eye, d), position4, normal4);
}

// Combine.frag:18-37 with baseColor = 1 (the default white material, RayTracedShadows.cpp:1013-1018), one pixel:
//   direct  = 1.25 * max(0, N.L) * shadowMask            shadowMask = mask / samples
//   ambient = 0.15 + 0.05 * (1 - max(0, N.(-cameraDirection)))
//   pixel discarded (left 0) where the normal is 0 (background)
// L = the light direction for a directional light; for the point-light extension L = normalize(light - P).
struct CombineParams { V3 cam, viewDir /* normalised */, light; uint32_t pointLight; float samples; };

RTS_HD uint8_t combinePixel(const CombineParams& c, const float* position4, const float* normal4, uint8_t mask) {
    V3 n{ normal4[0], normal4[1], normal4[2] };
    if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) return 0;
    V3 L = c.light;
    if (c.pointLight) {
        V3 p{ c.cam.x + position4[0], c.cam.y + position4[1], c.cam.z + position4[2] };
        L = sub(L, p);
        float ll = __builtin_sqrtf(dot(L, L));
        if (ll > 0) L = mul(L, 1.0f / ll);
    }
    float ndl = dot(n, L); if (ndl < 0) ndl = 0;
    float ndv = dot(n, mul(c.viewDir, -1.0f)); if (ndv < 0) ndv = 0;
    const float direct = 1.25f * ndl * ((float)mask / c.samples);          // frag:29
    const float ambient = 0.15f + 0.05f * (1.0f - ndv);                     // frag:30
    float v = direct + ambient;                                             // frag:32 (baseColor = 1)
    int q = (int)(v * 255.0f + 0.5f); if (q > 255) q = 255; if (q < 0) q = 0;
    return (uint8_t)q;
}

} // namespace rts_harness
