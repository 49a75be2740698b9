// =============================================================================
// rts_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// This file is NOT part of the product.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load it; the product library
// (raytracedshadows_amd/csrc) never links, includes or calls anything here.
//
// What it is: a scalar CPU restatement of the reference's hot path
//   * BVH producer ....... /root/reference/Source/BVHBuilder.cpp:24-368
//   * any-hit traversal .. /root/reference/Source/Shaders/RayTracedShadows.comp:41-151
// written so that every floating-point operation is evaluated exactly as the
// reference source spells it (no FMA contraction, IEEE divide, GLSL
// compare-select min/max).
//
// Why C++ and not plain C: the reference's tree depends on the tie order of
// libstdc++'s (unstable) std::sort (BVHBuilder.cpp:92,149,162).  Re-using the
// same std::sort with the same comparator on the same sequence is the only way
// to restate that faithfully.
//
// PARITY PINNING STATUS: the reference ships no tests, golden vectors or
// fixtures for this path, its compute shader is GLSL (no glslc / Vulkan device
// here) and BVHBuilder.cpp needs headers of the absent, un-vendored `librush`
// submodule, so it is unbuildable here without writing stand-ins.  The only
// recorded output of the real reference builder is the 4-triangle dump in
// SURVEY.md Appendix A, which this oracle reproduces byte-for-byte
// (tests/test_oracle_golden.py).  Beyond that vector: "parity unpinned".
// Assumed librush math semantics: Box3::expandInit =
// {+FLT_MAX,-FLT_MAX}, Box3::expand = componentwise min/max,
// Box3::center = (min+max)*0.5f, Box3::dimensions = max-min.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -ffast-math).
// =============================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef uint32_t u32;
static const u32 kInvalid = 0xFFFFFFFFu;

static inline u32 f2u(float f) { u32 u; memcpy(&u, &f, 4); return u; }
static inline float u2f(u32 u) { float f; memcpy(&f, &u, 4); return f; }

// ----------------------------------------------------------------------------
// Builder restatement
// ----------------------------------------------------------------------------

// BVHBuilder.cpp:8-22 (TempNode : BVHNode).  primArea is dead in the reference
// (computed cpp:275, never read) and is omitted.
struct ONode {
    float lo[3]; u32 prim;      // BVHNode::bboxMin, prim      (BVHBuilder.h:13-14)
    float hi[3]; u32 next;      // BVHNode::bboxMax, next      (BVHBuilder.h:16-17)
    u32 order, parent, left, right;
    float ctr[3];
    float saL, saR;
};

struct OBox { float lo[3], hi[3]; };

static inline void boxInit(OBox& b) {              // librush Box3::expandInit (assumed)
    for (int k = 0; k < 3; ++k) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; }
}
static inline void boxExpand(OBox& b, const float* p) { // librush Box3::expand (assumed)
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = (p[k] < b.lo[k]) ? p[k] : b.lo[k];
        b.hi[k] = (b.hi[k] < p[k]) ? p[k] : b.hi[k];
    }
}

// BVHBuilder.cpp:24-28
static inline float surfaceArea(const float* lo, const float* hi) {
    float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ey + ey * ez + ez * ex) * 2.0f;
}

// BVHBuilder.cpp:53-76.  _mm_min_ps(a,b) = a<b ? a : b ; _mm_max_ps(a,b) = a>b ? a : b.
static OBox rangeBounds(const std::vector<ONode>& n, u32 begin, u32 end) {
    OBox r;
    if (begin == end) {
        for (int k = 0; k < 3; ++k) { r.lo[k] = 0.0f; r.hi[k] = 0.0f; }
        return r;
    }
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (u32 i = begin; i < end; ++i)
        for (int k = 0; k < 3; ++k) {
            mn[k] = (mn[k] < n[i].lo[k]) ? mn[k] : n[i].lo[k];
            mx[k] = (mx[k] > n[i].hi[k]) ? mx[k] : n[i].hi[k];
        }
    for (int k = 0; k < 3; ++k) { r.lo[k] = mn[k]; r.hi[k] = mx[k]; }
    return r;
}

// BVHBuilder.cpp:78-179
// tieByPrim (not in the reference): equal centroids are ordered by triangle id instead of being left wherever
// std::sort puts them -- the deterministic variant the GPU SAH builder (RTS_GPU_BUILD_SAH) is specified by.  On input
// without equal centroids both rules give the same tree.
struct CtrLess {
    u32 axis; bool tieByPrim;
    bool operator()(const ONode& a, const ONode& b) const {
        if (a.ctr[axis] < b.ctr[axis]) return true;
        return tieByPrim && a.ctr[axis] == b.ctr[axis] && a.prim < b.prim;
    }
};

static u32 splitRange(std::vector<ONode>& n, u32 begin, u32 end, const OBox& nodeBounds, u32 sahLimit, bool tieByPrim = false) {
    const u32 count = end - begin;
    u32 bestSplit = begin;                                    // cpp:81 (carried across axes)
    if (count <= sahLimit) {                                  // cpp:83 (1000000 in the reference)
        u32 bestAxis = 0, globalBestSplit = begin;
        float globalBestCost = FLT_MAX;
        for (u32 axis = 0; axis < 3; ++axis) {
            std::sort(n.begin() + begin, n.begin() + end, CtrLess{ axis, tieByPrim });     // cpp:92-96
            OBox bl, br; boxInit(bl); boxInit(br);
            for (u32 il = 0; il < count; ++il) {              // cpp:104-119
                u32 ir = count - il - 1;
                boxExpand(bl, n[begin + il].lo); boxExpand(bl, n[begin + il].hi);
                boxExpand(br, n[begin + ir].lo); boxExpand(br, n[begin + ir].hi);
                n[begin + il].saL = surfaceArea(bl.lo, bl.hi);
                n[begin + ir].saR = surfaceArea(br.lo, br.hi);
            }
            float bestCost = FLT_MAX;
            for (u32 mid = begin + 1; mid < end; ++mid) {     // cpp:121-139
                float costL = n[mid - 1].saL * (float)(mid - begin);
                float costR = n[mid].saR * (float)(end - mid);
                float cost = costL + costR;
                if (cost < bestCost) { bestSplit = mid; bestCost = cost; }
            }
            if (bestCost < globalBestCost) {                  // cpp:141-146
                globalBestSplit = bestSplit; globalBestCost = bestCost; bestAxis = axis;
            }
        }
        std::sort(n.begin() + begin, n.begin() + end, CtrLess{ bestAxis, tieByPrim });     // cpp:149-153
        return globalBestSplit;
    }
    // cpp:157-178: spatial median on the widest axis (first maximum wins, std::max_element)
    float ext[3] = { nodeBounds.hi[0] - nodeBounds.lo[0], nodeBounds.hi[1] - nodeBounds.lo[1],
                     nodeBounds.hi[2] - nodeBounds.lo[2] };
    int major = 0;
    for (int k = 1; k < 3; ++k) if (ext[major] < ext[k]) major = k;
    std::sort(n.begin() + begin, n.begin() + end, CtrLess{ (u32)major, tieByPrim });
    float splitPos = (nodeBounds.lo[major] + nodeBounds.hi[major]) * 0.5f;
    for (u32 mid = begin + 1; mid < end; ++mid)
        if (n[mid].ctr[major] >= splitPos) return mid;
    return end - 1;
}

// BVHBuilder.cpp:181-220
static u32 buildRange(std::vector<ONode>& n, u32 begin, u32 end, u32 sahLimit, bool tieByPrim = false) {
    if (end - begin == 1) return begin;
    OBox bounds = rangeBounds(n, begin, end);
    u32 mid = splitRange(n, begin, end, bounds, sahLimit, tieByPrim);
    u32 id = (u32)n.size();
    n.push_back(ONode());
    ONode node; memset(&node, 0, sizeof(node));
    node.order = kInvalid; node.parent = kInvalid; node.next = kInvalid;
    node.left = buildRange(n, begin, mid, sahLimit, tieByPrim);
    node.right = buildRange(n, mid, end, sahLimit, tieByPrim);
    float saLeft = surfaceArea(n[node.left].lo, n[node.left].hi);
    float saRight = surfaceArea(n[node.right].lo, n[node.right].hi);
    if (saRight > saLeft) std::swap(node.left, node.right);   // cpp:205-208
    for (int k = 0; k < 3; ++k) {
        node.lo[k] = bounds.lo[k]; node.hi[k] = bounds.hi[k];
        node.ctr[k] = (bounds.lo[k] + bounds.hi[k]) * 0.5f;   // Box3::center (assumed)
    }
    node.prim = kInvalid;
    n[node.left].parent = id; n[node.right].parent = id;
    n[id] = node;
    return id;
}

// BVHBuilder.cpp:222-238
static void dfsOrder(std::vector<ONode>& n, u32 id, u32 nextId, u32& order) {
    n[id].order = order++;
    n[id].next = nextId;
    u32 l = n[id].left, r = n[id].right;
    if (l != kInvalid) dfsOrder(n, l, r, order);
    if (r != kInvalid) dfsOrder(n, r, nextId, order);
}

// ----------------------------------------------------------------------------
// Traversal restatement (RayTracedShadows.comp)
// ----------------------------------------------------------------------------

// GLSL 4.50 spec 8.3: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gmax(float x, float y) { return (x < y) ? y : x; }

struct V3 { float x, y, z; };
static inline V3 crossv(V3 a, V3 b) {    // GLSL cross()
    V3 r; r.x = a.y * b.z - b.y * a.z; r.y = a.z * b.x - b.z * a.x; r.z = a.x * b.y - b.x * a.y; return r;
}
static inline float dotv(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 subv(V3 a, V3 b) { V3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }

// comp:41-59
static inline bool rayTri(V3 o, float tmax, V3 d, V3 v0, V3 e0, V3 e1) {
    V3 s1 = crossv(d, e1);
    float invd = 1.0f / dotv(s1, e0);
    V3 dd = subv(o, v0);
    float b1 = dotv(dd, s1) * invd;
    V3 s2 = crossv(dd, e0);
    float b2 = dotv(d, s2) * invd;
    float t = dotv(e1, s2) * invd;
    if (b1 < 0.0f || b1 > 1.0f || b2 < 0.0f || b1 + b2 > 1.0f || t < 0.0f || t > tmax) return false;
    return true;
}

// The same test, telling WHERE the disjunction of rejects first holds (for counting what a wave-wide early-out would skip):
// 0 hit, 1 b1 out of [0, 1], 2 b2 < 0 or b1 + b2 > 1, 3 t out of [0, tmax].  rayTri(...) == (rayTriStage(...) == 0).
static inline int rayTriStage(V3 o, float tmax, V3 d, V3 v0, V3 e0, V3 e1) {
    V3 s1 = crossv(d, e1);
    float invd = 1.0f / dotv(s1, e0);
    V3 dd = subv(o, v0);
    float b1 = dotv(dd, s1) * invd;
    V3 s2 = crossv(dd, e0);
    float b2 = dotv(d, s2) * invd;
    float t = dotv(e1, s2) * invd;
    if (b1 < 0.0f || b1 > 1.0f) return 1;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return 2;
    if (t < 0.0f || t > tmax) return 3;
    return 0;
}

// comp:61-73
static inline bool rayBox(V3 o, V3 invdir, V3 pmin, V3 pmax) {
    float fx = (pmax.x - o.x) * invdir.x, fy = (pmax.y - o.y) * invdir.y, fz = (pmax.z - o.z) * invdir.z;
    float nx = (pmin.x - o.x) * invdir.x, ny = (pmin.y - o.y) * invdir.y, nz = (pmin.z - o.z) * invdir.z;
    float tmaxx = gmax(fx, nx), tmaxy = gmax(fy, ny), tmaxz = gmax(fz, nz);
    float tminx = gmin(fx, nx), tminy = gmin(fy, ny), tminz = gmin(fz, nz);
    float t1 = gmin(tmaxx, gmin(tmaxy, tmaxz));
    float t0 = gmax(gmax(tminx, gmax(tminy, tminz)), 0.0f);
    return t1 >= t0;
}

// comp:75-111.  `bvh` = packed vec4 stream (4 u32 per vec4).  Counts nodes visited (V) and
// triangle tests (L) for the algorithmic-bytes figure.
static inline bool anyHit(const u32* bvh, V3 o, float tmax, V3 d, u32* V, u32* L) {
    V3 invdir = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    u32 node = 0, v = 0, l = 0;
    bool hit = false;
    while (node != kInvalid) {
        const u32* a = bvh + (size_t)node * 8;
        const u32* b = a + 4;
        ++v;
        u32 prim = a[3];
        if (prim != kInvalid) {
            const u32* t = bvh + (size_t)prim * 4;
            V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) };
            V3 e1 = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
            V3 v0 = { u2f(t[0]), u2f(t[1]), u2f(t[2]) };
            ++l;
            if (rayTri(o, tmax, d, v0, e0, e1)) { hit = true; break; }
        } else {
            V3 pmin = { u2f(a[0]), u2f(a[1]), u2f(a[2]) };
            V3 pmax = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
            if (rayBox(o, invdir, pmin, pmax)) { ++node; continue; }
        }
        node = b[3];
    }
    if (V) *V = v;
    if (L) *L = l;
    return hit;
}

// comp:113-120
static inline float epsilonFor(float f, u32 diff) {
    u32 u = f2u(f);
    u32 e = (u >> 23) & 0xFFu;
    e -= (diff < e) ? diff : e;
    u = (u & ~(0xFFu << 23)) | (e << 23);
    return u2f(u);
}
static inline float max3abs(V3 v) { return gmax(gmax(fabsf(v.x), fabsf(v.y)), fabsf(v.z)); }

// Light description shared with the product API (include/rts.h: rts_light).
//   type 0: directional, xyz = direction  -> exactly comp:128-151
//   type 1: point light at xyz (extension, BASELINE.json configs 2-5):
//           o0 = cam+rel; bias as comp:138-140; dn = (L-o0)/|L-o0|; o = o0 + dn*bias; d = L-o; tmax = 1
//   nsamples>1: sample j uses light xyz + offsets[j]; output = number of UNoccluded samples.
//   table != 0 (include/rts.h, rts_light.table): per-pixel jitter -- pixel p starts at entry (hash32(p) * table) >> 32
//           of a table of `table` offsets and takes nsamples consecutive entries (mod table).  Integers only.
struct OLight { u32 type; u32 nsamples; float xyz[3]; u32 table; float offsets[64][4]; };

static inline u32 pixelHash(u32 v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
static inline u32 tableEntry(const OLight& lt, u32 j, u32 pixel) {
    if (lt.table == 0) return j;
    const u32 start = (u32)(((uint64_t)pixelHash(pixel) * lt.table) >> 32);
    return (start + j) % lt.table;
}

static inline void genRay(const float* cam, V3 rel, const OLight& lt, u32 j, V3* o, float* tmax, V3* d, u32 pixel = 0) {
    V3 origin = { cam[0] + rel.x, cam[1] + rel.y, cam[2] + rel.z };                  // comp:136
    float bias = gmax(epsilonFor(max3abs(origin), 13), epsilonFor(max3abs(rel), 13)); // comp:138-140
    V3 L = { lt.xyz[0], lt.xyz[1], lt.xyz[2] };
    if (lt.nsamples > 1) { const u32 e = tableEntry(lt, j, pixel); L.x = L.x + lt.offsets[e][0]; L.y = L.y + lt.offsets[e][1]; L.z = L.z + lt.offsets[e][2]; }
    if (lt.type == 0) {
        origin.x = origin.x + L.x * bias; origin.y = origin.y + L.y * bias; origin.z = origin.z + L.z * bias; // comp:143
        *o = origin; *tmax = 1e9f; *d = L;                                            // comp:145-146
    } else {
        V3 d0 = subv(L, origin);
        float inv = 1.0f / sqrtf(dotv(d0, d0));
        origin.x = origin.x + (d0.x * inv) * bias; origin.y = origin.y + (d0.y * inv) * bias;
        origin.z = origin.z + (d0.z * inv) * bias;
        *o = origin; *tmax = 1.0f; *d = subv(L, origin);
    }
}

} // namespace

extern "C" {

// Number of vec4 in the packed buffer: 2N + P with N = 2P-1 (BVHBuilder.cpp:308-367).
uint64_t orc_packed_count(uint32_t P) { return P ? 5ull * P - 2 : 0; }

// Restates BVHBuilder::build (cpp:248-368).  out_packed: 4*(5P-2) u32.  out_nodes (optional):
// 8*N u32 = m_nodes (BVHNode, 32 B each).  Tail .w words are written as 0 (reference: stack
// garbage).  sah_limit = 1000000 reproduces cpp:83; tests may lower it to reach
// the median-split branch on small inputs.  Returns 0, or -1 on P==0.
static int buildPacked(const float* vertices, uint32_t stride, const uint32_t* indices, uint32_t P,
                       uint32_t sah_limit, uint32_t* out_packed, uint32_t* out_nodes, bool tieByPrim) {
    if (P == 0) return -1;
    std::vector<ONode> n;
    n.reserve((size_t)P * 2 - 1);
    for (u32 p = 0; p < P; ++p) {                              // cpp:261-284
        ONode node; memset(&node, 0, sizeof(node));
        OBox box; boxInit(box);
        for (int c = 0; c < 3; ++c) boxExpand(box, vertices + (size_t)stride * indices[p * 3 + c]);
        for (int k = 0; k < 3; ++k) {
            node.lo[k] = box.lo[k]; node.hi[k] = box.hi[k];
            node.ctr[k] = (box.lo[k] + box.hi[k]) * 0.5f;
        }
        node.prim = p; node.next = kInvalid; node.order = kInvalid; node.parent = kInvalid;
        node.left = kInvalid; node.right = kInvalid;
        n.push_back(node);
    }
    u32 root = buildRange(n, 0, P, sah_limit, tieByPrim);      // cpp:286
    u32 order = 0;
    dfsOrder(n, root, kInvalid, order);                        // cpp:288
    const u32 N = (u32)n.size();
    std::vector<ONode> dfs(N);                                 // m_nodes, cpp:290-306
    for (u32 i = 0; i < N; ++i) {
        ONode& dst = dfs[n[i].order];
        dst = n[i];
        dst.next = (n[i].next == kInvalid) ? kInvalid : n[n[i].next].order;
    }
    u32* out = out_packed;
    for (u32 i = 0; i < N; ++i) {                              // cpp:310-359
        const ONode& nd = dfs[i];
        if (out_nodes) {
            u32* o = out_nodes + (size_t)i * 8;
            o[0] = f2u(nd.lo[0]); o[1] = f2u(nd.lo[1]); o[2] = f2u(nd.lo[2]); o[3] = nd.prim;
            o[4] = f2u(nd.hi[0]); o[5] = f2u(nd.hi[1]); o[6] = f2u(nd.hi[2]); o[7] = nd.next;
        }
        if (nd.prim != kInvalid) {
            const float* v0 = vertices + (size_t)stride * indices[nd.prim * 3 + 0];
            const float* v1 = vertices + (size_t)stride * indices[nd.prim * 3 + 1];
            const float* v2 = vertices + (size_t)stride * indices[nd.prim * 3 + 2];
            out[0] = f2u(v1[0] - v0[0]); out[1] = f2u(v1[1] - v0[1]); out[2] = f2u(v1[2] - v0[2]);
            out[3] = nd.prim + N * 2;                          // cpp:331
            out[4] = f2u(v2[0] - v0[0]); out[5] = f2u(v2[1] - v0[1]); out[6] = f2u(v2[2] - v0[2]);
            out[7] = nd.next;
        } else {
            out[0] = f2u(nd.lo[0]); out[1] = f2u(nd.lo[1]); out[2] = f2u(nd.lo[2]); out[3] = kInvalid;
            out[4] = f2u(nd.hi[0]); out[5] = f2u(nd.hi[1]); out[6] = f2u(nd.hi[2]); out[7] = nd.next;
        }
        out += 8;
    }
    for (u32 p = 0; p < P; ++p) {                              // cpp:361-367
        const float* v0 = vertices + (size_t)stride * indices[p * 3 + 0];
        out[0] = f2u(v0[0]); out[1] = f2u(v0[1]); out[2] = f2u(v0[2]); out[3] = 0;
        out += 4;
    }
    return 0;
}

int orc_bvh_build(const float* vertices, uint32_t stride, const uint32_t* indices, uint32_t P,
                  uint32_t sah_limit, uint32_t* out_packed, uint32_t* out_nodes) {
    return buildPacked(vertices, stride, indices, P, sah_limit, out_packed, out_nodes, false);
}

// The same builder with equal centroids ordered by triangle id (CtrLess above): what RTS_GPU_BUILD_SAH must produce.
int orc_bvh_build_ties_by_prim(const float* vertices, uint32_t stride, const uint32_t* indices, uint32_t P,
                               uint32_t sah_limit, uint32_t* out_packed, uint32_t* out_nodes) {
    return buildPacked(vertices, stride, indices, P, sah_limit, out_packed, out_nodes, true);
}

// One generic ray {o.xyz, tmax} {d.xyz, 0} (the shader's `Ray`, comp:28-32).  Returns 1 on hit.
int orc_any_hit(const uint32_t* packed, const float* o4, const float* d4, uint32_t* V, uint32_t* L) {
    V3 o = { o4[0], o4[1], o4[2] }, d = { d4[0], d4[1], d4[2] };
    return anyHit(packed, o, o4[3], d, V, L) ? 1 : 0;
}

// n generic rays (8 floats each).  out[i] = 1 if NOT occluded (reference polarity, comp:148).
// sums[0] += nodes visited, sums[1] += triangle tests (may be NULL).
void orc_trace_rays(const uint32_t* packed, const float* rays, uint64_t n, uint8_t* out,
                    uint64_t* sums, int threads) {
    uint64_t sv = 0, sl = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : sv, sl)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float* r = rays + i * 8;
        V3 o = { r[0], r[1], r[2] }, d = { r[4], r[5], r[6] };
        u32 v, l;
        out[i] = anyHit(packed, o, r[3], d, &v, &l) ? 0 : 1;
        sv += v; sl += l;
    }
    if (sums) { sums[0] += sv; sums[1] += sl; }
}

// The dispatch of RayTracedShadows.cpp:570-595 + comp:128-151 over rows [row_begin,row_end) of a
// W x H frame.  constants: 16 floats = RayTracingConstants (RayTracedShadows.h:56-62); only
// cameraPosition.xyz is read here, the light comes from `light` (type 0 with xyz = constants'
// lightDirection.xyz is exactly the reference).  positions: W*H*4 floats (RGBA32F, camera-relative).
// mask: W*H bytes, rows outside the range untouched; value = number of unoccluded samples
// (0/1 for one sample).  per_ray_v / per_ray_l (optional, W*H u32; summed over samples).
void orc_shadow_mask(const uint32_t* packed, const float* constants, const void* light_v,
                     const float* positions, uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
                     uint8_t* mask, uint64_t* sums, uint32_t* per_ray_v, uint32_t* per_ray_l, int threads) {
    const OLight& lt = *(const OLight*)light_v;
    const u32 ns = lt.nsamples ? lt.nsamples : 1;
    uint64_t sv = 0, sl = 0;
    (void)H;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : sv, sl)
#endif
    for (int64_t y = row_begin; y < (int64_t)row_end; ++y) {
        for (u32 x = 0; x < W; ++x) {
            size_t pix = (size_t)y * W + x;
            V3 rel = { positions[pix * 4 + 0], positions[pix * 4 + 1], positions[pix * 4 + 2] }; // comp:135
            u32 lit = 0, pv = 0, pl = 0;
            for (u32 j = 0; j < ns; ++j) {
                V3 o, d; float tmax;
                genRay(constants, rel, lt, j, &o, &tmax, &d, (u32)pix);
                u32 v, l;
                lit += anyHit(packed, o, tmax, d, &v, &l) ? 0 : 1;                         // comp:148
                pv += v; pl += l;
            }
            mask[pix] = (uint8_t)lit;
            sv += pv; sl += pl;
            if (per_ray_v) per_ray_v[pix] = pv;
            if (per_ray_l) per_ray_l[pix] = pl;
        }
    }
    if (sums) { sums[0] += sv; sums[1] += sl; }
}

// Writes the rays the mask dispatch would generate (8 floats per ray, sample-major within a pixel).
void orc_gen_rays(const float* constants, const void* light_v, const float* positions, uint64_t npix, float* rays) {
    const OLight& lt = *(const OLight*)light_v;
    const u32 ns = lt.nsamples ? lt.nsamples : 1;
    for (uint64_t p = 0; p < npix; ++p) {
        V3 rel = { positions[p * 4 + 0], positions[p * 4 + 1], positions[p * 4 + 2] };
        for (u32 j = 0; j < ns; ++j) {
            V3 o, d; float tmax;
            genRay(constants, rel, lt, j, &o, &tmax, &d, (u32)p);
            float* r = rays + (p * ns + j) * 8;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = tmax; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = 0.0f;
        }
    }
}

// Independent check of the oracle itself: brute-force any-hit over ALL leaves of the packed
// buffer with the same triangle test and no box culling.  out[i] = 1 if not occluded.
void orc_brute_force_rays(const uint32_t* packed, uint32_t P, const float* rays, uint64_t n, uint8_t* out) {
    const u32 N = 2 * P - 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float* r = rays + i * 8;
        V3 o = { r[0], r[1], r[2] }, d = { r[4], r[5], r[6] };
        bool hit = false;
        for (u32 k = 0; k < N && !hit; ++k) {
            const u32* a = packed + (size_t)k * 8;
            if (a[3] == kInvalid) continue;
            const u32* t = packed + (size_t)a[3] * 4;
            V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(a[4]), u2f(a[5]), u2f(a[6]) };
            V3 v0 = { u2f(t[0]), u2f(t[1]), u2f(t[2]) };
            hit = rayTri(o, r[3], d, v0, e0, e1);
        }
        out[i] = hit ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------------
// G-buffer oracle: what the harness's closest-hit pass must produce for the position target the
// shadow kernel reads, i.e. Source/Shaders/Model.frag:35-39 `outCameraRelativePosition = v_worldPos - g_cameraPosition`
// with the background left at the clear value 0.  For a ray-traced G-buffer the camera-relative
// world position of the visible surface is d * t.  Written independently of the product's stackless walk
// (rts_closest_hit.h): an explicit-stack descent that finds the children through the layout (left = i + 1,
// right = next(left)).  Contract shared with the harness, stated here once:
//   primary ray   o = eye, d = fwd + right*sx + up*sy (un-normalised), sx/sy from the pixel centre
//   triangle      Moeller-Trumbore as comp:41-59, but accepted only if det != 0, b1 >= 0, b2 >= 0, b1+b2 <= 1,
//                 0 < t < best (strict: of equal distances the triangle met first in DFS order wins)
//   box           slab interval per axis clipped to [0, best]; NaN slabs (0*inf) are ignored; entered iff t1 >= t0
//   camera        lookAt(eye -> target), +Y up, vertical fov (RayTracedShadows.cpp:238-242)
// `cull` = 0 switches the box test off (brute force over every triangle in DFS order): the independent check of
// the traversal itself on small scenes.
// ---------------------------------------------------------------------------------------------
struct OCamera { V3 eye, fwd, right, up; float tanHalf, aspect; };

static OCamera lookAt(const float* eye, const float* target, float fovy, u32 W, u32 H) {
    OCamera c;
    c.eye = V3{ eye[0], eye[1], eye[2] };
    V3 f = subv(V3{ target[0], target[1], target[2] }, c.eye);
    float fl = sqrtf(dotv(f, f));
    if (fl > 0) { float s = 1.0f / fl; c.fwd = V3{ f.x * s, f.y * s, f.z * s }; } else c.fwd = V3{ 0, 0, -1 };
    V3 up0 = { 0, 1, 0 };
    V3 r = { up0.y * c.fwd.z - up0.z * c.fwd.y, up0.z * c.fwd.x - up0.x * c.fwd.z, up0.x * c.fwd.y - up0.y * c.fwd.x };
    float rl = sqrtf(dotv(r, r));
    if (rl > 0) { float s = 1.0f / rl; c.right = V3{ r.x * s, r.y * s, r.z * s }; } else c.right = V3{ 1, 0, 0 };
    c.up = V3{ c.fwd.y * c.right.z - c.fwd.z * c.right.y, c.fwd.z * c.right.x - c.fwd.x * c.right.z,
               c.fwd.x * c.right.y - c.fwd.y * c.right.x };
    c.tanHalf = tanf(fovy * 0.5f);
    c.aspect = (float)W / (float)H;
    return c;
}

static bool nearestHit(const u32* bvh, V3 o, V3 d, int cull, float* tOut, u32* leafOut) {
    const float inf = u2f(0x7F800000u);
    const float oo[3] = { o.x, o.y, o.z };
    const float inv[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    float best = inf; u32 bestLeaf = kInvalid;
    std::vector<u32> todo;
    todo.push_back(0);
    while (!todo.empty()) {
        const u32 i = todo.back(); todo.pop_back();
        const u32* a = bvh + (size_t)i * 8; const u32* b = a + 4;
        if (a[3] != kInvalid) {
            const u32* tv = bvh + (size_t)a[3] * 4;
            V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
            V3 v0 = { u2f(tv[0]), u2f(tv[1]), u2f(tv[2]) };
            V3 s1 = { d.y * e1.z - d.z * e1.y, d.z * e1.x - d.x * e1.z, d.x * e1.y - d.y * e1.x };
            float det = s1.x * e0.x + s1.y * e0.y + s1.z * e0.z;
            if (det != 0.0f) {
                float invd = 1.0f / det;
                V3 dd = subv(o, v0);
                float b1 = (dd.x * s1.x + dd.y * s1.y + dd.z * s1.z) * invd;
                V3 s2 = { dd.y * e0.z - dd.z * e0.y, dd.z * e0.x - dd.x * e0.z, dd.x * e0.y - dd.y * e0.x };
                float b2 = (d.x * s2.x + d.y * s2.y + d.z * s2.z) * invd;
                float t = (e1.x * s2.x + e1.y * s2.y + e1.z * s2.z) * invd;
                if (b1 >= 0.0f && b2 >= 0.0f && b1 + b2 <= 1.0f && t > 0.0f && t < best) { best = t; bestLeaf = i; }
            }
            continue;
        }
        bool enter = true;
        if (cull) {
            float t0 = 0.0f, t1 = best;
            for (int k = 0; k < 3; ++k) {
                float f = (u2f(b[k]) - oo[k]) * inv[k], n = (u2f(a[k]) - oo[k]) * inv[k];
                float far_ = f > n ? f : n, near_ = f > n ? n : f;
                if (far_ < t1) t1 = far_;
                if (near_ > t0) t0 = near_;
            }
            enter = t1 >= t0;
        }
        if (enter) {
            const u32 left = i + 1, right = bvh[(size_t)left * 8 + 7];      // the left child's miss link IS the right child
            todo.push_back(right);
            todo.push_back(left);                                           // popped first: DFS order
        }
    }
    *tOut = best; *leafOut = bestLeaf;
    return bestLeaf != kInvalid;
}

void orc_primary_gbuffer(const uint32_t* packed, const float* eye, const float* target, float fovy, uint32_t W,
                         uint32_t H, int cull, float* positions, float* normals, uint64_t* hits, int threads) {
    const OCamera c = lookAt(eye, target, fovy, W, H);
    uint64_t count = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : count)
#endif
    for (int64_t y = 0; y < (int64_t)H; ++y) {
        for (u32 x = 0; x < W; ++x) {
            float sx = (((float)x + 0.5f) / (float)W * 2.0f - 1.0f) * c.tanHalf * c.aspect;
            float sy = (1.0f - ((float)y + 0.5f) / (float)H * 2.0f) * c.tanHalf;
            V3 d = { c.fwd.x + c.right.x * sx + c.up.x * sy, c.fwd.y + c.right.y * sx + c.up.y * sy,
                     c.fwd.z + c.right.z * sx + c.up.z * sy };
            float* p = positions + ((size_t)y * W + x) * 4;
            float* nn = normals ? normals + ((size_t)y * W + x) * 4 : nullptr;
            float t; u32 leaf;
            if (!nearestHit(packed, c.eye, d, cull, &t, &leaf)) {
                p[0] = p[1] = p[2] = p[3] = 0.0f;                           // clear value: background
                if (nn) nn[0] = nn[1] = nn[2] = nn[3] = 0.0f;
                continue;
            }
            ++count;
            p[0] = d.x * t; p[1] = d.y * t; p[2] = d.z * t; p[3] = 1.0f;    // Model.frag:35  worldPos - cameraPos
            if (nn) {                                                       // Model.frag:33,38: unit normal facing the viewer
                const u32* a = packed + (size_t)leaf * 8;
                V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(a[4]), u2f(a[5]), u2f(a[6]) };
                V3 n = { e0.y * e1.z - e0.z * e1.y, e0.z * e1.x - e0.x * e1.z, e0.x * e1.y - e0.y * e1.x };
                float len2 = n.x * n.x + n.y * n.y + n.z * n.z;
                float s = len2 > 0.0f ? 1.0f / sqrtf(len2) : 0.0f;
                if (n.x * d.x + n.y * d.y + n.z * d.z > 0.0f) s = -s;
                nn[0] = n.x * s; nn[1] = n.y * s; nn[2] = n.z * s; nn[3] = 0.0f;
            }
        }
    }
    if (hits) *hits = count;
}

// Single-call helpers so tests can pin the scalar pieces (KATs).
int orc_ray_box(const float* o3, const float* invdir3, const float* pmin3, const float* pmax3) {
    V3 o = { o3[0], o3[1], o3[2] }, i = { invdir3[0], invdir3[1], invdir3[2] };
    V3 a = { pmin3[0], pmin3[1], pmin3[2] }, b = { pmax3[0], pmax3[1], pmax3[2] };
    return rayBox(o, i, a, b) ? 1 : 0;
}
int orc_ray_tri(const float* o4, const float* d3, const float* v03, const float* e03, const float* e13) {
    V3 o = { o4[0], o4[1], o4[2] }, d = { d3[0], d3[1], d3[2] };
    V3 v0 = { v03[0], v03[1], v03[2] }, e0 = { e03[0], e03[1], e03[2] }, e1 = { e13[0], e13[1], e13[2] };
    return rayTri(o, o4[3], d, v0, e0, e1) ? 1 : 0;
}
float orc_epsilon_for(float f, uint32_t diff) { return epsilonFor(f, diff); }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// Combine pass, restated from /root/reference/Source/Shaders/Combine.frag:18-37 (checker for SURVEY.md 8 f4; written from
// the shader, independently of the product's combinePixel):
//     worldNormal = normal target, baseColor = 1 (the default white material, RayTracedShadows.cpp:1013-1018)
//     directLight  = 1.25 * max(0.0, dot(worldNormal, lightDirection.xyz)) * shadowMask                  (frag:29)
//     ambientLight = 0.15 + 0.05 * (1.0 - max(0.0f, dot(worldNormal, -cameraDirection.xyz)))             (frag:30)
//     result.xyz   = baseColor * vec3(directLight + ambientLight)                                          (frag:32)
//     discard where worldNormal == 0 (the target keeps its clear value, 0)                                (frag:35-36)
// Every product is evaluated left to right as the shader spells it, in fp32 without contraction.  What the shader reads
// from textures arrives here as arrays: shadowMask = mask byte / samples (R8_UNORM of a 0/1 mask in the reference; the
// 16-sample extension stores the count of unoccluded samples).  Two stated extensions of the harness, shared with the
// product's contract (include/rts_scene.h): cameraDirection is normalised first (the reference passes a unit vector,
// the harness passes target - eye); for a point light lightDirection = normalize(light - worldPosition).
// Output: UNORM8 of result.x, round to nearest (three equal channels).
// ---------------------------------------------------------------------------------------------
extern "C" void orc_combine(const float* constants, const void* light_v, const float* positions, const float* normals,
                            const uint8_t* mask, uint32_t W, uint32_t H, uint8_t* rgb) {
    const OLight* lt = (const OLight*)light_v;
    const float* camPos = constants;                 // cameraPosition
    const float* camDir = constants + 4;             // cameraDirection
    const float* sunDir = constants + 8;             // lightDirection
    float view[3];
    {
        const float len = sqrtf(camDir[0] * camDir[0] + camDir[1] * camDir[1] + camDir[2] * camDir[2]);
        const float inv = len > 0.0f ? 1.0f / len : 1.0f;
        for (int a = 0; a < 3; ++a) view[a] = len > 0.0f ? camDir[a] * inv : camDir[a];
    }
    const float samples = (lt && lt->nsamples > 1) ? (float)lt->nsamples : 1.0f;
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const float* N = normals + i * 4;
        uint8_t out = 0;
        if (!(N[0] == 0.0f && N[1] == 0.0f && N[2] == 0.0f)) {
            float L[3];
            if (lt && lt->type == 1) {
                for (int a = 0; a < 3; ++a) L[a] = lt->xyz[a] - (camPos[a] + positions[i * 4 + a]);
                const float len = sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
                if (len > 0.0f) { const float inv = 1.0f / len; for (int a = 0; a < 3; ++a) L[a] = L[a] * inv; }
            } else {
                for (int a = 0; a < 3; ++a) L[a] = lt ? lt->xyz[a] : sunDir[a];
            }
            const float shadowMask = (float)mask[i] / samples;
            const float nDotL = N[0] * L[0] + N[1] * L[1] + N[2] * L[2];
            const float nDotV = N[0] * (-view[0]) + N[1] * (-view[1]) + N[2] * (-view[2]);
            const float directLight = 1.25f * (nDotL > 0.0f ? nDotL : 0.0f) * shadowMask;
            const float ambientLight = 0.15f + 0.05f * (1.0f - (nDotV > 0.0f ? nDotV : 0.0f));
            const float result = directLight + ambientLight;
            const float scaled = result * 255.0f + 0.5f;
            out = scaled >= 255.0f ? 255 : (scaled <= 0.0f ? 0 : (uint8_t)(int)scaled);
        }
        rgb[i * 3] = rgb[i * 3 + 1] = rgb[i * 3 + 2] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// Analysis helper (not a checker): for every 8x8 pixel tile of a frame, how many DISTINCT nodes
// do its 64 rays visit (what a wave-uniform sweep in increasing DFS index would iterate over),
// against the longest single ray (what a lane-per-ray loop iterates over) and the sum.
// out[0]=tiles, out[1]=sum of union sizes, out[2]=sum of per-tile max V, out[3]=sum of V,
// out[4]=sum over union nodes that are leaves, out[5]=sum over union nodes that are the right sibling of the union node
// visited just before them (a left child nobody entered, or a leaf: the sibling follows at once), out[6]=sum over union
// nodes reached by a side-step (a miss link), out[7]=those of out[6] whose left sibling's subtree was walked in between
// ---------------------------------------------------------------------------------------------
extern "C" void orc_tile_union_stats(const uint32_t* packed, const float* constants, const void* light_v,
                                     const float* positions, uint32_t W, uint32_t H, uint64_t* out) {
    extern void orc_tile_union_stats_wh(const uint32_t*, const float*, const void*, const float*, uint32_t, uint32_t,
                                        uint32_t, uint32_t, uint64_t*);
    orc_tile_union_stats_wh(packed, constants, light_v, positions, W, H, 8, 8, out);
}
extern "C" void orc_tile_union_stats_wh(const uint32_t* packed, const float* constants, const void* light_v,
                                        const float* positions, uint32_t W, uint32_t H, uint32_t TW, uint32_t TH,
                                        uint64_t* out) {
    const OLight& lt = *(const OLight*)light_v;
    uint64_t tiles = 0, sumU = 0, sumM = 0, sumV = 0, sumLeafU = 0, sumImm = 0, sumSide = 0;
    const uint32_t tx = W / TW, ty = H / TH;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : tiles, sumU, sumM, sumV, sumLeafU, sumImm, sumSide)
#endif
    for (int64_t t = 0; t < (int64_t)tx * ty; ++t) {
        uint32_t bx = (uint32_t)(t % tx), by = (uint32_t)(t / tx);
        std::vector<u32> visited;
        u32 maxV = 0;
        for (u32 l = 0; l < TW * TH; ++l) {
            u32 x = bx * TW + (l % TW), y = by * TH + (l / TW);
            size_t pix = (size_t)y * W + x;
            V3 rel = { positions[pix * 4 + 0], positions[pix * 4 + 1], positions[pix * 4 + 2] };
            V3 o, d; float tmax;
            genRay(constants, rel, lt, 0, &o, &tmax, &d);
            V3 invdir = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
            u32 node = 0, v = 0;
            while (node != kInvalid) {
                const u32* a = packed + (size_t)node * 8; const u32* b = a + 4;
                ++v; visited.push_back(node);
                if (a[3] != kInvalid) {
                    const u32* tt = packed + (size_t)a[3] * 4;
                    V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
                    V3 v0 = { u2f(tt[0]), u2f(tt[1]), u2f(tt[2]) };
                    if (rayTri(o, tmax, d, v0, e0, e1)) break;
                } else {
                    V3 pmin = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, pmax = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
                    if (rayBox(o, invdir, pmin, pmax)) { ++node; continue; }
                }
                node = b[3];
            }
            sumV += v; if (v > maxV) maxV = v;
        }
        std::sort(visited.begin(), visited.end());
        visited.erase(std::unique(visited.begin(), visited.end()), visited.end());
        sumU += visited.size(); sumM += maxV; ++tiles;
        for (u32 n : visited) if (packed[(size_t)n * 8 + 3] != kInvalid) ++sumLeafU;
        for (size_t k = 1; k < visited.size(); ++k) {
            const u32 prev = visited[k - 1], cur = visited[k];
            if (cur == prev + 1 && packed[(size_t)prev * 8 + 3] == kInvalid) continue;          // down-step into a left child
            ++sumSide;                                                                          // reached through a miss link
            const bool prevIsLeftChild = prev > 0 && packed[(size_t)(prev - 1) * 8 + 3] == kInvalid;
            if (prevIsLeftChild && packed[(size_t)prev * 8 + 7] == cur) ++sumImm;                // ... of the node visited just before
        }
    }
    out[0] = tiles; out[1] = sumU; out[2] = sumM; out[3] = sumV; out[4] = sumLeafU; out[5] = sumImm; out[6] = sumSide; out[7] = sumSide - sumImm;
}

// Experiment only: how many nodes would be visited if the children of an inner node were walked in another
// order?  mode 0 = the reference's (left = larger surface area first); 1 = the child whose box centre comes first along
// the ray on the axis where the two centres differ most; 2 = the child whose box the ray enters first (exact entry
// distances).  The mask does not depend on the order (any-hit is an OR over the reachable triangles); the visit counts do.
// out: tiles, sum of per-tile union sizes, sum of per-tile longest ray, sum of visits over all rays, occluded rays.
extern "C" void orc_order_experiment(const uint32_t* packed, const float* constants, const void* light_v,
                                     const float* positions, uint32_t W, uint32_t H, int mode, uint64_t* out) {
    const OLight& lt = *(const OLight*)light_v;
    uint64_t tiles = 0, sumU = 0, sumM = 0, sumV = 0, occl = 0;
    const uint32_t TW = 8, TH = 8, tx = W / TW, ty = H / TH;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : tiles, sumU, sumM, sumV, occl)
#endif
    for (int64_t t = 0; t < (int64_t)tx * ty; ++t) {
        uint32_t bx = (uint32_t)(t % tx), by = (uint32_t)(t / tx);
        std::vector<u32> visited;
        u32 maxV = 0;
        for (u32 l = 0; l < TW * TH; ++l) {
            u32 x = bx * TW + (l % TW), y = by * TH + (l / TW);
            size_t pix = (size_t)y * W + x;
            V3 rel = { positions[pix * 4 + 0], positions[pix * 4 + 1], positions[pix * 4 + 2] };
            V3 o, d; float tmax;
            genRay(constants, rel, lt, 0, &o, &tmax, &d);
            V3 invdir = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
            u32 stack[128]; int sp = 0; stack[sp++] = 0;
            u32 v = 0; bool hit = false;
            while (sp > 0 && !hit) {
                const u32 node = stack[--sp];
                const u32* a = packed + (size_t)node * 8; const u32* b = a + 4;
                ++v; visited.push_back(node);
                if (a[3] != kInvalid) {
                    const u32* tt = packed + (size_t)a[3] * 4;
                    V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
                    V3 v0 = { u2f(tt[0]), u2f(tt[1]), u2f(tt[2]) };
                    hit = rayTri(o, tmax, d, v0, e0, e1);
                    continue;
                }
                V3 pmin = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, pmax = { u2f(b[0]), u2f(b[1]), u2f(b[2]) };
                if (!rayBox(o, invdir, pmin, pmax)) continue;
                const u32 left = node + 1, right = packed[(size_t)left * 8 + 7];
                bool leftFirst = true;
                if (mode != 0) {
                    const u32* la = packed + (size_t)left * 8; const u32* ra = packed + (size_t)right * 8;
                    const bool lLeaf = la[3] != kInvalid, rLeaf = ra[3] != kInvalid;
                    if (!lLeaf && !rLeaf) {
                        float lc[3], rc[3], lmin[3], rmin[3], lmax[3], rmax[3];
                        for (int k = 0; k < 3; ++k) {
                            lmin[k] = u2f(la[k]); lmax[k] = u2f(la[4 + k]); rmin[k] = u2f(ra[k]); rmax[k] = u2f(ra[4 + k]);
                            lc[k] = 0.5f * (lmin[k] + lmax[k]); rc[k] = 0.5f * (rmin[k] + rmax[k]);
                        }
                        const float dd[3] = { d.x, d.y, d.z }, oo[3] = { o.x, o.y, o.z }, ii[3] = { invdir.x, invdir.y, invdir.z };
                        if (mode == 1) {
                            int ax = 0; float best = -1.f;
                            for (int k = 0; k < 3; ++k) { float s = std::fabs(lc[k] - rc[k]); if (s > best) { best = s; ax = k; } }
                            leftFirst = (lc[ax] - rc[ax]) * dd[ax] <= 0.f;        // left centre comes first along the ray
                        } else {
                            float tl = 0.f, tr = 0.f;
                            for (int k = 0; k < 3; ++k) {
                                float l0 = (lmin[k] - oo[k]) * ii[k], l1 = (lmax[k] - oo[k]) * ii[k];
                                float r0 = (rmin[k] - oo[k]) * ii[k], r1 = (rmax[k] - oo[k]) * ii[k];
                                tl = std::max(tl, std::min(l0, l1)); tr = std::max(tr, std::min(r0, r1));
                            }
                            leftFirst = tl <= tr;
                        }
                    }
                }
                if (sp + 2 > 128) break;
                if (leftFirst) { stack[sp++] = right; stack[sp++] = left; } else { stack[sp++] = left; stack[sp++] = right; }
            }
            sumV += v; if (v > maxV) maxV = v; occl += hit ? 1 : 0;
        }
        std::sort(visited.begin(), visited.end());
        visited.erase(std::unique(visited.begin(), visited.end()), visited.end());
        sumU += visited.size(); sumM += maxV; ++tiles;
    }
    out[0] = tiles; out[1] = sumU; out[2] = sumM; out[3] = sumV; out[4] = occl;
}

// ---------------------------------------------------------------------------------------------
// Analysis helper (not a checker of the product; it checks a THEOREM the product's wide kernels rest on, and
// counts what such kernels would do).  A parent's box encloses its children's boxes (BVHBuilder.cpp:53-76: union by
// min/max, exact), and the slab test of comp:61-73 is monotone in the box when no NaN arises (IEEE subtraction,
// multiplication by a fixed 1/d and rounding are monotone): hit(child box) => hit(parent box).  So a ray reaches a leaf
// iff it hits the box of the leaf's PARENT, and the boxes in between need not be tested.  orc_wide_packet_sim walks every
// 8x8 tile as a packet over nodes that carry the boxes of their descendants `depth` levels down (depth 1: both children,
// 2: four grandchildren, 3: eight), with a stack of (node, member mask), testing triangles for the rays that hit the box
// of the leaf's parent; it compares the mask with anyHit() (out[7] = mismatching rays) and counts:
// out[0] tiles, out[1] steps (wide nodes entered, one dependent fetch each), out[2] box tests (wave-wide groups of 16
// VALU), out[3] member lanes summed over those tests, out[4] triangle tests (wave-wide), out[5] member lanes over them,
// out[6] longest tile (steps), out[7] mismatches, out[8] box tests if a slot repeating the previous slot's box is
// not tested again, out[9] sum over tiles of steps^2 (for the spread), out[10] rays whose 1/d is not finite-nonzero
// (excluded: the product sends such waves to the exact lane-per-ray walk); out[13] wave-wide triangle tests in which EVERY
// participating ray is rejected by the b1 condition alone, out[14] ... by the b1 / b2 / b1 + b2 conditions (no ray reaches the
// test of t), out[15] ... in which no ray hits.
// hist (nullable): 64 bins of steps per tile, bin = min(63, steps / histStep).
// ---------------------------------------------------------------------------------------------
namespace {
struct WSlot { u32 boxNode; u32 ref; bool leaf; };     // boxNode = inner node whose box is tested (kInvalid: none)
static void wideSlots(const u32* bvh, u32 n, u32 root, int d, std::vector<WSlot>& out, bool withRootBox = false) {
    const u32 left = n + 1, right = bvh[(size_t)left * 8 + 7];
    const u32 kids[2] = { left, right };
    for (u32 c : kids) {
        const bool leaf = bvh[(size_t)c * 8 + 3] != kInvalid;
        if (leaf) out.push_back(WSlot{ (n == root && !withRootBox) ? kInvalid : n, c, true });
        else if (d > 1) wideSlots(bvh, c, root, d - 1, out, withRootBox);
        else out.push_back(WSlot{ c, c, false });
    }
}
}

// depth 4 (14 with the cheap test): up to four slots chosen GREEDILY -- starting from the two children, the inner slot with the
// largest box surface is replaced by its two children until there are four slots (or only leaves): what a surface-area-driven
// collapse of the private copy would look like, against the fixed "two levels down" of depth 2.  (An experiment's count.)
static void wideSlotsGreedy(const u32* bvh, u32 n, std::vector<WSlot>& out, bool withRootBox) {
    struct Cand { u32 node; u32 parent; bool leaf; float area; };
    auto area = [&](u32 c) {
        const u32* a = bvh + (size_t)c * 8;
        const float dx = u2f(a[4]) - u2f(a[0]), dy = u2f(a[5]) - u2f(a[1]), dz = u2f(a[6]) - u2f(a[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    auto kids = [&](u32 m, Cand* k) {
        const u32 left = m + 1, right = bvh[(size_t)left * 8 + 7];
        const u32 cc[2] = { left, right };
        for (int i = 0; i < 2; ++i) {
            const bool leaf = bvh[(size_t)cc[i] * 8 + 3] != kInvalid;
            k[i] = Cand{ cc[i], m, leaf, leaf ? 0.f : area(cc[i]) };
        }
    };
    std::vector<Cand> cur(2);
    kids(n, cur.data());
    while (cur.size() < 4) {
        int best = -1;
        for (size_t i = 0; i < cur.size(); ++i) if (!cur[i].leaf && (best < 0 || cur[i].area > cur[best].area)) best = (int)i;
        if (best < 0) break;
        Cand k[2];
        kids(cur[best].node, k);
        cur[best] = k[0];
        cur.insert(cur.begin() + best + 1, k[1]);              // (stream order kept: left before right)
    }
    for (const Cand& c : cur) {
        if (c.leaf) out.push_back(WSlot{ (c.parent == n && !withRootBox) ? kInvalid : c.parent, c.node, true });
        else out.push_back(WSlot{ c.node, c.node, false });
    }
}

// Optional per-tile output of the next orc_wide_packet_sim call (tile t = by * (W / 8) + bx): nodes entered, box tests,
// member lanes summed over those tests.  NULL switches it off.
static uint32_t* g_tileSteps = nullptr; static uint32_t* g_tileTests = nullptr; static uint32_t* g_tileLanes = nullptr;
// More counts of the last orc_wide_packet_sim call: [0] triangle-test ROUNDS if every ray tested its own next leaf slot of a node
// in the same wave-wide test (per node: the largest number of leaf slots one ray has to test), [1] rays summed over those rounds,
// [2] wave-wide triangle tests with at most 16 participating rays, [3] ... with at most 8.
static uint64_t g_simExtra[4] = { 0, 0, 0, 0 };
extern "C" void orc_wide_packet_sim_extra(uint64_t* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_simExtra[i]; }
extern "C" void orc_wide_packet_sim_per_tile(uint32_t* steps, uint32_t* tests, uint32_t* lanes) {
    g_tileSteps = steps; g_tileTests = tests; g_tileLanes = lanes;
}

extern "C" void orc_wide_packet_sim(const uint32_t* packed, const float* constants, const void* light_v,
                                    const float* positions, uint32_t W, uint32_t H, int depth, uint32_t histStep,
                                    uint64_t* out, uint64_t* hist) {
    const OLight& lt = *(const OLight*)light_v;
    const u32* bvh = packed;
    // depth >= 10: the slots are culled with the CHEAP conservative test of the product's wide kernels (one fma per
    // plane against a per-ray constant that carries the slack, see wideRaySetup below) and a triangle hit only counts
    // after the exact test of the leaf's parent box; out[11] = tests where the exact form hits and the cheap one does
    // not (must be 0), out[12] = cheap hits that the exact form rejects (wasted work, not an error).
    // depth >= 100 (experiment's count): leaf tests are PARKED -- a ray remembers the leaf (one per ray) and walks on; when a ray
    // that already holds one meets another, every parked test of the tile is made at once (a "flush": each ray its own triangle).
    // g_simExtra[2] = flushes, [3] = rays summed over them (the counts of tests with <= 16 / <= 8 rays are not made then).
    const bool park = depth >= 100;
    if (park) depth -= 100;
    const bool cheap = depth >= 10;
    if (cheap) depth -= 10;
    uint64_t violations = 0, falsePos = 0;
    float rootLo[3] = { u2f(bvh[0]), u2f(bvh[1]), u2f(bvh[2]) }, rootHi[3] = { u2f(bvh[4]), u2f(bvh[5]), u2f(bvh[6]) };
    uint64_t tiles = 0, steps = 0, boxT = 0, boxLanes = 0, triT = 0, triLanes = 0, longest = 0, mism = 0, boxD = 0, sq = 0, unsafe = 0;
    uint64_t allB1 = 0, allBary = 0, allMiss = 0, rounds = 0, roundLanes = 0, tri16 = 0, tri8 = 0;
    const uint32_t tx = W / 8, ty = H / 8;
    std::vector<uint64_t> histAll(64, 0);
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        std::vector<uint64_t> histLoc(64, 0);
        std::vector<WSlot> slots;
        std::vector<std::pair<u32, uint64_t>> stack;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4) reduction(+ : tiles, steps, boxT, boxLanes, triT, triLanes, mism, boxD, sq, unsafe, violations, falsePos, allB1, allBary, allMiss, rounds, roundLanes, tri16, tri8) reduction(max : longest)
#endif
        for (int64_t t = 0; t < (int64_t)tx * ty; ++t) {
            const uint32_t bx = (uint32_t)(t % tx), by = (uint32_t)(t / tx);
            V3 o[64], d[64], inv[64]; float tm[64];
            float cUp[64][3], cDn[64][3];
            uint64_t live = 0, expect = 0;
            for (u32 l = 0; l < 64; ++l) {
                const size_t pix = (size_t)(by * 8 + l / 8) * W + bx * 8 + (l % 8);
                V3 rel = { positions[pix * 4 + 0], positions[pix * 4 + 1], positions[pix * 4 + 2] };
                genRay(constants, rel, lt, 0, &o[l], &tm[l], &d[l]);
                inv[l] = V3{ 1.0f / d[l].x, 1.0f / d[l].y, 1.0f / d[l].z };
                const bool safe = std::isfinite(o[l].x) && std::isfinite(o[l].y) && std::isfinite(o[l].z) &&
                                  std::isfinite(inv[l].x) && std::isfinite(inv[l].y) && std::isfinite(inv[l].z) &&
                                  inv[l].x != 0.f && inv[l].y != 0.f && inv[l].z != 0.f;
                if (!safe) { ++unsafe; continue; }
                if (cheap) {
                    // wideRaySetup: plane*inv - c with c = o*inv -/+ slack;  slack >= 3.5u*E*|inv| + 2u*|o*inv| + tiny
                    // covers both roundings of the exact form (fl(fl(P-o)*inv)) and the one of the fma, for every plane
                    // P inside the root box (E = largest |P - o| there); see DESIGN.md 4.7 for the derivation
                    const float oo[3] = { o[l].x, o[l].y, o[l].z }, ii[3] = { inv[l].x, inv[l].y, inv[l].z };
                    bool ok = true;
                    for (int a = 0; a < 3; ++a) {
                        const float E = fmaxf(fabsf(rootLo[a] - oo[a]), fabsf(rootHi[a] - oo[a]));
                        const float oi = oo[a] * ii[a];
                        const float slack = (E * fabsf(ii[a])) * 4.76837158e-7f + fabsf(oi) * 2.38418579e-7f + 7.5e-37f;   // 8u, 4u, ~2^-120
                        const float M = fmaxf(fabsf(rootLo[a]), fabsf(rootHi[a])) * fabsf(ii[a]);
                        cUp[l][a] = oi - slack; cDn[l][a] = oi + slack;
                        ok = ok && std::isfinite(slack) && M < 1e37f && fabsf(oi) < 1e37f && std::isfinite(cUp[l][a]) && std::isfinite(cDn[l][a]);
                    }
                    if (!ok) { ++unsafe; continue; }
                }
                live |= 1ull << l;
                if (anyHit(bvh, o[l], tm[l], d[l], nullptr, nullptr)) expect |= 1ull << l;
            }
            uint64_t occluded = 0, mySteps = 0;
            uint64_t parkedMask = 0; u32 parkedRef[64], parkedBox[64];
            const uint64_t boxT0 = boxT, boxLanes0 = boxLanes;
            auto box = [&](u32 n, u32 l) {
                const u32* a = bvh + (size_t)n * 8;
                V3 pmin = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, pmax = { u2f(a[4]), u2f(a[5]), u2f(a[6]) };
                return rayBox(o[l], inv[l], pmin, pmax);
            };
            auto boxCheap = [&](u32 n, u32 l) {
                const u32* a = bvh + (size_t)n * 8;
                const float lo[3] = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, hi[3] = { u2f(a[4]), u2f(a[5]), u2f(a[6]) };
                const float ii[3] = { inv[l].x, inv[l].y, inv[l].z };
                float t1 = INFINITY, t0 = 0.0f;
                for (int k = 0; k < 3; ++k) {
                    const float farP = ii[k] > 0.f ? hi[k] : lo[k], nearP = ii[k] > 0.f ? lo[k] : hi[k];
                    const float f = fmaf(farP, ii[k], -cUp[l][k]), nn = fmaf(nearP, ii[k], -cDn[l][k]);
                    t1 = fminf(t1, f); t0 = fmaxf(t0, nn);
                }
                return t1 >= t0;
            };
            auto tri = [&](u32 n, u32 l) {
                const u32* a = bvh + (size_t)n * 8; const u32* tt = bvh + (size_t)a[3] * 4;
                V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(a[4]), u2f(a[5]), u2f(a[6]) }, v0 = { u2f(tt[0]), u2f(tt[1]), u2f(tt[2]) };
                return rayTri(o[l], tm[l], d[l], v0, e0, e1);
            };
            stack.clear();
            if (bvh[3] != kInvalid) {                     // a single triangle: the root is a leaf
                for (u32 l = 0; l < 64; ++l) if (((live >> l) & 1) && tri(0, l)) occluded |= 1ull << l;
                ++triT; triLanes += __builtin_popcountll(live);
            } else {
                uint64_t m = cheap ? live : 0;
                if (!cheap) {
                    for (u32 l = 0; l < 64; ++l) if (((live >> l) & 1) && box(0, l)) m |= 1ull << l;
                    ++boxT; ++boxD; boxLanes += __builtin_popcountll(live);
                }
                if (m) stack.push_back({ 0u, m });
            }
            while (!stack.empty()) {
                const u32 n = stack.back().first; uint64_t m = stack.back().second & ~occluded; stack.pop_back();
                if (!m) continue;
                ++mySteps;
                slots.clear();
                if (depth == 4) wideSlotsGreedy(bvh, n, slots, cheap); else wideSlots(bvh, n, n, depth, slots, cheap);
                const size_t base = stack.size();
                u32 prevBox = kInvalid - 1;
                uint64_t prevHit = 0;
                unsigned char perLane[64] = { 0 };
                for (const WSlot& s : slots) {
                    uint64_t h = m & ~occluded;
                    if (s.boxNode != kInvalid) {
                        ++boxT; boxLanes += __builtin_popcountll(h);
                        if (s.boxNode == prevBox) h &= prevHit;
                        else {
                            ++boxD;
                            uint64_t hh = 0;
                            for (u32 l = 0; l < 64; ++l) if ((h >> l) & 1) {
                                if (!cheap) { if (box(s.boxNode, l)) hh |= 1ull << l; continue; }
                                const bool c = boxCheap(s.boxNode, l), e = box(s.boxNode, l);
                                if (e && !c) ++violations;
                                if (c && !e) ++falsePos;
                                if (c) hh |= 1ull << l;
                            }
                            prevBox = s.boxNode; prevHit = hh; h = hh;
                        }
                    }
                    if (!h) continue;
                    if (s.leaf && park) {
                        const u32 parentBox = s.boxNode != kInvalid ? s.boxNode : n;
                        if (h & parkedMask) {                                   // somebody already holds one: every parked test now
                            ++tri16; tri8 += __builtin_popcountll(parkedMask);
                            for (u32 l = 0; l < 64; ++l) if (((parkedMask >> l) & 1) && tri(parkedRef[l], l) && (!cheap || box(parkedBox[l], l))) occluded |= 1ull << l;
                            parkedMask = 0;
                            h &= ~occluded;
                        }
                        for (u32 l = 0; l < 64; ++l) if ((h >> l) & 1) { parkedRef[l] = s.ref; parkedBox[l] = parentBox; }
                        parkedMask |= h;
                        continue;
                    }
                    if (s.leaf) {
                        ++triT; triLanes += __builtin_popcountll(h);
                        for (u32 l = 0; l < 64; ++l) if ((h >> l) & 1) ++perLane[l];
                        if (__builtin_popcountll(h) <= 16) ++tri16;
                        if (__builtin_popcountll(h) <= 8) ++tri8;
                        {
                            const u32* a = bvh + (size_t)s.ref * 8; const u32* tt = bvh + (size_t)a[3] * 4;
                            const V3 e0 = { u2f(a[0]), u2f(a[1]), u2f(a[2]) }, e1 = { u2f(a[4]), u2f(a[5]), u2f(a[6]) }, v0 = { u2f(tt[0]), u2f(tt[1]), u2f(tt[2]) };
                            int worst = 3, best = 1;            // stages over the participating rays (1 = b1 ... 0 = hit)
                            bool hitAny = false;
                            for (u32 l = 0; l < 64; ++l) if ((h >> l) & 1) {
                                const int st = rayTriStage(o[l], tm[l], d[l], v0, e0, e1);
                                if (st == 0) hitAny = true; else { if (st > best) best = st; if (st < worst) worst = st; }
                            }
                            if (!hitAny) { ++allMiss; if (best == 1) ++allB1; if (best <= 2) ++allBary; }
                        }
                        for (u32 l = 0; l < 64; ++l) if (((h >> l) & 1) && tri(s.ref, l)) {
                            // cheap mode: the hit counts only if the ray hits the box of the leaf's parent (exact form)
                            const u32 parentBox = s.boxNode != kInvalid ? s.boxNode : n;
                            if (!cheap || box(parentBox, l)) occluded |= 1ull << l;
                        }
                    } else stack.push_back({ s.ref, h });
                }
                std::reverse(stack.begin() + base, stack.end());   // first slot on top: DFS order
                { unsigned mx = 0; for (u32 l = 0; l < 64; ++l) { roundLanes += perLane[l]; if (perLane[l] > mx) mx = perLane[l]; } rounds += mx; }
            }
            if (parkedMask) {                                                   // the walk is over: what is still parked
                ++tri16; tri8 += __builtin_popcountll(parkedMask);
                for (u32 l = 0; l < 64; ++l) if (((parkedMask >> l) & 1) && tri(parkedRef[l], l) && (!cheap || box(parkedBox[l], l))) occluded |= 1ull << l;
            }
            mism += __builtin_popcountll((occluded ^ expect) & live);
            steps += mySteps; sq += mySteps * mySteps; if (mySteps > longest) longest = mySteps; ++tiles;
            if (g_tileSteps) { g_tileSteps[t] = (uint32_t)mySteps; g_tileTests[t] = (uint32_t)(boxT - boxT0); g_tileLanes[t] = (uint32_t)(boxLanes - boxLanes0); }
            histLoc[std::min<uint64_t>(63, mySteps / (histStep ? histStep : 1))]++;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int i = 0; i < 64; ++i) histAll[i] += histLoc[i];
    }
    out[0] = tiles; out[1] = steps; out[2] = boxT; out[3] = boxLanes; out[4] = triT; out[5] = triLanes; out[6] = longest;
    out[7] = mism; out[8] = boxD; out[9] = sq; out[10] = unsafe; out[11] = violations; out[12] = falsePos;
    out[13] = allB1; out[14] = allBary; out[15] = allMiss;
    g_simExtra[0] = rounds; g_simExtra[1] = roundLanes; g_simExtra[2] = tri16; g_simExtra[3] = tri8;
    if (hist) for (int i = 0; i < 64; ++i) hist[i] = histAll[i];
}
