// BVH build on the GPU: an alternative PRODUCER of the packed node stream of SURVEY.md
// Appendix A.  Two topologies over the Morton order of the triangle centroids (63-bit codes, hipcub radix sort):
//   RTS_GPU_BUILD_LBVH  Karras' parallel hierarchy (one kernel) + bottom-up bounds: fastest build, weakest tree
//   RTS_GPU_BUILD_PLOC  parallel locally-ordered clustering (Meister & Bittner 2018): every cluster looks `radius`
//                       neighbours up and down the Morton order for the partner whose union has the smallest surface
//                       area, mutual pairs merge, the array is compacted, repeat until one cluster is left (default)
//   RTS_GPU_BUILD_PLOC_SAH  the same until at most 65 536 clusters are left, then the top of the tree over those clusters
//                       by the reference's own split rule -- full-sweep SAH on all three axes, BVHBuilder.cpp:78-156,
//                       with the clusters' triangle counts as weights -- on the host (the clusters' boxes travel, 1.8 MB)
// then the reference's own layout rules applied to that topology:
//   * child with the larger surface area first            (Source/BVHBuilder.cpp:202-208, strict `>` on the right one)
//   * depth-first (pre-order) numbering, left child = i+1 (cpp:222-238)
//   * miss link = first index after the subtree, 0xFFFFFFFF at the end (cpp:231-236)
//   * inner {bboxMin|0xFFFFFFFF}{bboxMax|next}, leaf {v1-v0|2N+prim}{v2-v0|next}, tail v0 per triangle (cpp:308-367)
// The TREE is not the reference's full-sweep SAH tree (that builder is bvh_builder.cpp, byte-identical to the oracle);
// any valid stream is a drop-in for the consumer, and masks agree with the SAH stream's up to the slab test's
// non-conservativeness.
//
// Synchronisation: every dependency between nodes crosses a KERNEL BOUNDARY (refit sweeps are repeated launches,
// numbering walks a finished tree), so nothing relies on in-kernel visibility between CUs / XCDs.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <algorithm>
#include <utility>
#include <vector>
#include "../../include/rts.h"

extern "C" int rts_ctx_adopt_device_bvh(rts_ctx* ctx, void* d_packed, size_t count_vec4, uint32_t prim_count);  // rts_api.cpp
extern "C" int rts_ctx_device_ordinal(rts_ctx* ctx);

namespace {

constexpr uint32_t END = 0xFFFFFFFFu;

struct Lbvh {
    uint32_t P;
    const float* verts; uint32_t stride; const uint32_t* indices;
    float* leafLo; float* leafHi;          // 3 floats per triangle (by prim id)
    uint32_t* sceneBox;                    // 6 order-preserving encoded floats: min xyz, max xyz
    uint64_t* keys; uint32_t* order;       // Morton key / prim id, sorted position -> prim
    uint32_t* child;                       // 2 per internal node: node ids (internal i = i, leaf at sorted pos j = P-1+j)
    uint32_t* parent;                      // per node id
    float* nodeLo; float* nodeHi;          // 3 floats per internal node
    uint32_t* leaves;                      // triangles below, per internal node
    uint32_t* done;                        // per internal node: bounds final
    uint32_t* pending;                     // [0] = internal nodes not final yet
    uint32_t* flags;                       // [0] = a non-finite vertex was seen
};

__device__ __forceinline__ uint32_t encodeOrdered(float f) {          // monotone float -> uint map
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float decodeOrdered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void leafBoxesKernel(Lbvh b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.P) return;
    float lo[3], hi[3];
    bool finite = true;
    for (int c = 0; c < 3; ++c) {
        const float* v = b.verts + (size_t)b.stride * b.indices[(size_t)p * 3 + c];
        for (int k = 0; k < 3; ++k) {
            const float x = v[k];
            finite = finite && (__builtin_fabsf(x) < __builtin_inff());
            lo[k] = (c == 0 || x < lo[k]) ? x : lo[k];
            hi[k] = (c == 0 || hi[k] < x) ? x : hi[k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        b.leafLo[(size_t)p * 3 + k] = lo[k];
        b.leafHi[(size_t)p * 3 + k] = hi[k];
        const float c = (lo[k] + hi[k]) * 0.5f;                          // centroid as in BVHBuilder (Box3::center)
        atomicMin(&b.sceneBox[k], encodeOrdered(c));
        atomicMax(&b.sceneBox[3 + k], encodeOrdered(c));
    }
    if (!finite) b.flags[0] = 1;
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {                 // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void mortonKernel(Lbvh b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.P) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float mn = decodeOrdered(b.sceneBox[k]), mx = decodeOrdered(b.sceneBox[3 + k]);
        const float c = (b.leafLo[(size_t)p * 3 + k] + b.leafHi[(size_t)p * 3 + k]) * 0.5f;
        const float ext = mx - mn;
        float t = ext > 0.0f ? (c - mn) / ext : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        uint32_t v = (uint32_t)(t * 2097151.0f);
        q[k] = v > 2097151u ? 2097151u : v;
    }
    b.keys[p] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    b.order[p] = p;
}

// Karras 2012: length of the common prefix of keys i and j (ties broken by the position, so all keys are distinct)
__device__ __forceinline__ int commonPrefix(const uint64_t* keys, uint32_t P, int i, int j) {
    if (j < 0 || j >= (int)P) return -1;
    const uint64_t a = keys[i], c = keys[j];
    if (a == c) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ c));
}

__global__ void hierarchyKernel(Lbvh b, const uint64_t* keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = (int)b.P;
    if (i >= P - 1) return;
    const int d = commonPrefix(keys, b.P, i, i + 1) - commonPrefix(keys, b.P, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = commonPrefix(keys, b.P, i, i - d);
    int lmax = 2;
    while (commonPrefix(keys, b.P, i, i + lmax * d) > dmin) lmax *= 2;       // bounded: prefix is -1 outside [0,P)
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (commonPrefix(keys, b.P, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = commonPrefix(keys, b.P, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (commonPrefix(keys, b.P, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t left = (lo == gamma) ? (uint32_t)(P - 1 + gamma) : (uint32_t)gamma;
    const uint32_t right = (hi == gamma + 1) ? (uint32_t)(P - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    b.child[2 * i] = left;
    b.child[2 * i + 1] = right;
    b.parent[left] = (uint32_t)i;
    b.parent[right] = (uint32_t)i;
    if (i == 0) b.parent[0] = END;
}

__device__ __forceinline__ float surfaceArea(const float* lo, const float* hi) {   // BVHBuilder.cpp:24-28
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ey + ey * ez + ez * ex) * 2.0f;
}

// One sweep: every internal node whose children were final BEFORE this launch becomes final.
__global__ void refitSweepKernel(Lbvh b, const uint32_t* order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.P - 1 || b.done[i] == 1) return;
    uint32_t c[2] = { b.child[2 * i], b.child[2 * i + 1] };
    float lo[2][3], hi[2][3];
    uint32_t n[2];
    for (int k = 0; k < 2; ++k) {
        if (c[k] >= b.P - 1) {                                   // leaf at sorted position c-(P-1)
            const uint32_t prim = order[c[k] - (b.P - 1)];
            for (int a = 0; a < 3; ++a) { lo[k][a] = b.leafLo[(size_t)prim * 3 + a]; hi[k][a] = b.leafHi[(size_t)prim * 3 + a]; }
            n[k] = 1;
        } else {
            if (b.done[c[k]] != 1) return;                       // not yet: a later sweep
            for (int a = 0; a < 3; ++a) { lo[k][a] = b.nodeLo[(size_t)c[k] * 3 + a]; hi[k][a] = b.nodeHi[(size_t)c[k] * 3 + a]; }
            n[k] = b.leaves[c[k]];
        }
    }
    if (surfaceArea(lo[1], hi[1]) > surfaceArea(lo[0], hi[0])) {  // larger child first (cpp:202-208)
        b.child[2 * i] = c[1];
        b.child[2 * i + 1] = c[0];
    }
    for (int a = 0; a < 3; ++a) {
        b.nodeLo[(size_t)i * 3 + a] = lo[0][a] < lo[1][a] ? lo[0][a] : lo[1][a];
        b.nodeHi[(size_t)i * 3 + a] = hi[0][a] > hi[1][a] ? hi[0][a] : hi[1][a];
    }
    b.leaves[i] = n[0] + n[1];
    b.done[i] = 2;                                               // final, becomes visible as 1 after this launch
    atomicSub(&b.pending[0], 1u);
}
__global__ void refitCommitKernel(Lbvh b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.P - 1 && b.done[i] == 2) b.done[i] = 1;
}

// ---- PLOC ------------------------------------------------------------------------------------------------------
// Cluster arrays (position in the current Morton-ordered list -> node id and box).  Node ids as everywhere in this
// file: internal 0..P-2, leaf at sorted position j = P-1+j.
struct Ploc {
    uint32_t n;                            // clusters in the list
    uint32_t radius;
    uint32_t* id; float* lo; float* hi;    // current list (3 floats per box)
    uint32_t* idOut; float* loOut; float* hiOut;   // compacted list of the next round
    uint32_t* nn;                          // nearest neighbour (position) of every cluster
    uint32_t* valid;                       // 1: the cluster at this position survives the round
    uint32_t* pos;                         // exclusive scan of valid
    uint32_t* nextId;                      // [0] = next free internal node id
};

__global__ void plocInitKernel(Lbvh b, Ploc c, const uint32_t* order) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.P) return;
    const uint32_t prim = order[j];
    c.id[j] = b.P - 1 + j;
    for (int a = 0; a < 3; ++a) { c.lo[(size_t)j * 3 + a] = b.leafLo[(size_t)prim * 3 + a]; c.hi[(size_t)j * 3 + a] = b.leafHi[(size_t)prim * 3 + a]; }
}

// Nearest neighbour within `radius` positions: smallest surface area of the union; of equal areas the nearer position
// wins, the lower one at equal distance (regular meshes are full of exact ties: preferring the far end of the window
// pairs clusters across the grid and doubled the traversal cost of the atrium).  The pair with the globally smallest
// area -- smallest gap, then lowest position among those -- is always mutual, so every round merges something.
__global__ void plocNearestKernel(Ploc c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = c.lo[(size_t)i * 3 + a]; hi[a] = c.hi[(size_t)i * 3 + a]; }
    float best = __builtin_inff();
    uint32_t bestJ = END;
    for (uint32_t d = 1; d <= c.radius; ++d) {
        for (int side = 0; side < 2; ++side) {
            if (side == 0 ? i < d : i + d >= c.n) continue;
            const uint32_t j = side == 0 ? i - d : i + d;
            float ulo[3], uhi[3];
            for (int a = 0; a < 3; ++a) {
                const float l = c.lo[(size_t)j * 3 + a], h = c.hi[(size_t)j * 3 + a];
                ulo[a] = l < lo[a] ? l : lo[a];
                uhi[a] = h > hi[a] ? h : hi[a];
            }
            const float area = surfaceArea(ulo, uhi);
            if (area < best || bestJ == END) { best = area; bestJ = j; }      // (also takes the first candidate when every area is inf/NaN)
        }
    }
    c.nn[i] = bestJ;
}

// Mutual pairs merge into a new internal node at the lower position; the higher position is dropped.
__global__ void plocMergeKernel(Lbvh b, Ploc c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    const uint32_t j = c.nn[i];
    const bool mutual = j != END && c.nn[j] == i;
    if (!mutual) { c.valid[i] = 1; return; }
    if (i > j) { c.valid[i] = 0; return; }
    c.valid[i] = 1;
    const uint32_t node = atomicAdd(&c.nextId[0], 1u);
    uint32_t kid[2] = { c.id[i], c.id[j] };
    float lo[2][3], hi[2][3];
    for (int a = 0; a < 3; ++a) {
        lo[0][a] = c.lo[(size_t)i * 3 + a]; hi[0][a] = c.hi[(size_t)i * 3 + a];
        lo[1][a] = c.lo[(size_t)j * 3 + a]; hi[1][a] = c.hi[(size_t)j * 3 + a];
    }
    const bool swap = surfaceArea(lo[1], hi[1]) > surfaceArea(lo[0], hi[0]);    // larger child first (cpp:202-208)
    b.child[2 * node] = kid[swap ? 1 : 0];
    b.child[2 * node + 1] = kid[swap ? 0 : 1];
    b.parent[kid[0]] = node;
    b.parent[kid[1]] = node;
    uint32_t leaves = 0;
    for (int k = 0; k < 2; ++k) leaves += kid[k] >= b.P - 1 ? 1u : b.leaves[kid[k]];
    b.leaves[node] = leaves;
    for (int a = 0; a < 3; ++a) {
        const float l = lo[0][a] < lo[1][a] ? lo[0][a] : lo[1][a], h = hi[0][a] > hi[1][a] ? hi[0][a] : hi[1][a];
        b.nodeLo[(size_t)node * 3 + a] = l; b.nodeHi[(size_t)node * 3 + a] = h;
        c.lo[(size_t)i * 3 + a] = l; c.hi[(size_t)i * 3 + a] = h;          // position i now holds the merged cluster
    }
    c.id[i] = node;
}

__global__ void plocCompactKernel(Ploc c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n || !c.valid[i]) return;
    const uint32_t o = c.pos[i];
    c.idOut[o] = c.id[i];
    for (int a = 0; a < 3; ++a) { c.loOut[(size_t)o * 3 + a] = c.lo[(size_t)i * 3 + a]; c.hiOut[(size_t)o * 3 + a] = c.hi[(size_t)i * 3 + a]; }
}

__global__ void plocRootKernel(Lbvh b, Ploc c) {                       // the last cluster is the root
    if (blockIdx.x == 0 && threadIdx.x == 0) b.parent[c.id[0]] = END;
}

// ---- top of the tree over PLOC clusters (host) -------------------------------------------------------------------
// Input: n clusters (node id, box, triangles below).  Output: n - 1 internal nodes with ids firstId .. firstId + n - 2
// (a parent's id is lower than its children's), children ordered "larger surface area first" (cpp:202-208).
struct TopTree {
    std::vector<uint32_t> child, leaves, pairs;     // 2 per node; 1 per node; (child id, parent id) per link
    std::vector<float> lo, hi;                      // 3 per node
    uint32_t root = 0;
};

static float hostArea(const float* lo, const float* hi) {
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ey + ey * ez + ez * ex) * 2.0f;
}

static void buildTopSah(uint32_t n, const uint32_t* id, const float* lo, const float* hi, const uint32_t* weight,
                        uint32_t firstId, TopTree* out) {
    const uint32_t inner = n - 1;
    out->child.assign((size_t)inner * 2, 0); out->leaves.assign(inner, 0);
    out->lo.assign((size_t)inner * 3, 0.f); out->hi.assign((size_t)inner * 3, 0.f);
    std::vector<float> ctr((size_t)n * 3);
    for (size_t i = 0; i < (size_t)n * 3; ++i) ctr[i] = (lo[i] + hi[i]) * 0.5f;
    // three index arrays, each sorted once by the centroid on its axis (ties by position: deterministic); a split keeps
    // them sorted by partitioning the two other arrays stably: O(n) per level instead of a sort per node
    std::vector<uint32_t> ord[3], tmp(n);
    for (int axis = 0; axis < 3; ++axis) {
        ord[axis].resize(n);
        for (uint32_t i = 0; i < n; ++i) ord[axis][i] = i;
        std::stable_sort(ord[axis].begin(), ord[axis].end(), [&](uint32_t x, uint32_t y) { return ctr[(size_t)x * 3 + axis] < ctr[(size_t)y * 3 + axis]; });
    }
    std::vector<uint8_t> left(n, 0);
    std::vector<float> saR(n);
    std::vector<double> wR(n);
    struct Range { uint32_t begin, end, parent, side; };           // parent = index of the new node (0-based), ~0u for the root
    std::vector<Range> work;
    work.push_back(Range{ 0, n, 0xFFFFFFFFu, 0 });
    uint32_t next = 0;
    auto link = [&](const Range& r, uint32_t ref) {
        if (r.parent == 0xFFFFFFFFu) out->root = ref; else out->child[(size_t)r.parent * 2 + r.side] = ref;
    };
    while (!work.empty()) {
        const Range r = work.back();
        work.pop_back();
        const uint32_t cnt = r.end - r.begin;
        if (cnt == 1) { link(r, id[ord[0][r.begin]]); continue; }
        float bestCost = __builtin_inff();
        uint32_t bestMid = cnt / 2; int bestAxis = 0;              // (if every cost overflows: the median on x)
        for (int axis = 0; axis < 3; ++axis) {
            const uint32_t* perm = ord[axis].data() + r.begin;
            float blo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, bhi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
            double w = 0;
            for (uint32_t k = cnt; k-- > 0;) {                   // suffix areas and weights
                const uint32_t c = perm[k];
                for (int a = 0; a < 3; ++a) { blo[a] = lo[(size_t)c * 3 + a] < blo[a] ? lo[(size_t)c * 3 + a] : blo[a]; bhi[a] = hi[(size_t)c * 3 + a] > bhi[a] ? hi[(size_t)c * 3 + a] : bhi[a]; }
                w += weight[c];
                saR[k] = hostArea(blo, bhi); wR[k] = w;
            }
            for (int a = 0; a < 3; ++a) { blo[a] = __builtin_inff(); bhi[a] = -__builtin_inff(); }
            w = 0;
            for (uint32_t k = 0; k + 1 < cnt; ++k) {             // prefix sweep: split after position k
                const uint32_t c = perm[k];
                for (int a = 0; a < 3; ++a) { blo[a] = lo[(size_t)c * 3 + a] < blo[a] ? lo[(size_t)c * 3 + a] : blo[a]; bhi[a] = hi[(size_t)c * 3 + a] > bhi[a] ? hi[(size_t)c * 3 + a] : bhi[a]; }
                w += weight[c];
                const float cost = (float)(hostArea(blo, bhi) * w + saR[k + 1] * wR[k + 1]);
                if (cost < bestCost) { bestCost = cost; bestMid = k + 1; bestAxis = axis; }
            }
        }
        for (uint32_t k = 0; k < cnt; ++k) left[ord[bestAxis][r.begin + k]] = k < bestMid ? 1 : 0;
        for (int axis = 0; axis < 3; ++axis) {
            if (axis == bestAxis) continue;
            uint32_t* perm = ord[axis].data() + r.begin;
            uint32_t nl = 0, nr = 0;
            for (uint32_t k = 0; k < cnt; ++k) { if (left[perm[k]]) perm[nl++] = perm[k]; else tmp[nr++] = perm[k]; }
            std::copy(tmp.begin(), tmp.begin() + nr, perm + nl);
        }
        const uint32_t node = next++;
        link(r, firstId + node);
        work.push_back(Range{ r.begin + bestMid, r.end, node, 1 });
        work.push_back(Range{ r.begin, r.begin + bestMid, node, 0 });
    }
    // boxes, counts, child order: children have higher ids than their parent, so walk the new nodes from the last to the first
    std::vector<float> clo((size_t)n * 3), chi((size_t)n * 3);
    auto boxOf = [&](uint32_t ref, const float** l, const float** h, uint32_t* cntOut, const std::vector<uint32_t>& where) {
        if (ref >= firstId && ref < firstId + inner) { const uint32_t k = ref - firstId; *l = &out->lo[(size_t)k * 3]; *h = &out->hi[(size_t)k * 3]; *cntOut = out->leaves[k]; }
        else { const uint32_t c = where[ref]; *l = &lo[(size_t)c * 3]; *h = &hi[(size_t)c * 3]; *cntOut = weight[c]; }
    };
    (void)clo; (void)chi;
    std::vector<uint32_t> where;                                   // node id -> cluster position
    {
        uint32_t maxId = 0;
        for (uint32_t i = 0; i < n; ++i) maxId = id[i] > maxId ? id[i] : maxId;
        where.assign((size_t)maxId + 1, 0);
        for (uint32_t i = 0; i < n; ++i) where[id[i]] = i;
    }
    out->pairs.clear();
    for (uint32_t k = inner; k-- > 0;) {
        uint32_t* ch = &out->child[(size_t)k * 2];
        const float *l0, *h0, *l1, *h1; uint32_t c0, c1;
        boxOf(ch[0], &l0, &h0, &c0, where); boxOf(ch[1], &l1, &h1, &c1, where);
        for (int a = 0; a < 3; ++a) { out->lo[(size_t)k * 3 + a] = l0[a] < l1[a] ? l0[a] : l1[a]; out->hi[(size_t)k * 3 + a] = h0[a] > h1[a] ? h0[a] : h1[a]; }
        out->leaves[k] = c0 + c1;
        if (hostArea(l1, h1) > hostArea(l0, h0)) std::swap(ch[0], ch[1]);       // larger child first (cpp:202-208)
        out->pairs.push_back(ch[0]); out->pairs.push_back(firstId + k);
        out->pairs.push_back(ch[1]); out->pairs.push_back(firstId + k);
    }
}

__global__ void setParentsKernel(Lbvh b, const uint32_t* pairs, uint32_t nPairs, uint32_t root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nPairs) b.parent[pairs[2 * i]] = pairs[2 * i + 1];
    if (i == 0) b.parent[root] = END;
}

// Pre-order index of a node = sum over its ancestors of (1 + size of the sibling subtree visited before it).
__global__ void emitKernel(Lbvh b, const uint32_t* order, uint32_t* packed) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t P = b.P, N = 2 * P - 1;
    if (id >= N) return;
    const bool leaf = id >= P - 1;
    const uint32_t size = leaf ? 1u : 2u * b.leaves[id] - 1u;
    uint32_t index = 0, cur = id;
    for (uint32_t guard = 0; guard < 4096 && b.parent[cur] != END; ++guard) {
        const uint32_t par = b.parent[cur];
        index += 1;
        if (b.child[2 * par + 1] == cur) {                       // second child: the first one's subtree comes before
            const uint32_t first = b.child[2 * par];
            index += first >= P - 1 ? 1u : 2u * b.leaves[first] - 1u;
        }
        cur = par;
    }
    const uint32_t next = index + size >= N ? END : index + size;
    uint32_t* o = packed + (size_t)index * 8;
    if (leaf) {
        const uint32_t prim = order[id - (P - 1)];
        const float* v0 = b.verts + (size_t)b.stride * b.indices[(size_t)prim * 3 + 0];
        const float* v1 = b.verts + (size_t)b.stride * b.indices[(size_t)prim * 3 + 1];
        const float* v2 = b.verts + (size_t)b.stride * b.indices[(size_t)prim * 3 + 2];
        o[0] = __float_as_uint(v1[0] - v0[0]); o[1] = __float_as_uint(v1[1] - v0[1]); o[2] = __float_as_uint(v1[2] - v0[2]);
        o[3] = 2 * N + prim;
        o[4] = __float_as_uint(v2[0] - v0[0]); o[5] = __float_as_uint(v2[1] - v0[1]); o[6] = __float_as_uint(v2[2] - v0[2]);
        o[7] = next;
        uint32_t* t = packed + ((size_t)2 * N + prim) * 4;
        t[0] = __float_as_uint(v0[0]); t[1] = __float_as_uint(v0[1]); t[2] = __float_as_uint(v0[2]); t[3] = 0;
    } else {
        for (int a = 0; a < 3; ++a) { o[a] = __float_as_uint(b.nodeLo[(size_t)id * 3 + a]); o[4 + a] = __float_as_uint(b.nodeHi[(size_t)id * 3 + a]); }
        o[3] = END;
        o[7] = next;
    }
}

__global__ void emitSingleKernel(Lbvh b, uint32_t* packed) {               // P == 1: the root is the leaf
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float* v0 = b.verts + (size_t)b.stride * b.indices[0];
    const float* v1 = b.verts + (size_t)b.stride * b.indices[1];
    const float* v2 = b.verts + (size_t)b.stride * b.indices[2];
    packed[0] = __float_as_uint(v1[0] - v0[0]); packed[1] = __float_as_uint(v1[1] - v0[1]); packed[2] = __float_as_uint(v1[2] - v0[2]);
    packed[3] = 2;
    packed[4] = __float_as_uint(v2[0] - v0[0]); packed[5] = __float_as_uint(v2[1] - v0[1]); packed[6] = __float_as_uint(v2[2] - v0[2]);
    packed[7] = END;
    packed[8] = __float_as_uint(v0[0]); packed[9] = __float_as_uint(v0[1]); packed[10] = __float_as_uint(v0[2]); packed[11] = 0;
}

struct DeviceArena {            // frees everything it handed out, whatever path leaves the function
    void* ptrs[40]; int n = 0;
    hipEvent_t ev[2] = { nullptr, nullptr };
    template <typename T> hipError_t get(T** p, size_t bytes) {
        void* v = nullptr;
        hipError_t e = hipMalloc(&v, bytes ? bytes : 16);
        if (e == hipSuccess) { ptrs[n++] = v; *p = (T*)v; }
        return e;
    }
    void release(void* keep = nullptr) {
        for (int i = 0; i < n; ++i) if (ptrs[i] != keep) (void)hipFree(ptrs[i]);
        n = 0;
        for (hipEvent_t& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
};

#define LB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { arena.release(); return RTS_ERR_HIP + (int)e_; } } while (0)

} // namespace

extern "C" int rts_bvh_build_device_ex(rts_ctx* ctx, const float* vertices, size_t vertex_floats, uint32_t stride,
                                       const uint32_t* indices, uint32_t P, int algorithm, uint32_t radius,
                                       rts_vec4u* out_packed, size_t out_cap, int install, float* build_ms) {
    if (!ctx || !vertices || !indices || P == 0 || stride < 3 || P > 0x0CCCCCCCu) return RTS_ERR_INVALID_ARG;
    if ((algorithm != RTS_GPU_BUILD_LBVH && algorithm != RTS_GPU_BUILD_PLOC && algorithm != RTS_GPU_BUILD_PLOC_SAH) || radius > 256)
        return RTS_ERR_INVALID_ARG;
    if (radius == 0) radius = 16;
    const size_t count = (size_t)5 * P - 2;
    if (out_packed && out_cap < count) return RTS_ERR_CAPACITY;
    for (size_t i = 0; i < (size_t)3 * P; ++i)
        if ((size_t)indices[i] * stride + 3 > vertex_floats) return RTS_ERR_INVALID_ARG;
    hipError_t e0 = hipSetDevice(rts_ctx_device_ordinal(ctx));
    if (e0 != hipSuccess) return RTS_ERR_HIP + (int)e0;

    DeviceArena arena;
    Lbvh b{};
    b.P = P; b.stride = stride;
    float* d_verts; uint32_t* d_idx; uint32_t* d_packed;
    uint64_t* keysAlt; uint32_t* orderAlt;
    LB_HIP(arena.get(&d_verts, vertex_floats * 4));
    LB_HIP(arena.get(&d_idx, (size_t)P * 12));
    LB_HIP(arena.get(&d_packed, count * 16 + 64));
    LB_HIP(arena.get(&b.leafLo, (size_t)P * 12)); LB_HIP(arena.get(&b.leafHi, (size_t)P * 12));
    LB_HIP(arena.get(&b.sceneBox, 32));
    LB_HIP(arena.get(&b.keys, (size_t)P * 8)); LB_HIP(arena.get(&keysAlt, (size_t)P * 8));
    LB_HIP(arena.get(&b.order, (size_t)P * 4)); LB_HIP(arena.get(&orderAlt, (size_t)P * 4));
    LB_HIP(arena.get(&b.child, (size_t)P * 8)); LB_HIP(arena.get(&b.parent, (size_t)P * 8));
    LB_HIP(arena.get(&b.nodeLo, (size_t)P * 12)); LB_HIP(arena.get(&b.nodeHi, (size_t)P * 12));
    LB_HIP(arena.get(&b.leaves, (size_t)P * 4)); LB_HIP(arena.get(&b.done, (size_t)P * 4));
    LB_HIP(arena.get(&b.pending, 16)); LB_HIP(arena.get(&b.flags, 16));
    b.verts = d_verts; b.indices = d_idx;

    LB_HIP(hipMemcpy(d_verts, vertices, vertex_floats * 4, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(d_idx, indices, (size_t)P * 12, hipMemcpyHostToDevice));
    LB_HIP(hipEventCreate(&arena.ev[0])); LB_HIP(hipEventCreate(&arena.ev[1]));
    const hipEvent_t ev0 = arena.ev[0], ev1 = arena.ev[1];
    LB_HIP(hipEventRecord(ev0, nullptr));

    const uint32_t boxInit[8] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0 };
    LB_HIP(hipMemcpy(b.sceneBox, boxInit, 32, hipMemcpyHostToDevice));
    LB_HIP(hipMemset(b.flags, 0, 16));
    LB_HIP(hipMemset(b.done, 0, (size_t)P * 4));
    const uint32_t pend = P - 1;
    LB_HIP(hipMemcpy(b.pending, &pend, 4, hipMemcpyHostToDevice));
    const dim3 block(256), gridP((P + 255) / 256), gridN((2 * P - 1 + 255) / 256);
    hipLaunchKernelGGL(leafBoxesKernel, gridP, block, 0, nullptr, b);
    uint32_t flag = 0;
    LB_HIP(hipMemcpy(&flag, b.flags, 4, hipMemcpyDeviceToHost));
    if (flag) { arena.release(); return RTS_ERR_NONFINITE; }

    const uint64_t* sortedKeys = b.keys;
    const uint32_t* sortedOrder = b.order;
    if (P > 1) {
        hipLaunchKernelGGL(mortonKernel, gridP, block, 0, nullptr, b);
        size_t tempBytes = 0;
        LB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, b.keys, keysAlt, b.order, orderAlt, (int)P, 0, 63, nullptr));
        void* temp;
        LB_HIP(arena.get(&temp, tempBytes));
        LB_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, b.keys, keysAlt, b.order, orderAlt, (int)P, 0, 63, nullptr));
        sortedKeys = keysAlt; sortedOrder = orderAlt;
        if (algorithm == RTS_GPU_BUILD_PLOC || algorithm == RTS_GPU_BUILD_PLOC_SAH) {
            const uint32_t stopAt = algorithm == RTS_GPU_BUILD_PLOC_SAH ? 65536u : 1u;
            Ploc c{};
            c.n = P; c.radius = radius;
            LB_HIP(arena.get(&c.id, (size_t)P * 4)); LB_HIP(arena.get(&c.idOut, (size_t)P * 4));
            LB_HIP(arena.get(&c.lo, (size_t)P * 12)); LB_HIP(arena.get(&c.hi, (size_t)P * 12));
            LB_HIP(arena.get(&c.loOut, (size_t)P * 12)); LB_HIP(arena.get(&c.hiOut, (size_t)P * 12));
            LB_HIP(arena.get(&c.nn, (size_t)P * 4)); LB_HIP(arena.get(&c.valid, (size_t)P * 4)); LB_HIP(arena.get(&c.pos, (size_t)P * 4));
            LB_HIP(arena.get(&c.nextId, 16));
            LB_HIP(hipMemset(c.nextId, 0, 16));
            size_t scanBytes = 0;
            LB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, c.valid, c.pos, (int)P, nullptr));
            void* scanTemp;
            LB_HIP(arena.get(&scanTemp, scanBytes));
            hipLaunchKernelGGL(plocInitKernel, gridP, block, 0, nullptr, b, c, sortedOrder);
            for (uint32_t round = 0; c.n > stopAt; ++round) {
                if (round > 4u * 1024u * 1024u) { arena.release(); return RTS_ERR_BAD_BVH; }      // (every round merges at least one pair)
                const dim3 grid((c.n + 255) / 256);
                hipLaunchKernelGGL(plocNearestKernel, grid, block, 0, nullptr, c);
                hipLaunchKernelGGL(plocMergeKernel, grid, block, 0, nullptr, b, c);
                LB_HIP(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, c.valid, c.pos, (int)c.n, nullptr));
                hipLaunchKernelGGL(plocCompactKernel, grid, block, 0, nullptr, c);
                uint32_t tail[2] = { 0, 0 };
                LB_HIP(hipMemcpy(&tail[0], c.pos + (c.n - 1), 4, hipMemcpyDeviceToHost));
                LB_HIP(hipMemcpy(&tail[1], c.valid + (c.n - 1), 4, hipMemcpyDeviceToHost));
                const uint32_t next = tail[0] + tail[1];
                if (next == 0 || next >= c.n) { arena.release(); return RTS_ERR_BAD_BVH; }
                c.n = next;
                std::swap(c.id, c.idOut); std::swap(c.lo, c.loOut); std::swap(c.hi, c.hiOut);
            }
            if (c.n == 1) {
                hipLaunchKernelGGL(plocRootKernel, dim3(1), dim3(64), 0, nullptr, b, c);
            } else {
                // the top of the tree over the remaining clusters: full-sweep SAH on the host
                try {
                    const uint32_t n = c.n;
                    std::vector<uint32_t> ids(n), allLeaves(P), weight(n);
                    std::vector<float> lo((size_t)n * 3), hi((size_t)n * 3);
                    uint32_t firstId = 0;
                    LB_HIP(hipMemcpy(ids.data(), c.id, (size_t)n * 4, hipMemcpyDeviceToHost));
                    LB_HIP(hipMemcpy(lo.data(), c.lo, (size_t)n * 12, hipMemcpyDeviceToHost));
                    LB_HIP(hipMemcpy(hi.data(), c.hi, (size_t)n * 12, hipMemcpyDeviceToHost));
                    LB_HIP(hipMemcpy(allLeaves.data(), b.leaves, (size_t)(P - 1) * 4, hipMemcpyDeviceToHost));
                    LB_HIP(hipMemcpy(&firstId, c.nextId, 4, hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < n; ++i) weight[i] = ids[i] >= P - 1 ? 1u : allLeaves[ids[i]];
                    TopTree top;
                    buildTopSah(n, ids.data(), lo.data(), hi.data(), weight.data(), firstId, &top);
                    uint32_t* d_pairs;
                    LB_HIP(arena.get(&d_pairs, top.pairs.size() * 4));
                    LB_HIP(hipMemcpy(d_pairs, top.pairs.data(), top.pairs.size() * 4, hipMemcpyHostToDevice));
                    LB_HIP(hipMemcpy(b.child + (size_t)firstId * 2, top.child.data(), top.child.size() * 4, hipMemcpyHostToDevice));
                    LB_HIP(hipMemcpy(b.leaves + firstId, top.leaves.data(), top.leaves.size() * 4, hipMemcpyHostToDevice));
                    LB_HIP(hipMemcpy(b.nodeLo + (size_t)firstId * 3, top.lo.data(), top.lo.size() * 4, hipMemcpyHostToDevice));
                    LB_HIP(hipMemcpy(b.nodeHi + (size_t)firstId * 3, top.hi.data(), top.hi.size() * 4, hipMemcpyHostToDevice));
                    const uint32_t nPairs = (uint32_t)(top.pairs.size() / 2);
                    hipLaunchKernelGGL(setParentsKernel, dim3((nPairs + 255) / 256), block, 0, nullptr, b, d_pairs, nPairs, top.root);
                } catch (...) {
                    arena.release();
                    return RTS_ERR_CAPACITY;
                }
            }
        } else {
            hipLaunchKernelGGL(hierarchyKernel, gridP, block, 0, nullptr, b, sortedKeys);
            // bottom-up bounds: repeated sweeps, each finalising the nodes whose children were final before it
            uint32_t left = pend;
            for (int sweep = 0; sweep < 4096 && left != 0; ++sweep) {
                hipLaunchKernelGGL(refitSweepKernel, gridP, block, 0, nullptr, b, sortedOrder);
                hipLaunchKernelGGL(refitCommitKernel, gridP, block, 0, nullptr, b);
                if ((sweep & 7) == 7 || sweep < 2) LB_HIP(hipMemcpy(&left, b.pending, 4, hipMemcpyDeviceToHost));
            }
            LB_HIP(hipMemcpy(&left, b.pending, 4, hipMemcpyDeviceToHost));
            if (left != 0) { arena.release(); return RTS_ERR_BAD_BVH; }
        }
        hipLaunchKernelGGL(emitKernel, gridN, block, 0, nullptr, b, sortedOrder, (uint32_t*)d_packed);
    } else {
        hipLaunchKernelGGL(emitSingleKernel, dim3(1), dim3(64), 0, nullptr, b, (uint32_t*)d_packed);
    }
    LB_HIP(hipEventRecord(ev1, nullptr));
    LB_HIP(hipEventSynchronize(ev1));
    LB_HIP(hipGetLastError());
    float ms = 0;
    LB_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    if (build_ms) *build_ms = ms;
    if (out_packed) LB_HIP(hipMemcpy(out_packed, d_packed, count * 16, hipMemcpyDeviceToHost));
    if (install) {
        int s = rts_ctx_adopt_device_bvh(ctx, d_packed, count, P);   // the context owns d_packed from here on
        arena.release(s == RTS_OK ? (void*)d_packed : nullptr);
        return s;
    }
    arena.release();
    return RTS_OK;
}

extern "C" int rts_bvh_build_device(rts_ctx* ctx, const float* vertices, size_t vertex_floats, uint32_t stride,
                                    const uint32_t* indices, uint32_t P, rts_vec4u* out_packed, size_t out_cap,
                                    int install, float* build_ms) {
    return rts_bvh_build_device_ex(ctx, vertices, vertex_floats, stride, indices, P, RTS_GPU_BUILD_PLOC, 16, out_packed,
                                   out_cap, install, build_ms);
}
