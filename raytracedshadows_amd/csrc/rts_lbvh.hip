// BVH build on the GPU: an alternative PRODUCER of the packed node stream of SURVEY.md
// Appendix A.
//   RTS_GPU_BUILD_SAH   the reference's own split rule for every node, all ranges of a level at once (default; below)
// Three topologies over the Morton order of the triangle centroids (63-bit codes, hipcub radix sort):
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
// The Morton-order TREES are not the reference's full-sweep SAH tree (bvh_builder.cpp on the host and RTS_GPU_BUILD_SAH
// here are); any valid stream is a drop-in for the consumer, and masks agree with the SAH stream's up to the slab test's
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
extern "C" void* rts_ctx_scratch(rts_ctx* ctx, size_t bytes);

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

// Leaf boxes + the box of the centroids (the Morton grid).  The six scene-box atomics hit six addresses: reduced over the
// wave first and skipped when the (possibly stale) current value already covers the wave -- one per triangle cost 1 ms for 1M.
__global__ void leafBoxesKernel(Lbvh b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < b.P;
    float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };
    bool finite = true;
    if (live) {
        for (int c = 0; c < 3; ++c) {
            const float* v = b.verts + (size_t)b.stride * b.indices[(size_t)p * 3 + c];
            for (int k = 0; k < 3; ++k) {
                const float x = v[k];
                finite = finite && (__builtin_fabsf(x) < __builtin_inff());
                lo[k] = (c == 0 || x < lo[k]) ? x : lo[k];
                hi[k] = (c == 0 || hi[k] < x) ? x : hi[k];
            }
        }
        if (!finite) b.flags[0] = 1;
    }
    for (int k = 0; k < 3; ++k) {
        if (live) { b.leafLo[(size_t)p * 3 + k] = lo[k]; b.leafHi[(size_t)p * 3 + k] = hi[k]; }
        const uint32_t c = encodeOrdered((lo[k] + hi[k]) * 0.5f);        // centroid as in BVHBuilder (Box3::center)
        uint32_t mn = live ? c : 0xFFFFFFFFu, mx = live ? c : 0u;
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t omn = (uint32_t)__shfl_xor((int)mn, d, 64), omx = (uint32_t)__shfl_xor((int)mx, d, 64);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if ((threadIdx.x & 63) == 0) {
            if (mn < __atomic_load_n(&b.sceneBox[k], __ATOMIC_RELAXED)) atomicMin(&b.sceneBox[k], mn);
            if (mx > __atomic_load_n(&b.sceneBox[3 + k], __ATOMIC_RELAXED)) atomicMax(&b.sceneBox[3 + k], mx);
        }
    }
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

// ---- full-sweep SAH on the GPU (RTS_GPU_BUILD_SAH) ---------------------------------------------------------------
// The reference's own split rule (BVHBuilder.cpp:78-179) for EVERY node, level by level: all ranges ("segments") of a
// level are split at once.  Three index lists hold the triangles of every segment sorted by the centroid on x, y and z
// (sorted once with a stable radix sort, kept sorted by stable partitions: ties are ordered by triangle id -- the one
// difference from the reference, whose std::sort leaves ties in an unspecified order; on meshes without equal
// centroids the stream equals BVHBuilder's byte for byte).  Per level:
//   1. six segmented scans (3 axes x forward / backward) of the triangle boxes  -> saL[i], saR[i]     (cpp:104-119)
//      and a seventh in the order the range was left in by its parent's split -> the node's box, with the reference's
//      min/max operand order (cpp:63-71: of a +0 and a -0 the later one stays)
//   2. cost(m) = saL[m-1] * m + saR[m] * (n - m), minimum over positions (first wins), then over axes (first wins):
//      one 64-bit atomicMin per segment on (cost bits, axis, m)                                        (cpp:121-146)
//      ranges above `limit` triangles: spatial median on the widest axis                             (cpp:157-178)
//   3. the new node (id = global position of the split - 1, unique), child links, "larger area first" (cpp:202-208)
//   4. stable partition of the two other lists (exclusive sums of the "goes left" flags); the boxes travel with their
//      list entries, one array per component, so every scan streams coalesced floats (gathering 32-byte boxes by
//      triangle id, or reading them as float4 pairs 128 B apart per lane, made the scans twice as slow)
constexpr int SCAN_ITEMS = 4, SCAN_TILE = 256 * SCAN_ITEMS, SCAN_TILE_SHIFT = 10;
static_assert(SCAN_TILE == 1 << SCAN_TILE_SHIFT, "tile size");

// Workgroups go round-robin over the 8 XCDs: give each XCD one contiguous eighth of the positions, so the second pass of
// a scan finds the tile its first pass read in the same XCD's L2.
__device__ __forceinline__ uint32_t xcdContiguousBlock(uint32_t blk, uint32_t nBlocks) {
    const uint32_t per = (nBlocks + 7) / 8, mapped = (blk & 7u) * per + (blk >> 3);
    return mapped;                                                       // may be >= nBlocks: such a block has nothing to do
}

struct Sah {
    uint32_t P, limit;
    const float* leafLo; const float* leafHi;
    uint32_t* ord[3]; uint32_t* ordOut[3];          // position -> triangle, per axis
    float* bc[4][6]; float* bcOut[3][6];            // the triangles' boxes IN LIST ORDER, one array per component (lo.xyz,
                                                    // hi.xyz): they move with the partition, so a thread's four consecutive
                                                    // positions are 16 contiguous bytes per component and a wave's loads are
                                                    // coalesced (list 3 = triangle order, the slot order of the root)
    uint32_t* segB; uint32_t* segE;                 // per position: its segment [begin, end)
    uint32_t* segBOut; uint32_t* segEOut;
    float* saL[3]; float* saR[3];                   // area of the boxes of [begin .. i] / [i .. end)
    float* segLo; float* segHi;                     // at [begin]: the segment's box
    unsigned long long* bestKey;                    // at [begin]: (cost bits << 32) | axis << 30 | m
    uint32_t* splitInfo;                            // at [begin]: axis << 30 | m of the split taken this level
    uint32_t* segParent;                            // at [begin]: parent node id * 2 + child slot, END for the root
    uint32_t* posAxis; uint32_t* posAxisOut;        // per position: axis its segment's parent was split on (3: the root)
    uint32_t* side;                                 // per triangle: 1 = goes left
    uint32_t* cnt[3];                               // exclusive sum of side over the list of each axis
    void* boxAggs; uint32_t* cntAggs;               // per-block aggregates of the scans
    uint32_t* flags;                                // [0] a segment with more than one triangle was created, [1] degenerate
    uint32_t* tileLive; uint32_t* tileLivePrev; uint32_t* tileLiveOut;   // per SCAN_TILE positions: holds a range of more than one
                                                    // triangle at this level / the previous one / the next one (finished tiles are skipped)
    uint32_t* child; uint32_t* parent; uint32_t* leaves; float* nodeLo; float* nodeHi;
};

struct BoxAgg { float lo0, lo1, lo2, hi0, hi1, hi2; uint32_t head; };   // head: the run contains the start of a segment

struct BoxOp {
    typedef BoxAgg Agg;
    // aggregate slices: axis * 2 + (0 forward | 1 backward); 6 = forward in the slot order of the reference (node boxes)
    __device__ static Agg identity() { const float inf = __builtin_inff(); return Agg{ inf, inf, inf, -inf, -inf, -inf, 0u }; }
    __device__ static Agg combine(const Agg& a, const Agg& b) {
        if (b.head) return b;
        return Agg{ a.lo0 < b.lo0 ? a.lo0 : b.lo0, a.lo1 < b.lo1 ? a.lo1 : b.lo1, a.lo2 < b.lo2 ? a.lo2 : b.lo2,      // cpp:63-71
                    a.hi0 > b.hi0 ? a.hi0 : b.hi0, a.hi1 > b.hi1 ? a.hi1 : b.hi1, a.hi2 > b.hi2 ? a.hi2 : b.hi2, a.head };
    }
    __device__ static Agg shflUp(const Agg& v, int d) {
        return Agg{ __shfl_up(v.lo0, d, 64), __shfl_up(v.lo1, d, 64), __shfl_up(v.lo2, d, 64),
                    __shfl_up(v.hi0, d, 64), __shfl_up(v.hi1, d, 64), __shfl_up(v.hi2, d, 64), (uint32_t)__shfl_up((int)v.head, d, 64) };
    }
    __device__ static Agg shflDown(const Agg& v, int d) {
        return Agg{ __shfl_down(v.lo0, d, 64), __shfl_down(v.lo1, d, 64), __shfl_down(v.lo2, d, 64),
                    __shfl_down(v.hi0, d, 64), __shfl_down(v.hi1, d, 64), __shfl_down(v.hi2, d, 64), (uint32_t)__shfl_down((int)v.head, d, 64) };
    }
    __device__ static Agg* aggs(const Sah& s) { return (Agg*)s.boxAggs; }
};

struct CountOp {
    typedef uint32_t Agg;
    __device__ static Agg identity() { return 0u; }
    __device__ static Agg combine(Agg a, Agg b) { return a + b; }
    __device__ static Agg shflUp(Agg v, int d) { return (uint32_t)__shfl_up((int)v, d, 64); }
    __device__ static Agg* aggs(const Sah& s) { return s.cntAggs; }
    __device__ static Agg load(const Sah& s, int y, uint32_t t) { return s.side[s.ord[y][t]]; }
    __device__ static void store(const Sah& s, int y, uint32_t t, Agg, Agg excl) { s.cnt[y][t] = excl; }
};

// Exclusive prefix of one value per thread over the block (in thread order; the operator need not commute) + the total.
template <class Op, int WAVES>
__device__ __forceinline__ void blockExclusive(const typename Op::Agg& mine, typename Op::Agg* excl, typename Op::Agg* total,
                                               typename Op::Agg* lds) {
    typedef typename Op::Agg A;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    A x = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const A o = Op::shflUp(x, d);
        if (lane >= d) x = Op::combine(o, x);
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    const A prev = Op::shflUp(x, 1);
    A pre = Op::identity(), tot = Op::identity();
    for (int w = 0; w < WAVES; ++w) {
        if (w == wave) pre = tot;
        tot = Op::combine(tot, lds[w]);
    }
    *excl = lane == 0 ? pre : Op::combine(pre, prev);
    *total = tot;
    __syncthreads();
}

// The same in the opposite thread order (the backward scans): thread t + 1 comes before thread t.
template <class Op, int WAVES>
__device__ __forceinline__ void blockExclusiveRev(const typename Op::Agg& mine, typename Op::Agg* excl, typename Op::Agg* total,
                                                  typename Op::Agg* lds) {
    typedef typename Op::Agg A;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    A x = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const A o = Op::shflDown(x, d);
        if (lane + d < 64) x = Op::combine(o, x);
    }
    if (lane == 0) lds[wave] = x;
    __syncthreads();
    const A prev = Op::shflDown(x, 1);
    A pre = Op::identity(), tot = Op::identity();
    for (int w = WAVES - 1; w >= 0; --w) {
        if (w == wave) pre = tot;
        tot = Op::combine(tot, lds[w]);
    }
    *excl = lane == 63 ? pre : Op::combine(pre, prev);
    *total = tot;
    __syncthreads();
}

// Box scans, fused: blockIdx.y = 0..2 runs the forward AND the backward scan of that axis' list from one load of the tile
// (slices 2z and 2z + 1 of the aggregates; the backward slice is indexed from the last tile down, so the same
// scanBlocksKernel serves it), blockIdx.y = 3 the forward scan in the reference's slot order (slice 6, node boxes).
struct BoxTile {
    float c[SCAN_ITEMS][6];
    uint32_t b[SCAN_ITEMS], e[SCAN_ITEMS];
};

__device__ __forceinline__ void loadBoxTile(const Sah& s, uint32_t z, uint32_t base, BoxTile* t) {
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {                               // clamped, no branch: every load is in flight at once
        const uint32_t i = base + j < s.P ? base + j : s.P - 1;
        const uint32_t axis = z < 3 ? z : s.posAxis[i];
        float* const* src = s.bc[axis];
#pragma unroll
        for (int k = 0; k < 6; ++k) t->c[j][k] = src[k][i];
        t->b[j] = s.segB[i]; t->e[j] = s.segE[i];
    }
}

__device__ __forceinline__ BoxAgg tileItem(const BoxTile& t, int j, bool head) {
    return BoxAgg{ t.c[j][0], t.c[j][1], t.c[j][2], t.c[j][3], t.c[j][4], t.c[j][5], head ? 1u : 0u };
}

__device__ __forceinline__ float aggArea(const BoxAgg& a) {
    const float lo[3] = { a.lo0, a.lo1, a.lo2 }, hi[3] = { a.hi0, a.hi1, a.hi2 };
    return surfaceArea(lo, hi);
}

__global__ __launch_bounds__(256) void boxReduceFusedKernel(Sah s, uint32_t nBlocks) {
    __shared__ BoxAgg lds[4];
    const uint32_t z = blockIdx.y;
    const uint32_t blk = xcdContiguousBlock(blockIdx.x, nBlocks);
    if (blk >= nBlocks) return;
    BoxAgg* aggs = (BoxAgg*)s.boxAggs;
    const size_t fIdx = (size_t)(z < 3 ? 2 * z : 6) * nBlocks + blk, bIdx = (size_t)(2 * z + 1) * nBlocks + (nBlocks - 1 - blk);
    if (!s.tileLive[blk]) {                                              // nothing but finished ranges: they neither take nor pass a carry
        if (threadIdx.x == 0) { aggs[fIdx] = BoxOp::identity(); if (z < 3) aggs[bIdx] = BoxOp::identity(); }
        return;
    }
    const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    BoxTile t;
    loadBoxTile(s, z, base, &t);
    BoxAgg acc = BoxOp::identity(), excl, tot;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < s.P) acc = BoxOp::combine(acc, tileItem(t, j, base + j == t.b[j]));
    blockExclusive<BoxOp, 4>(acc, &excl, &tot, lds);
    if (threadIdx.x == 0) aggs[fIdx] = tot;
    if (z == 3) return;
    acc = BoxOp::identity();
#pragma unroll
    for (int j = SCAN_ITEMS - 1; j >= 0; --j)
        if (base + j < s.P) acc = BoxOp::combine(acc, tileItem(t, j, base + j + 1 == t.e[j]));
    blockExclusiveRev<BoxOp, 4>(acc, &excl, &tot, lds);
    if (threadIdx.x == 0) aggs[bIdx] = tot;
}

__global__ __launch_bounds__(256) void boxApplyFusedKernel(Sah s, uint32_t nBlocks) {
    __shared__ BoxAgg lds[4];
    const uint32_t z = blockIdx.y;
    const uint32_t blk = xcdContiguousBlock(blockIdx.x, nBlocks);
    if (blk >= nBlocks) return;
    if (!s.tileLive[blk]) return;                                        // (its outputs are never read)
    const BoxAgg* aggs = (const BoxAgg*)s.boxAggs;
    const size_t fIdx = (size_t)(z < 3 ? 2 * z : 6) * nBlocks + blk, bIdx = (size_t)(2 * z + 1) * nBlocks + (nBlocks - 1 - blk);
    const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    BoxTile t;
    loadBoxTile(s, z, base, &t);
    BoxAgg incl[SCAN_ITEMS], acc = BoxOp::identity(), excl, tot;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < s.P) acc = BoxOp::combine(acc, tileItem(t, j, base + j == t.b[j]));
        incl[j] = acc;
    }
    blockExclusive<BoxOp, 4>(acc, &excl, &tot, lds);
    BoxAgg pre = BoxOp::combine(aggs[fIdx], excl);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const uint32_t i = base + j;
        if (i >= s.P) continue;
        const BoxAgg full = BoxOp::combine(pre, incl[j]);
        if (z < 3) {
            s.saL[z][i] = aggArea(full);
        } else if (i + 1 == t.e[j]) {                                    // the whole range: its box is the node's (cpp:190)
            const uint32_t b = t.b[j];
            s.segLo[(size_t)b * 3] = full.lo0; s.segLo[(size_t)b * 3 + 1] = full.lo1; s.segLo[(size_t)b * 3 + 2] = full.lo2;
            s.segHi[(size_t)b * 3] = full.hi0; s.segHi[(size_t)b * 3 + 1] = full.hi1; s.segHi[(size_t)b * 3 + 2] = full.hi2;
        }
    }
    if (z == 3) return;
    acc = BoxOp::identity();
#pragma unroll
    for (int j = SCAN_ITEMS - 1; j >= 0; --j) {
        if (base + j < s.P) acc = BoxOp::combine(acc, tileItem(t, j, base + j + 1 == t.e[j]));
        incl[j] = acc;
    }
    blockExclusiveRev<BoxOp, 4>(acc, &excl, &tot, lds);
    pre = BoxOp::combine(aggs[bIdx], excl);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < s.P) s.saR[z][base + j] = aggArea(BoxOp::combine(pre, incl[j]));
}

template <class Op>
__global__ __launch_bounds__(256) void scanReduceKernel(Sah s, uint32_t nBlocks) {
    typedef typename Op::Agg A;
    __shared__ A lds[4];
    const int y = blockIdx.y;
    const uint32_t blk = xcdContiguousBlock(blockIdx.x, nBlocks);
    if (blk >= nBlocks) return;
    if (!s.tileLive[blk]) {                                              // nothing but finished ranges: their sum is never read
        if (threadIdx.x == 0) Op::aggs(s)[(size_t)y * nBlocks + blk] = Op::identity();
        return;
    }
    const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    A item[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)                                 // no branch around a load: all of them are in flight together
        item[j] = Op::load(s, y, base + j < s.P ? base + j : s.P - 1);
    A acc = Op::identity();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < s.P) acc = Op::combine(acc, item[j]);
    A excl, tot;
    blockExclusive<Op, 4>(acc, &excl, &tot, lds);
    if (threadIdx.x == 0) Op::aggs(s)[(size_t)y * nBlocks + blk] = tot;
}

template <class Op>
__global__ __launch_bounds__(1024) void scanBlocksKernel(Sah s, uint32_t nBlocks) {     // aggregates -> what enters each block
    typedef typename Op::Agg A;
    __shared__ A lds[16];
    A* aggs = Op::aggs(s) + (size_t)blockIdx.x * nBlocks;
    A carry = Op::identity();
    for (uint32_t c = 0; c < nBlocks; c += 1024) {
        const uint32_t idx = c + threadIdx.x;
        const A v = idx < nBlocks ? aggs[idx] : Op::identity();
        A excl, tot;
        blockExclusive<Op, 16>(v, &excl, &tot, lds);
        if (idx < nBlocks) aggs[idx] = Op::combine(carry, excl);
        carry = Op::combine(carry, tot);
    }
}

template <class Op>
__global__ __launch_bounds__(256) void scanApplyKernel(Sah s, uint32_t nBlocks) {
    typedef typename Op::Agg A;
    __shared__ A lds[4];
    const int y = blockIdx.y;
    const uint32_t blk = xcdContiguousBlock(blockIdx.x, nBlocks);
    if (blk >= nBlocks) return;
    if (!s.tileLive[blk]) return;                                        // (its outputs are never read)
    const uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    A incl[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        incl[j] = Op::load(s, y, base + j < s.P ? base + j : s.P - 1);
    A acc = Op::identity();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < s.P) acc = Op::combine(acc, incl[j]);
        incl[j] = acc;
    }
    A excl, tot;
    blockExclusive<Op, 4>(acc, &excl, &tot, lds);
    const A pre = Op::combine(Op::aggs(s)[(size_t)y * nBlocks + blk], excl);
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < s.P) Op::store(s, y, base + j, Op::combine(pre, incl[j]), j ? Op::combine(pre, incl[j - 1]) : pre);
}

__device__ __forceinline__ int widestAxis(const float* lo, const float* hi) {      // first maximum wins (cpp:157-160)
    const float ext[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    int major = 0;
    for (int k = 1; k < 3; ++k) if (ext[major] < ext[k]) major = k;
    return major;
}

__global__ __launch_bounds__(256) void sahCostKernel(Sah s) {
    if (!s.tileLive[(blockIdx.x * 256u) >> SCAN_TILE_SHIFT]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long none = ~0ull;
    unsigned long long key = none;
    uint32_t b = END;
    if (i < s.P) {
        b = s.segB[i];
        const uint32_t e = s.segE[i], n = e - b;
        if (n > 1 && n <= s.limit) {
            if (i + 1 < e) {
                const uint32_t m = i - b + 1;                            // split after position i
                for (int a = 0; a < 3; ++a) {
                    const float cost = (s.saL[a][i] * (float)m + s.saR[a][i + 1] * (float)(n - m)) + 0.0f;     // cpp:121-123 (-0 -> +0: the key orders bits)
                    if (cost < 3.402823466e+38f) {                      // cpp:119,124: bestCost starts at FLT_MAX
                        const unsigned long long k = ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)a << 30) | m;
                        key = k < key ? k : key;
                    }
                }
            }
        } else if (n > 1) {                                              // spatial median (cpp:157-178)
            const int major = widestAxis(s.segLo + (size_t)b * 3, s.segHi + (size_t)b * 3);
            const float splitPos = (s.segLo[(size_t)b * 3 + major] + s.segHi[(size_t)b * 3 + major]) * 0.5f;
            if (i > b) {
                const uint32_t prim = s.ord[major][i];
                const float c = (s.leafLo[(size_t)prim * 3 + major] + s.leafHi[(size_t)prim * 3 + major]) * 0.5f;
                if (c >= splitPos) key = ((unsigned long long)major << 30) | (i - b);
            }
        }
    }
    // one atomic per wave when the wave lies inside one segment; a stale read of the current minimum only costs an atomic
    const uint32_t b0 = __builtin_amdgcn_readfirstlane(b);
    if (__all(b == b0)) {
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(key, d, 64);
            key = o < key ? o : key;
        }
        if ((threadIdx.x & 63) != 0) key = none;
    }
    if (key != none && key < __atomic_load_n(&s.bestKey[b], __ATOMIC_RELAXED)) atomicMin(&s.bestKey[b], key);
}

__global__ __launch_bounds__(256) void sahSplitKernel(Sah s) {
    if ((blockIdx.x & 3u) == 0 && threadIdx.x == 0) s.tileLiveOut[(blockIdx.x * 256u) >> SCAN_TILE_SHIFT] = 0;   // (the scatter of this level sets it)
    if (!s.tileLive[(blockIdx.x * 256u) >> SCAN_TILE_SHIFT]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.P) return;
    const uint32_t b = s.segB[i], e = s.segE[i], n = e - b;
    if (n <= 1) return;
    const unsigned long long key = s.bestKey[b];
    uint32_t axis, m;
    if (key != ~0ull) { axis = (uint32_t)(key >> 30) & 3u; m = (uint32_t)key & 0x3FFFFFFFu; }
    else if (n > s.limit) { axis = (uint32_t)widestAxis(s.segLo + (size_t)b * 3, s.segHi + (size_t)b * 3); m = n - 1; }   // cpp:178
    else { s.flags[1] = 1; axis = 0; m = 1; }                            // no finite cost: the reference would not terminate (E-4/E-5);
                                                                         // the host gives up at its next look at the flags, until then
                                                                         // this range is split like any other (every index stays valid)
    const uint32_t prim = s.ord[axis][i];
    s.side[prim] = (i - b) < m ? 1u : 0u;
    if (i != b) return;
    const uint32_t id = b + m - 1;                                       // one node per gap between two positions
    const uint32_t up = s.segParent[b];
    s.parent[id] = up == END ? END : up >> 1;
    if (up != END) s.child[up] = id;
    s.leaves[id] = n;
    for (int a = 0; a < 3; ++a) { s.nodeLo[(size_t)id * 3 + a] = s.segLo[(size_t)b * 3 + a]; s.nodeHi[(size_t)id * 3 + a] = s.segHi[(size_t)b * 3 + a]; }
    const uint32_t rightFirst = s.saR[axis][b + m] > s.saL[axis][b + m - 1] ? 1u : 0u;      // cpp:205-208
    const uint32_t slotL = 2 * id + rightFirst, slotR = 2 * id + (rightFirst ^ 1u);
    s.splitInfo[b] = (axis << 30) | m;
    if (m == 1) { const uint32_t leaf = s.P - 1 + s.ord[axis][b]; s.child[slotL] = leaf; s.parent[leaf] = id; }
    else { s.segParent[b] = slotL; s.flags[0] = 1; }
    if (n - m == 1) { const uint32_t leaf = s.P - 1 + s.ord[axis][e - 1]; s.child[slotR] = leaf; s.parent[leaf] = id; }
    else { s.segParent[b + m] = slotR; s.flags[0] = 1; }
}

__global__ __launch_bounds__(256) void sahScatterKernel(Sah s) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) s.flags[0] = 0;     // (the split of the next level sets it)
    const uint32_t tile = (blockIdx.x * 256u) >> SCAN_TILE_SHIFT;
    if (!s.tileLive[tile] && !s.tileLivePrev[tile]) return;             // finished for two levels: both buffers hold its final state
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t a = blockIdx.y;
    if (i >= s.P) return;
    const uint32_t b = s.segB[i], e = s.segE[i], n = e - b;
    const uint32_t prim = s.ord[a][i];
    float c[6];
    for (int k = 0; k < 6; ++k) c[k] = s.bc[a][k][i];
    if (n <= 1) {
        s.ordOut[a][i] = prim;
        for (int k = 0; k < 6; ++k) s.bcOut[a][k][i] = c[k];
        if (a == 0) { s.segBOut[i] = b; s.segEOut[i] = e; s.posAxisOut[i] = s.posAxis[i]; }
        return;
    }
    const uint32_t info = s.splitInfo[b], axis = info >> 30, m = info & 0x3FFFFFFFu;
    uint32_t to = i;
    if (a != axis) {                                                     // stable partition: lefts first, both in list order
        const uint32_t lefts = s.cnt[a][i] - s.cnt[a][b];
        to = s.side[prim] ? b + lefts : b + m + (i - b - lefts);
    }
    s.ordOut[a][to] = prim;
    for (int k = 0; k < 6; ++k) s.bcOut[a][k][to] = c[k];
    if (a == 0) {
        s.posAxisOut[i] = axis;
        if (i < b + m) { s.segBOut[i] = b; s.segEOut[i] = b + m; } else { s.segBOut[i] = b + m; s.segEOut[i] = e; }
        if ((i < b + m ? m : n - m) > 1) s.tileLiveOut[i >> SCAN_TILE_SHIFT] = 1;
        if (i == b) { s.bestKey[b] = ~0ull; s.bestKey[b + m] = ~0ull; }
    }
}

__global__ void sahSortKeysKernel(Sah s, int axis, uint32_t* keys, uint32_t* vals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= s.P) return;
    const float c = (s.leafLo[(size_t)p * 3 + axis] + s.leafHi[(size_t)p * 3 + axis]) * 0.5f;      // Box3::center
    keys[p] = encodeOrdered(c + 0.0f);                                   // -0 and +0 compare equal in the reference's sort
    vals[p] = p;
}

__global__ void sahListBoxesKernel(Sah s) {                              // boxes into list order, once
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t a = blockIdx.y;
    if (i >= s.P) return;
    const uint32_t prim = a < 3 ? s.ord[a][i] : i;
    for (int k = 0; k < 3; ++k) { s.bc[a][k][i] = s.leafLo[(size_t)prim * 3 + k]; s.bc[a][3 + k][i] = s.leafHi[(size_t)prim * 3 + k]; }
}

__global__ void sahInitKernel(Sah s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.P) return;
    s.segB[i] = 0; s.segE[i] = s.P;
    s.bestKey[i] = ~0ull;
    s.side[i] = 0;
    s.posAxis[i] = 3;
    if (i == 0) s.segParent[0] = END;
}

__global__ void setParentsKernel(Lbvh b, const uint32_t* pairs, uint32_t nPairs, uint32_t root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nPairs) b.parent[pairs[2 * i]] = pairs[2 * i + 1];
    if (i == 0) b.parent[root] = END;
}

// Pre-order index of a node = sum over the node and its ancestors of w = 1 + (size of the sibling subtree visited before
// it).  Computed by pointer doubling -- after k rounds acc[i] covers the 2^k nearest ancestors-or-self, up[i] is the 2^k-th
// ancestor -- so the cost is N log(depth) whatever the shape of the tree (a walk to the root per node was N * depth: tens of
// seconds for a chain of 10^5 equal boxes).
__global__ void preorderInitKernel(Lbvh b, uint32_t* acc, uint32_t* up) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t P = b.P, N = 2 * P - 1;
    if (id >= N) return;
    const uint32_t par = b.parent[id];
    uint32_t w = 0;
    if (par != END) {
        w = 1;
        if (b.child[2 * par + 1] == id) {                        // second child: the first one's subtree comes before
            const uint32_t first = b.child[2 * par];
            w += first >= P - 1 ? 1u : 2u * b.leaves[first] - 1u;
        }
    }
    acc[id] = w; up[id] = par;
}
__global__ void preorderJumpKernel(uint32_t N, const uint32_t* accIn, const uint32_t* upIn, uint32_t* accOut, uint32_t* upOut, uint32_t* unfinished) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t u = upIn[i];
    if (u == END) { accOut[i] = accIn[i]; upOut[i] = END; return; }
    accOut[i] = accIn[i] + accIn[u];
    const uint32_t uu = upIn[u];
    upOut[i] = uu;
    if (uu != END) *unfinished = 1u;
}

__global__ void emitKernel(Lbvh b, const uint32_t* order, const uint32_t* preorder, uint32_t* packed) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t P = b.P, N = 2 * P - 1;
    if (id >= N) return;
    const bool leaf = id >= P - 1;
    const uint32_t size = leaf ? 1u : 2u * b.leaves[id] - 1u;
    const uint32_t index = preorder[id];
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

__global__ void checkIndicesKernel(const uint32_t* idx, uint32_t n, uint32_t stride, size_t vertexFloats, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (size_t)idx[i] * stride + 3 > vertexFloats) flags[1] = 1;
}

// ordinal of the device a pointer's memory lives on, -1 for host memory (pageable, pinned or managed: those are copied)
int deviceOf(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return a.type == hipMemoryTypeDevice ? a.device : -1;
}

struct DeviceArena {            // working buffers: carved out of the context's scratch buffer; what does not fit (and what
    char* slab = nullptr;       // must outlive the build) is a hipMalloc of its own, freed whatever path leaves the function
    size_t slabBytes = 0, used = 0;
    std::vector<void*> own;     // (no fixed limit: without the context's slab every buffer of a build lands here)
    hipEvent_t ev[2] = { nullptr, nullptr };
    template <typename T> hipError_t getOwn(T** p, size_t bytes) {
        void* v = nullptr;
        hipError_t e = hipMalloc(&v, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        try { own.push_back(v); } catch (...) { (void)hipFree(v); return hipErrorOutOfMemory; }
        *p = (T*)v;
        return e;
    }
    template <typename T> hipError_t get(T** p, size_t bytes) {
        const size_t aligned = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
        if (slab && used + aligned <= slabBytes) { *p = (T*)(slab + used); used += aligned; return hipSuccess; }
        return getOwn(p, bytes);
    }
    void release(void* keep = nullptr) {
        for (void* v : own) if (v != keep) (void)hipFree(v);
        own.clear();
        for (hipEvent_t& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
};

#define LB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { arena.release(); return RTS_ERR_HIP + (int)e_; } } while (0)

// pre-order indices of all nodes into one of the two acc buffers (returned); six rounds cover 64 levels, then the device says
// whether any node still has ancestors to add
hipError_t preorderIndices(const Lbvh& b, uint32_t* acc[2], uint32_t* up[2], uint32_t* flag, const uint32_t** result) {
    const uint32_t N = 2 * b.P - 1;
    const dim3 block(256), grid((N + 255) / 256);
    hipLaunchKernelGGL(preorderInitKernel, grid, block, 0, nullptr, b, acc[0], up[0]);
    int cur = 0;
    for (uint32_t round = 0; round < 40; ++round) {
        const bool look = round >= 5;
        if (look) { hipError_t e = hipMemsetAsync(flag, 0, 4, nullptr); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(preorderJumpKernel, grid, block, 0, nullptr, N, acc[cur], up[cur], acc[cur ^ 1], up[cur ^ 1], flag);
        cur ^= 1;
        if (look) {
            uint32_t unfinished = 0;
            hipError_t e = hipMemcpy(&unfinished, flag, 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return e;
            if (!unfinished) break;
        }
    }
    *result = acc[cur];
    return hipGetLastError();
}

} // namespace

extern "C" int rts_bvh_build_device_ex(rts_ctx* ctx, const float* vertices, size_t vertex_floats, uint32_t stride,
                                       const uint32_t* indices, uint32_t P, int algorithm, uint32_t radius,
                                       rts_vec4u* out_packed, size_t out_cap, int install, float* build_ms) {
    if (!ctx || !vertices || !indices || P == 0 || stride < 3 || P > 0x0CCCCCCCu) return RTS_ERR_INVALID_ARG;
    if (algorithm != RTS_GPU_BUILD_LBVH && algorithm != RTS_GPU_BUILD_PLOC && algorithm != RTS_GPU_BUILD_PLOC_SAH && algorithm != RTS_GPU_BUILD_SAH)
        return RTS_ERR_INVALID_ARG;
    if (algorithm != RTS_GPU_BUILD_SAH && radius > 256) return RTS_ERR_INVALID_ARG;
    if (radius == 0) radius = algorithm == RTS_GPU_BUILD_SAH ? 1000000u : 16u;      // (SAH: BVHBuilder.cpp:83)
    const size_t count = (size_t)5 * P - 2;
    if (out_packed && out_cap < count) return RTS_ERR_CAPACITY;
    hipError_t e0 = hipSetDevice(rts_ctx_device_ordinal(ctx));
    if (e0 != hipSuccess) return RTS_ERR_HIP + (int)e0;
    // geometry that already lives on the context's device is used where it lies (no copy; its indices are checked by a kernel)
    const int vertsOn = deviceOf(vertices), idxOn = deviceOf(indices);
    if ((vertsOn >= 0 && vertsOn != rts_ctx_device_ordinal(ctx)) || (idxOn >= 0 && idxOn != rts_ctx_device_ordinal(ctx))) return RTS_ERR_INVALID_ARG;
    if (idxOn < 0)
        for (size_t i = 0; i < (size_t)3 * P; ++i)
            if ((size_t)indices[i] * stride + 3 > vertex_floats) return RTS_ERR_INVALID_ARG;

    DeviceArena arena;
    // every buffer of the largest path (SAH) + sort scratch; geometry that is used where it lies needs no room
    arena.slabBytes = (vertsOn >= 0 ? 0 : vertex_floats * 4) + (idxOn >= 0 ? 0 : (size_t)P * 12) + (size_t)P * 620 + ((size_t)4 << 20);
    arena.slab = getenv("RTS_NO_BUILDER_SLAB") ? nullptr : (char*)rts_ctx_scratch(ctx, arena.slabBytes);   // (test hook: the no-slab path)
    if (!arena.slab) arena.slabBytes = 0;
    Lbvh b{};
    b.P = P; b.stride = stride;
    float* d_verts; uint32_t* d_idx; uint32_t* d_packed;
    uint64_t* keysAlt; uint32_t* orderAlt;
    if (vertsOn >= 0) d_verts = const_cast<float*>(vertices); else LB_HIP(arena.get(&d_verts, vertex_floats * 4));
    if (idxOn >= 0) d_idx = const_cast<uint32_t*>(indices); else LB_HIP(arena.get(&d_idx, (size_t)P * 12));
    LB_HIP(arena.getOwn(&d_packed, count * 16 + 64));                   // (may become the context's BVH)
    LB_HIP(arena.get(&b.leafLo, (size_t)P * 12)); LB_HIP(arena.get(&b.leafHi, (size_t)P * 12));
    LB_HIP(arena.get(&b.sceneBox, 32));
    LB_HIP(arena.get(&b.keys, (size_t)P * 8)); LB_HIP(arena.get(&keysAlt, (size_t)P * 8));
    LB_HIP(arena.get(&b.order, (size_t)P * 4)); LB_HIP(arena.get(&orderAlt, (size_t)P * 4));
    LB_HIP(arena.get(&b.child, (size_t)P * 8)); LB_HIP(arena.get(&b.parent, (size_t)P * 8));
    LB_HIP(arena.get(&b.nodeLo, (size_t)P * 12)); LB_HIP(arena.get(&b.nodeHi, (size_t)P * 12));
    LB_HIP(arena.get(&b.leaves, (size_t)P * 4)); LB_HIP(arena.get(&b.done, (size_t)P * 4));
    LB_HIP(arena.get(&b.pending, 16)); LB_HIP(arena.get(&b.flags, 16));
    uint32_t* preAcc[2]; uint32_t* preUp[2];                     // pre-order numbering by pointer doubling (emit)
    for (int i = 0; i < 2; ++i) { LB_HIP(arena.get(&preAcc[i], (size_t)P * 8)); LB_HIP(arena.get(&preUp[i], (size_t)P * 8)); }
    b.verts = d_verts; b.indices = d_idx;

    if (vertsOn < 0) LB_HIP(hipMemcpy(d_verts, vertices, vertex_floats * 4, hipMemcpyHostToDevice));
    if (idxOn < 0) LB_HIP(hipMemcpy(d_idx, indices, (size_t)P * 12, hipMemcpyHostToDevice));
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
    if (idxOn >= 0) {
        hipLaunchKernelGGL(checkIndicesKernel, dim3((3 * P + 255) / 256), block, 0, nullptr, d_idx, 3 * P, stride, vertex_floats, b.flags);
        uint32_t bad = 0;
        LB_HIP(hipMemcpy(&bad, b.flags + 1, 4, hipMemcpyDeviceToHost));
        if (bad) { arena.release(); return RTS_ERR_INVALID_ARG; }
    }
    hipLaunchKernelGGL(leafBoxesKernel, gridP, block, 0, nullptr, b);
    uint32_t flag = 0;
    LB_HIP(hipMemcpy(&flag, b.flags, 4, hipMemcpyDeviceToHost));
    if (flag) { arena.release(); return RTS_ERR_NONFINITE; }

    const uint64_t* sortedKeys = b.keys;
    const uint32_t* sortedOrder = b.order;
    if (P > 1 && algorithm == RTS_GPU_BUILD_SAH) {
        Sah s{};
        s.P = P; s.limit = radius;
        s.leafLo = b.leafLo; s.leafHi = b.leafHi;
        s.child = b.child; s.parent = b.parent; s.leaves = b.leaves; s.nodeLo = b.nodeLo; s.nodeHi = b.nodeHi;
        const uint32_t nBlocks = (P + SCAN_TILE - 1) / SCAN_TILE;
        uint32_t *keys32, *keysAlt32, *identity;
        LB_HIP(arena.get(&keys32, (size_t)P * 4)); LB_HIP(arena.get(&keysAlt32, (size_t)P * 4)); LB_HIP(arena.get(&identity, (size_t)P * 4));
        for (int a = 0; a < 3; ++a) {
            LB_HIP(arena.get(&s.ord[a], (size_t)P * 4)); LB_HIP(arena.get(&s.ordOut[a], (size_t)P * 4));
            LB_HIP(arena.get(&s.saL[a], (size_t)P * 4)); LB_HIP(arena.get(&s.saR[a], (size_t)P * 4));
            LB_HIP(arena.get(&s.cnt[a], (size_t)P * 4));
        }
        float* comps;                                                   // 4 lists + 3 partition targets, 6 components each
        LB_HIP(arena.get(&comps, (size_t)P * 4 * 42));
        for (int a = 0; a < 4; ++a) for (int k = 0; k < 6; ++k) s.bc[a][k] = comps + (size_t)P * (a * 6 + k);
        for (int a = 0; a < 3; ++a) for (int k = 0; k < 6; ++k) s.bcOut[a][k] = comps + (size_t)P * (24 + a * 6 + k);
        LB_HIP(arena.get(&s.segB, (size_t)P * 4)); LB_HIP(arena.get(&s.segE, (size_t)P * 4));
        LB_HIP(arena.get(&s.segBOut, (size_t)P * 4)); LB_HIP(arena.get(&s.segEOut, (size_t)P * 4));
        LB_HIP(arena.get(&s.segLo, (size_t)P * 12)); LB_HIP(arena.get(&s.segHi, (size_t)P * 12));
        LB_HIP(arena.get(&s.bestKey, (size_t)P * 8));
        LB_HIP(arena.get(&s.splitInfo, (size_t)P * 4)); LB_HIP(arena.get(&s.segParent, (size_t)P * 4)); LB_HIP(arena.get(&s.posAxis, (size_t)P * 4)); LB_HIP(arena.get(&s.posAxisOut, (size_t)P * 4)); LB_HIP(arena.get(&s.side, (size_t)P * 4));
        LB_HIP(arena.get(&s.boxAggs, (size_t)nBlocks * 7 * sizeof(BoxAgg))); LB_HIP(arena.get(&s.cntAggs, (size_t)nBlocks * 3 * 4));
        LB_HIP(arena.get(&s.flags, 16));
        LB_HIP(arena.get(&s.tileLive, (size_t)nBlocks * 4)); LB_HIP(arena.get(&s.tileLivePrev, (size_t)nBlocks * 4));
        LB_HIP(arena.get(&s.tileLiveOut, (size_t)nBlocks * 4));
        LB_HIP(hipMemsetD32Async((hipDeviceptr_t)s.tileLive, 1, nBlocks, nullptr));
        LB_HIP(hipMemsetD32Async((hipDeviceptr_t)s.tileLivePrev, 1, nBlocks, nullptr));
        size_t tempBytes = 0;
        LB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keys32, keysAlt32, identity, s.ord[0], (int)P, 0, 32, nullptr));
        void* temp;
        LB_HIP(arena.get(&temp, tempBytes));
        for (int a = 0; a < 3; ++a) {                                   // stable: equal centroids stay in triangle order
            hipLaunchKernelGGL(sahSortKeysKernel, gridP, block, 0, nullptr, s, a, keys32, identity);
            LB_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keys32, keysAlt32, identity, s.ord[a], (int)P, 0, 32, nullptr));
        }
        hipLaunchKernelGGL(sahInitKernel, gridP, block, 0, nullptr, s);
        hipLaunchKernelGGL(sahListBoxesKernel, dim3(gridP.x, 4), block, 0, nullptr, s);
        const uint32_t scanGrid = (nBlocks + 7) / 8 * 8;               // (xcdContiguousBlock)
        const dim3 gridBox(scanGrid, 4), gridCnt(scanGrid, 3), gridScatter(gridP.x, 3);
        // The host looks at the flags (a round trip that drains the queue) only where the tree can end: not before level
        // ceil(log2 P) - 1, then every third level; a level past the end finds every tile finished and does nothing.
        uint32_t firstLook = 0;
        while ((1ull << (firstLook + 1)) < (unsigned long long)P) ++firstLook;
        LB_HIP(hipMemsetAsync(s.flags, 0, 16, nullptr));
        for (uint32_t level = 0;; ++level) {
            if (level > 262144u) { arena.release(); return RTS_ERR_DEGENERATE; }     // (a chain of equal boxes: a level each, ~0.1 ms)
            hipLaunchKernelGGL(boxReduceFusedKernel, gridBox, block, 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(scanBlocksKernel<BoxOp>, dim3(7), dim3(1024), 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(boxApplyFusedKernel, gridBox, block, 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(sahCostKernel, gridP, block, 0, nullptr, s);
            hipLaunchKernelGGL(sahSplitKernel, gridP, block, 0, nullptr, s);
            if (level >= firstLook && (level - firstLook) % 3 == 0) {
                uint32_t f[2] = { 0, 0 };
                LB_HIP(hipMemcpy(f, s.flags, 8, hipMemcpyDeviceToHost));
                if (f[1]) { arena.release(); return RTS_ERR_DEGENERATE; }
                if (!f[0]) break;                                       // every child made by this level is a leaf
            }
            hipLaunchKernelGGL(scanReduceKernel<CountOp>, gridCnt, block, 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(scanBlocksKernel<CountOp>, dim3(3), dim3(1024), 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(scanApplyKernel<CountOp>, gridCnt, block, 0, nullptr, s, nBlocks);
            hipLaunchKernelGGL(sahScatterKernel, gridScatter, block, 0, nullptr, s);
            for (int a = 0; a < 3; ++a) {
                std::swap(s.ord[a], s.ordOut[a]);
                for (int k = 0; k < 6; ++k) std::swap(s.bc[a][k], s.bcOut[a][k]);
            }
            std::swap(s.segB, s.segBOut); std::swap(s.segE, s.segEOut); std::swap(s.posAxis, s.posAxisOut);
            { uint32_t* t = s.tileLivePrev; s.tileLivePrev = s.tileLive; s.tileLive = s.tileLiveOut; s.tileLiveOut = t; }
        }
        const uint32_t* pre = nullptr;
        LB_HIP(preorderIndices(b, preAcc, preUp, b.flags + 3, &pre));
        hipLaunchKernelGGL(emitKernel, gridN, block, 0, nullptr, b, identity, pre, (uint32_t*)d_packed);
    } else if (P > 1) {
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
        const uint32_t* pre = nullptr;
        LB_HIP(preorderIndices(b, preAcc, preUp, b.flags + 3, &pre));
        hipLaunchKernelGGL(emitKernel, gridN, block, 0, nullptr, b, sortedOrder, pre, (uint32_t*)d_packed);
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
    return rts_bvh_build_device_ex(ctx, vertices, vertex_floats, stride, indices, P, RTS_GPU_BUILD_SAH, 0, out_packed,
                                   out_cap, install, build_ms);
}
