// BVH producer of the shadow path (host, C++).  Output contract: SURVEY.md Appendix A, i.e. the
// byte layout produced by the reference's BVHBuilder::build (Source/BVHBuilder.cpp:248-368).
//
// This is NOT a transcription of the reference.  Design:
//   * leaves stay put in prim-id order; a `slot -> prim` permutation is what gets sorted.  Each of
//     the reference's four std::sort calls per range (BVHBuilder.cpp:92,149 / 162) becomes a
//     std::sort of 8-byte {key, prim} pairs -- the same comparison outcomes on the same sequence, so
//     libstdc++'s introsort produces the same tie order as sorting the reference's 72-byte nodes;
//   * an inner node is identified by its split position (`mid`), so node storage is pre-sized and
//     sub-ranges can be built by different threads with no allocation order to reproduce;
//   * no recursion anywhere (explicit work stacks): degenerate inputs give depth ~P trees;
//   * the "larger child first" swap (cpp:202-208) and the DFS numbering + miss links (cpp:222-244)
//     are separate linear passes over the finished topology.
#include "bvh_builder.h"
#include "../../include/rts.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace rts {
namespace {

struct KeyPrim { float key; u32 prim; };

struct Inner {
    float lo[3], hi[3];
    u32 child[2];   // ref: < P -> leaf slot, >= P -> inner (ref - P)
    u32 leaves;     // number of triangles below
};

struct Range { u32 begin, end, parent, side; };

struct Build {
    u32 P = 0;
    u32 sahLimit = 1000000;
    std::vector<float> lo, hi, ctr;   // per prim, xyz interleaved
    std::vector<u32> slotPrim;        // slot -> prim id (the array the reference physically sorts)
    std::vector<Inner> inner;         // P-1 entries, index = mid-1
    u32 root = 0;

    // shared work queue
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Range> queue;
    std::atomic<long> pending{0};
    std::atomic<bool> degenerate{false};   // some range had no split position (every SAH cost overflowed)
};

struct Scratch {
    std::vector<KeyPrim> keys;
    std::vector<float> saL, saR;
    void reserve(u32 n) {
        if (keys.size() < n) { keys.resize(n); saL.resize(n); saR.resize(n); }
    }
};

inline float surfaceArea(const float* lo, const float* hi) {   // BVHBuilder.cpp:24-28
    float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ey + ey * ez + ez * ex) * 2.0f;
}

struct Box {
    float lo[3], hi[3];
    void init() { for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = (p[k] < lo[k]) ? p[k] : lo[k];
            hi[k] = (hi[k] < p[k]) ? p[k] : hi[k];
        }
    }
};

// Sort slots [begin,end) by centre[axis]; identical comparison sequence to cpp:92-96.
inline void sortSlots(Build& b, Scratch& s, u32 begin, u32 end, int axis) {
    const u32 n = end - begin;
    KeyPrim* k = s.keys.data();
    for (u32 i = 0; i < n; ++i) {
        u32 p = b.slotPrim[begin + i];
        k[i].key = b.ctr[(size_t)p * 3 + axis];
        k[i].prim = p;
    }
    std::sort(k, k + n, [](const KeyPrim& x, const KeyPrim& y) { return x.key < y.key; });
    for (u32 i = 0; i < n; ++i) b.slotPrim[begin + i] = k[i].prim;
}

// Union of the leaf boxes in slot order with SSE min/max operand order (cpp:63-71):
// acc = acc < x ? acc : x  /  acc > x ? acc : x.
inline void rangeBounds(const Build& b, u32 begin, u32 end, float* lo, float* hi) {
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (u32 i = begin; i < end; ++i) {
        const float* l = &b.lo[(size_t)b.slotPrim[i] * 3];
        const float* h = &b.hi[(size_t)b.slotPrim[i] * 3];
        for (int k = 0; k < 3; ++k) {
            mn[k] = (mn[k] < l[k]) ? mn[k] : l[k];
            mx[k] = (mx[k] > h[k]) ? mx[k] : h[k];
        }
    }
    for (int k = 0; k < 3; ++k) { lo[k] = mn[k]; hi[k] = mx[k]; }
}

// Split position for slots [begin,end) (cpp:78-179); leaves the slots in the reference's order.
u32 chooseSplit(Build& b, Scratch& s, u32 begin, u32 end, const float* lo, const float* hi) {
    const u32 n = end - begin;
    s.reserve(n);
    if (n <= b.sahLimit) {
        u32 carried = begin;            // `bestSplit` lives outside the axis loop (cpp:81)
        u32 bestAxis = 0, best = begin;
        float bestCostAll = FLT_MAX;
        for (int axis = 0; axis < 3; ++axis) {
            sortSlots(b, s, begin, end, axis);
            Box fwd, bwd; fwd.init(); bwd.init();
            for (u32 i = 0; i < n; ++i) {                   // cpp:104-119
                u32 j = n - i - 1;
                size_t pf = (size_t)b.slotPrim[begin + i] * 3, pb = (size_t)b.slotPrim[begin + j] * 3;
                fwd.grow(&b.lo[pf]); fwd.grow(&b.hi[pf]);
                bwd.grow(&b.lo[pb]); bwd.grow(&b.hi[pb]);
                s.saL[i] = surfaceArea(fwd.lo, fwd.hi);
                s.saR[j] = surfaceArea(bwd.lo, bwd.hi);
            }
            float bestCost = FLT_MAX;
            for (u32 m = 1; m < n; ++m) {                   // cpp:121-139
                float cost = s.saL[m - 1] * (float)m + s.saR[m] * (float)(n - m);
                if (cost < bestCost) { carried = begin + m; bestCost = cost; }
            }
            if (bestCost < bestCostAll) { best = carried; bestCostAll = bestCost; bestAxis = (u32)axis; }
        }
        sortSlots(b, s, begin, end, (int)bestAxis);         // cpp:149-153
        return best;
    }
    // spatial median on the widest axis, first maximum wins (cpp:157-178)
    float ext[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    int major = 0;
    for (int k = 1; k < 3; ++k) if (ext[major] < ext[k]) major = k;
    sortSlots(b, s, begin, end, major);
    float splitPos = (lo[major] + hi[major]) * 0.5f;
    for (u32 m = begin + 1; m < end; ++m)
        if (b.ctr[(size_t)b.slotPrim[m] * 3 + major] >= splitPos) return m;
    return end - 1;
}

const u32 kShareThreshold = 4096; // ranges at least this big may be handed to another thread

void link(Build& b, const Range& r, u32 ref) {
    if (r.parent == BVHNode::InvalidMask) b.root = ref;
    else b.inner[r.parent].child[r.side] = ref;
}

void worker(Build* bp, bool shareWork) {
    Build& b = *bp;
    Scratch s;
    std::vector<Range> local;
    for (;;) {
        Range r;
        {
            std::unique_lock<std::mutex> lk(b.mu);
            b.cv.wait(lk, [&] { return !b.queue.empty() || b.pending.load() == 0; });
            if (b.queue.empty()) return;
            r = b.queue.front();
            b.queue.pop_front();
        }
        local.clear();
        local.push_back(r);
        while (!local.empty()) {
            Range cur = local.back();
            local.pop_back();
            if (cur.end - cur.begin == 1) { link(b, cur, cur.begin); continue; }   // cpp:185-188
            float lo[3], hi[3];
            rangeBounds(b, cur.begin, cur.end, lo, hi);                                // cpp:190
            u32 mid = chooseSplit(b, s, cur.begin, cur.end, lo, hi);                   // cpp:192
            // Extents around 1e19 and beyond overflow the surface areas to +inf: no cost is < FLT_MAX, the split stays
            // at `begin` and the reference recurses without bound on an empty range.  Here the
            // range is abandoned and build() reports RTS_ERR_DEGENERATE; nothing is written for it.
            if (mid <= cur.begin || mid >= cur.end) { b.degenerate.store(true); continue; }
            u32 id = mid - 1;
            Inner& in = b.inner[id];
            for (int k = 0; k < 3; ++k) { in.lo[k] = lo[k]; in.hi[k] = hi[k]; }
            in.leaves = cur.end - cur.begin;
            link(b, cur, b.P + id);
            Range kids[2] = { { cur.begin, mid, id, 0 }, { mid, cur.end, id, 1 } };
            for (int c = 1; c >= 0; --c) {
                if (shareWork && kids[c].end - kids[c].begin >= kShareThreshold && c == 1) {
                    b.pending.fetch_add(1);
                    { std::lock_guard<std::mutex> lk(b.mu); b.queue.push_back(kids[c]); }
                    b.cv.notify_one();
                } else {
                    local.push_back(kids[c]);
                }
            }
        }
        if (b.pending.fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> lk(b.mu);
            b.cv.notify_all();
        }
    }
}

inline u32 f2u(float f) { u32 u; memcpy(&u, &f, 4); return u; }

} // namespace

bool BVHBuilder::build(const float* vertices, u32 stride, const u32* indices, u32 primCount) {
    m_nodes.clear();
    m_packedNodes.clear();
    lastError = RTS_OK;
    if (!vertices || !indices || primCount == 0 || stride < 3 || primCount > 0x33333333u) {
        lastError = RTS_ERR_INVALID_ARG;
        return false;
    }
    const u32 P = primCount;
    const u32 N = 2 * P - 1;

    Build b;
    b.P = P;
    b.sahLimit = sahPrimLimit;
    b.lo.resize((size_t)P * 3); b.hi.resize((size_t)P * 3); b.ctr.resize((size_t)P * 3);
    b.slotPrim.resize(P);
    b.inner.resize(P - 1);

    // Leaf records (cpp:261-284).  Box3::expand on v0,v1,v2 in order; centre = (min+max)*0.5f.
    bool finite = true;
    for (u32 p = 0; p < P; ++p) {
        Box box; box.init();
        for (int c = 0; c < 3; ++c) {
            const float* v = vertices + (size_t)stride * indices[(size_t)p * 3 + c];
            finite = finite && std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]);
            box.grow(v);
        }
        for (int k = 0; k < 3; ++k) {
            b.lo[(size_t)p * 3 + k] = box.lo[k];
            b.hi[(size_t)p * 3 + k] = box.hi[k];
            b.ctr[(size_t)p * 3 + k] = (box.lo[k] + box.hi[k]) * 0.5f;
        }
        b.slotPrim[p] = p;
    }
    if (!finite) { lastError = RTS_ERR_NONFINITE; return false; }

    // Topology, top-down, shared between threads.
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if (P < 2 * kShareThreshold) nt = 1;
    b.pending.store(1);
    b.queue.push_back(Range{ 0, P, BVHNode::InvalidMask, 0 });
    if (nt == 1) {
        worker(&b, false);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(worker, &b, true);
        for (auto& t : pool) t.join();
    }
    if (b.degenerate.load()) { lastError = RTS_ERR_DEGENERATE; return false; }

    // Larger-surface-area child first (cpp:202-208): strict `>` on the right child.
    auto refBox = [&](u32 ref, const float*& lo, const float*& hi) {
        if (ref < P) { size_t p = (size_t)b.slotPrim[ref] * 3; lo = &b.lo[p]; hi = &b.hi[p]; }
        else { lo = b.inner[ref - P].lo; hi = b.inner[ref - P].hi; }
    };
    for (u32 i = 0; i + 1 < P; ++i) {
        Inner& in = b.inner[i];
        const float *llo, *lhi, *rlo, *rhi;
        refBox(in.child[0], llo, lhi);
        refBox(in.child[1], rlo, rhi);
        if (surfaceArea(rlo, rhi) > surfaceArea(llo, lhi)) std::swap(in.child[0], in.child[1]);
    }

    // Depth-first numbering with miss links (cpp:222-244, 290-306), straight into m_nodes.
    m_nodes.resize(N);
    {
        struct Visit { u32 ref, index, next; };
        std::vector<Visit> stack;
        stack.push_back(Visit{ b.root, 0, BVHNode::InvalidMask });
        while (!stack.empty()) {
            Visit v = stack.back();
            stack.pop_back();
            BVHNode& out = m_nodes[v.index];
            out.next = v.next;
            if (v.ref < P) {
                u32 prim = b.slotPrim[v.ref];
                for (int k = 0; k < 3; ++k) {
                    out.bboxMin[k] = b.lo[(size_t)prim * 3 + k];
                    out.bboxMax[k] = b.hi[(size_t)prim * 3 + k];
                }
                out.prim = prim;
            } else {
                const Inner& in = b.inner[v.ref - P];
                for (int k = 0; k < 3; ++k) { out.bboxMin[k] = in.lo[k]; out.bboxMax[k] = in.hi[k]; }
                out.prim = BVHNode::InvalidMask;
                u32 leftLeaves = in.child[0] < P ? 1u : b.inner[in.child[0] - P].leaves;
                u32 leftIndex = v.index + 1;
                u32 rightIndex = leftIndex + (2 * leftLeaves - 1);
                stack.push_back(Visit{ in.child[1], rightIndex, v.next });
                stack.push_back(Visit{ in.child[0], leftIndex, rightIndex });
            }
        }
    }

    // Packed stream (cpp:308-367): 2 vec4 per node in DFS order, then one vec4 (v0) per triangle.
    m_packedNodes.resize((size_t)2 * N + P);
    BVHPackedNode* out = m_packedNodes.data();
    for (u32 i = 0; i < N; ++i, out += 2) {
        const BVHNode& nd = m_nodes[i];
        if (nd.isLeaf()) {
            const float* v0 = vertices + (size_t)stride * indices[(size_t)nd.prim * 3 + 0];
            const float* v1 = vertices + (size_t)stride * indices[(size_t)nd.prim * 3 + 1];
            const float* v2 = vertices + (size_t)stride * indices[(size_t)nd.prim * 3 + 2];
            out[0] = BVHPackedNode{ f2u(v1[0] - v0[0]), f2u(v1[1] - v0[1]), f2u(v1[2] - v0[2]), nd.prim + 2 * N };
            out[1] = BVHPackedNode{ f2u(v2[0] - v0[0]), f2u(v2[1] - v0[1]), f2u(v2[2] - v0[2]), nd.next };
        } else {
            out[0] = BVHPackedNode{ f2u(nd.bboxMin[0]), f2u(nd.bboxMin[1]), f2u(nd.bboxMin[2]), BVHNode::InvalidMask };
            out[1] = BVHPackedNode{ f2u(nd.bboxMax[0]), f2u(nd.bboxMax[1]), f2u(nd.bboxMax[2]), nd.next };
        }
    }
    for (u32 p = 0; p < P; ++p, ++out) {
        const float* v0 = vertices + (size_t)stride * indices[(size_t)p * 3 + 0];
        *out = BVHPackedNode{ f2u(v0[0]), f2u(v0[1]), f2u(v0[2]), 0u };
    }
    return true;
}

} // namespace rts
