// Host-side BVH producer: same public surface as the reference's BVHBuilder
// (Source/BVHBuilder.h:8-32 -- BVHNode, BVHPackedNode, BVHBuilder::{m_nodes,m_packedNodes,build}),
// re-implemented as an iterative, index-sorting, multi-threaded builder whose output is
// byte-identical to a literal evaluation of Source/BVHBuilder.cpp:53-368.
#pragma once
#include <cstdint>
#include <vector>

namespace rts {

typedef uint32_t u32;

struct BVHNode {                       // BVHBuilder.h:8-20 (32 bytes)
    static const u32 InvalidMask = 0xFFFFFFFFu;
    float bboxMin[3];
    u32 prim = InvalidMask;
    float bboxMax[3];
    u32 next = InvalidMask;
    bool isLeaf() const { return prim != InvalidMask; }
};

struct BVHPackedNode { u32 a, b, c, d; }; // BVHBuilder.h:22-25 (one GLSL vec4)

struct BVHBuilder {
    std::vector<BVHNode> m_nodes;             // DFS order, 2P-1 entries
    std::vector<BVHPackedNode> m_packedNodes; // 2*(2P-1) + P entries

    // The two constants the reference hard-codes.
    u32 sahPrimLimit = 1000000;  // BVHBuilder.cpp:83
    int threads = 0;             // 0 = hardware concurrency (capped); never changes the tree

    // Same arguments as the reference (stride in FLOATS).  Unlike the reference, m_packedNodes is
    // cleared first (the reference appends on re-use) and bad input is rejected
    // instead of recursing forever (E-4): returns false and leaves both vectors empty when
    // primCount == 0 or a referenced vertex is not finite (lastError tells which).
    bool build(const float* vertices, u32 stride, const u32* indices, u32 primCount);

    int lastError = 0; // RTS_* status of the last build
};

} // namespace rts
