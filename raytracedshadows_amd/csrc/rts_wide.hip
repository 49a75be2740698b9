// GPU-private copy of the packed node stream for the WIDE packet kernel, and the device-side validation of a stream.
//
// The API-level blob stays SURVEY.md Appendix A (BVHBuilder.cpp:308-367) byte for byte; this file derives from it, on the
// device, at rts_ctx_set_bvh / rts_bvh_build_device time:
//
//   * wide nodes (128 B): an inner node n at even depth with the boxes of the nodes TWO levels below it (up to four
//     "slots": the grandchildren; where a child is a leaf, the leaf itself).  One dependent fetch then decides two levels
//     of the reference's walk (RayTracedShadows.comp:75-111).  Why the boxes in between may be skipped: a parent's box
//     is the union of its children's (BVHBuilder.cpp:53-76, min/max: exact) and the slab test (comp:61-73) is monotone in
//     the box when no NaN arises, so  hit(child box) => hit(parent box)  and a ray reaches a leaf iff it hits the box of the
//     leaf's PARENT.  A leaf slot therefore carries its parent's box; inner slots carry their own.  The property is
//     checked per stream (validateKernel, "enclosed"); streams without it keep the stackless kernels.
//   * triangle records (64 B, one s_load_dwordx16): v0, e0, e1 of a leaf in one place (the stream keeps v0 in the tail: a
//     second dependent fetch), the leaf's node index, and the box of the leaf's PARENT (what a triangle hit is confirmed
//     against: no further fetch); in the order of the leaves in the stream.
//       [0..2] v0  [3..5] e0  [6..8] e1  [9] leaf node  [10..12] parent bboxMin  [13..15] parent bboxMax
//   * the parent table: parent node of every node (the stackless walk of a dissolved wide packet confirms a triangle hit
//     against the box of the leaf's parent).
//
// Layout of a wide node (32 dwords, fetched with two s_load_dwordx16):
//   [6k .. 6k+5]  slot k: bboxMin.xyz, bboxMax.xyz        (empty slot: +FLT_MAX / -FLT_MAX: nothing hits it)
//   [24 + k]      slot k: byte offset of the child's wide node (low bit 0) | byte offset of the leaf's triangle
//                 record + 1 (low bit 1) | 0xFFFFFFFF (empty)
//   [28]          node index of n in the stream (where a dissolved packet's rays continue the reference's walk)
//   [28 + k]      k = 1..3: node index in the stream of slot k's node (END: empty slot).  Slot k's subtree is the index
//                 range [index of slot k, index of slot k + 1) -- what the split-tile pieces (rts_kernels.hip) filter on;
//                 the everyday loops load dwords 0..27 only
// Slots are in the stream's (depth-first) order, and so are the wide nodes themselves: a subtree is one contiguous range.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "rts_device.h"

namespace rts {

namespace {

constexpr uint32_t END = 0xFFFFFFFFu;

struct StreamView {
    const uint32_t* w;        // the stream as dwords (8 per node, then 4 per tail vec4)
    uint32_t P, N;
    __device__ uint32_t tag(uint32_t i) const { return w[(size_t)i * 8 + 3]; }
    __device__ uint32_t link(uint32_t i) const { return w[(size_t)i * 8 + 7]; }
    __device__ bool leaf(uint32_t i) const { return tag(i) != END; }
};

__device__ __forceinline__ bool finiteBits(uint32_t u) { return (u & 0x7F800000u) != 0x7F800000u; }

// One thread per node, then one per tail vec4.  bad[0] bit 0: structure (the rules of rts_bvh_validate), bit 1: a
// non-finite float, bit 2: an inner node with bboxMin > bboxMax, bit 3: not a pre-order binary tree with the reference's
// miss links (right child = link of the left child, link of the right child = link of the parent, a leaf's link = the node
// after it -- END only for the last node --, the root's link = END) or a box that does not enclose its inner children's
// boxes.  With bit 3 clear every node lies in the tree: by induction over link(i) - i the subtree of node i is exactly
// [i, link(i)), so the root's is [0, N) -- no orphan node exists whose parent entry nobody writes (parentKernel).
__global__ void validateKernel(StreamView s, uint32_t* bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t flags = 0;
    if (t < s.N) {
        const uint32_t i = (uint32_t)t;
        const uint32_t* a = s.w + (size_t)i * 8;
        for (int k = 0; k < 3; ++k) if (!finiteBits(a[k]) || !finiteBits(a[4 + k])) flags |= 2u;
        const uint32_t next = a[7];
        if (next != END && !(next > i && next < s.N)) flags |= 1u;
        if (i == 0 && next != END) flags |= 8u;
        if (a[3] != END && next != (i + 1 < s.N ? i + 1 : END)) flags |= 8u;
        if (a[3] == END) {
            if (i + 1 >= s.N) flags |= 1u;
            else {
                for (int k = 0; k < 3; ++k) if (!(__uint_as_float(a[k]) <= __uint_as_float(a[4 + k]))) flags |= 4u;
                const uint32_t left = i + 1, right = s.link(left);
                if (right == END || right <= left || right >= s.N || s.link(right) != next) flags |= 8u;
                else {
                    const uint32_t kids[2] = { left, right };
                    for (uint32_t c : kids) {
                        if (s.leaf(c)) continue;
                        const uint32_t* b = s.w + (size_t)c * 8;
                        for (int k = 0; k < 3; ++k)
                            if (!(__uint_as_float(a[k]) <= __uint_as_float(b[k])) || !(__uint_as_float(b[4 + k]) <= __uint_as_float(a[4 + k]))) flags |= 8u;
                    }
                }
            }
        } else if (a[3] < 2 * s.N || a[3] >= 2 * s.N + s.P) flags |= 1u;
    } else if (t < (uint64_t)s.N + s.P) {
        const uint32_t* v = s.w + (size_t)s.N * 8 + (size_t)(t - s.N) * 4;
        for (int k = 0; k < 3; ++k) if (!finiteBits(v[k])) flags |= 2u;
    }
    if (flags) atomicOr(bad, flags);
}

// ---- construction, all in the stream's own (depth-first) order ------------------------------------------------------------
// A ray's successive nodes lie close together in the stream (a subtree is a contiguous range); the private copy keeps that
// property: wide node i is the i-th inner node of EVEN depth in stream order, triangle record j belongs to the j-th leaf in
// stream order.  (A first version numbered the wide nodes level by level: every step of a walk then landed in a different
// part of a 60 MB array -- 1.9 us per lane-per-ray iteration against 0.9 for the stream itself.)

// parents[i] = the inner node whose child i is (END for the root): one thread per inner node.
__global__ void parentKernel(StreamView s, uint32_t* parents) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.N) return;
    if (i == 0) parents[0] = END;
    if (s.leaf(i)) return;
    const uint32_t left = i + 1;
    parents[left] = i;
    parents[s.link(left)] = i;
}

// One thread per node: depth by walking up (at most maxDepth levels: deeper trees get no wide copy).  counts[i] packs
// {1 if i is a wide root} << 32 | {1 if i is a leaf}; stats[0] = deepest level seen, stats[1] = 1 if a walk was cut off.
__global__ void classifyKernel(StreamView s, const uint32_t* parents, uint32_t maxDepth, uint64_t* counts, uint32_t* stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.N) return;
    uint32_t depth = 0, a = i;
    while (parents[a] != END && depth <= maxDepth) { a = parents[a]; ++depth; }
    if (depth > maxDepth) { stats[1] = 1u; counts[i] = 0; return; }
    const bool leaf = s.leaf(i);
    counts[i] = leaf ? 1ull : ((depth & 1u) ? 0ull : (1ull << 32));
    if (leaf) atomicMax(&stats[0], depth);
}

// Exclusive prefix sum over 64-bit items (two 32-bit counters side by side), three kernels: per-tile sums, the sums'
// own scan by one workgroup, per-tile scan with the tile's offset.  1024 items per tile.
constexpr uint32_t SCAN_TILE = 1024;
__device__ __forceinline__ uint64_t blockExclusive(uint64_t v, uint64_t* total, uint64_t* sh) {    // 256 threads
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint64_t x = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    const uint64_t incl = sh[t];
    if (total) *total = sh[255];
    __syncthreads();
    return incl - v;
}
__global__ void scanTileSums(const uint64_t* in, uint32_t n, uint64_t* tileSums) {
    __shared__ uint64_t sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint64_t v = 0;
    for (uint32_t k = 0; k < 4; ++k) if (base + k < n) v += in[base + k];
    uint64_t total;
    blockExclusive(v, &total, sh);
    if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ void scanOfSums(uint64_t* tileSums, uint32_t tiles, uint64_t* grandTotal) {                // one workgroup
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < tiles ? tileSums[i] : 0;
        uint64_t total;
        const uint64_t ex = blockExclusive(v, &total, sh);
        if (i < tiles) tileSums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grandTotal = carry;
}
__global__ void scanTiles(const uint64_t* in, uint32_t n, const uint64_t* tileSums, uint64_t* out) {
    __shared__ uint64_t sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint64_t x[4], v = 0;
    for (uint32_t k = 0; k < 4; ++k) { x[k] = base + k < n ? in[base + k] : 0; v += x[k]; }
    uint64_t run = blockExclusive(v, nullptr, sh) + tileSums[blockIdx.x];
    for (uint32_t k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += x[k]; }
}

// One thread per node.  A wide root writes its wide node (children's numbers from the scan); a leaf writes its triangle
// record {v0, e0, e1, leaf node, parent's box}.  rank[i] = {wide roots before i} << 32 | {leaves before i}.
__global__ void emitWideKernel(StreamView s, const uint32_t* parents, const uint64_t* counts, const uint64_t* rank,
                               uint32_t* wide, uint32_t* tris) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= s.N) return;
    const uint32_t* a = s.w + (size_t)n * 8;
    if (a[3] != END) {
        const uint32_t* v0 = s.w + (size_t)a[3] * 4;
        uint32_t* o = tris + (size_t)(uint32_t)rank[n] * 16;
        o[0] = v0[0]; o[1] = v0[1]; o[2] = v0[2];
        o[3] = a[0]; o[4] = a[1]; o[5] = a[2];
        o[6] = a[4]; o[7] = a[5]; o[8] = a[6];
        o[9] = n;
        if (parents[n] == END) return;                              // (P >= 2 and a validated stream: every leaf has a parent)
        const uint32_t* pb = s.w + (size_t)parents[n] * 8;
        o[10] = pb[0]; o[11] = pb[1]; o[12] = pb[2];
        o[13] = pb[4]; o[14] = pb[5]; o[15] = pb[6];
        return;
    }
    if (!(counts[n] >> 32)) return;                                  // an inner node of odd depth: folded into its parent
    if (n != 0 && parents[n] == END) return;                         // (cannot happen for a stream that passed validateKernel)
    uint32_t* o = wide + (size_t)(uint32_t)(rank[n] >> 32) * 32;
    uint32_t k = 0;
    auto put = [&](uint32_t boxNode, uint32_t ref, uint32_t node) {
        const uint32_t* b = s.w + (size_t)boxNode * 8;
        o[6 * k + 0] = b[0]; o[6 * k + 1] = b[1]; o[6 * k + 2] = b[2];
        o[6 * k + 3] = b[4]; o[6 * k + 4] = b[5]; o[6 * k + 5] = b[6];
        o[24 + k] = ref;
        if (k) o[28 + k] = node;                                     // (slot 0 starts at n + 1 or n + 2)
        ++k;
    };
    auto leafRef = [&](uint32_t c) { return (uint32_t)rank[c] * 64u + 1u; };
    const uint32_t left = n + 1, right = s.link(left);
    const uint32_t kids[2] = { left, right };
    for (uint32_t c : kids) {
        if (s.leaf(c)) { put(n, leafRef(c), c); continue; }
        const uint32_t gl = c + 1, gr = s.link(gl);
        const uint32_t grand[2] = { gl, gr };
        for (uint32_t g : grand) {
            if (s.leaf(g)) put(c, leafRef(g), g);
            else put(g, (uint32_t)(rank[g] >> 32) * 128u, g);
        }
    }
    for (; k < 4; ++k) {
        o[6 * k + 0] = o[6 * k + 1] = o[6 * k + 2] = __float_as_uint(FLT_MAX);
        o[6 * k + 3] = o[6 * k + 4] = o[6 * k + 5] = __float_as_uint(-FLT_MAX);
        o[24 + k] = END;
        o[28 + k] = END;
    }
    o[28] = n;
}

} // namespace

// flagsOut: bit 0 structure broken, 1 non-finite float, 2 unordered box, 3 not an enclosing pre-order tree (see validateKernel)
// (default stream; the copies are synchronous)
hipError_t validateStreamDevice(const void* d_packed, uint32_t P, uint32_t* d_word, uint32_t* flagsOut) {
    StreamView s{ (const uint32_t*)d_packed, P, 2 * P - 1 };
    hipError_t e = hipMemset(d_word, 0, 4);
    if (e != hipSuccess) return e;
    const uint64_t threads = (uint64_t)s.N + P;
    hipLaunchKernelGGL(validateKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, s, d_word);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpy(flagsOut, d_word, 4, hipMemcpyDeviceToHost);
}

// counts + ranks (8 bytes each per node), tile sums, statistics
size_t wideScratchBytes(uint32_t P) {
    const size_t N = 2 * (size_t)P - 1, tiles = (N + SCAN_TILE - 1) / SCAN_TILE;
    return N * 16 + tiles * 8 + 1024;
}

// Builds the private copy: d_wide (P * 128 bytes), d_tris (P * 64 bytes), d_parents ((2P - 1) * 4 bytes);
// d_scratch: wideScratchBytes(P).  Returns the number of wide nodes and of levels; levels == 0 means "not built" (a tree
// deeper than maxDepth; the caller keeps the stackless kernels).
hipError_t buildWideDevice(const void* d_packed, uint32_t P, void* d_wide, void* d_tris, void* d_parents, void* d_scratch,
                           uint32_t maxDepth, uint32_t* wideCount, uint32_t* levels) {
    StreamView s{ (const uint32_t*)d_packed, P, 2 * P - 1 };
    *wideCount = 0; *levels = 0;
    if (P < 2) return hipSuccess;
    const uint32_t N = s.N, tiles = (N + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t* counts = (uint64_t*)d_scratch;
    uint64_t* rank = counts + N;
    uint64_t* tileSums = rank + N;
    uint64_t* grand = tileSums + tiles;                         // [0] = totals of the scan
    uint32_t* stats = (uint32_t*)(grand + 1);                   // [0] deepest leaf, [1] cut off
    hipError_t e = hipMemset(grand, 0, 64);
    if (e != hipSuccess) return e;
    const dim3 block(256), gridN((N + 255) / 256);
    e = hipMemsetAsync(d_parents, 0xFF, (size_t)N * 4, nullptr);     // END everywhere: no stale entry of a previous stream survives
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(parentKernel, gridN, block, 0, nullptr, s, (uint32_t*)d_parents);
    hipLaunchKernelGGL(classifyKernel, gridN, block, 0, nullptr, s, (const uint32_t*)d_parents, maxDepth, counts, stats);
    hipLaunchKernelGGL(scanTileSums, dim3(tiles), block, 0, nullptr, counts, N, tileSums);
    hipLaunchKernelGGL(scanOfSums, dim3(1), block, 0, nullptr, tileSums, tiles, grand);
    hipLaunchKernelGGL(scanTiles, dim3(tiles), block, 0, nullptr, counts, N, tileSums, rank);
    uint64_t host[2] = { 0, 0 };                                 // totals, stats
    e = hipMemcpy(host, grand, 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    const uint32_t deepest = (uint32_t)host[1], cut = (uint32_t)(host[1] >> 32);
    if (cut) return hipSuccess;                                  // *levels stays 0
    hipLaunchKernelGGL(emitWideKernel, gridN, block, 0, nullptr, s, (const uint32_t*)d_parents, counts, rank, (uint32_t*)d_wide,
                       (uint32_t*)d_tris);
    *wideCount = (uint32_t)(host[0] >> 32);
    *levels = deepest / 2 + 1;
    return hipGetLastError();
}

} // namespace rts
