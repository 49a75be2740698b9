// Internal interface between the C-ABI layer (rts_api.cpp) and the HIP kernels (rts_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rts {

enum Variant {
    V_STRAIGHT = 0,    // loop as the shader spells it
    V_WHILEWHILE = 1,  // inner-node descent loop + batched leaf test
    V_POSTPONE = 2,    // leaves parked in registers, tested wave-wide
    V_PACKET = 3,      // wave (8x8 px) walks the union of its rays' paths; nodes via scalar loads
    V_PACKET2 = 4,     // same, 2 rays per lane (16x8 px per wave)
    V_PACKET4 = 5,     // same, 4 rays per lane (16x16 px per wave)
    V_PACKET_PF = 6,   // V_PACKET with the sequential successor node prefetched into a second SGPR set
    V_SHARE = 7,       // lane-per-ray with work sharing inside the wave (idle lanes take half of a busy ray's range)
    V_WIDE = 8,        // packet over the private WIDE nodes (rts_wide.hip): four grandchild boxes per dependent fetch,
                       // cheap conservative slab test, triangle hits confirmed by the exact test of the leaf's parent box
    V_WIDE_C = 9,      // V_WIDE with the loop compiled from C++ instead of hand-written (reference form of the same algorithm)
    V_COUNT,
    V_AUTO = -1        // packet for big launches, V_SHARE for small ones and for generic rays
};

// Kernel argument block (passed by value; lives in the kernarg segment, read with scalar loads).
struct TraceParams {
    // ---- what a tile wave needs before it can ask for its G-buffer texel: 64 bytes, one batch of scalar loads -------------
    const float4* positions;  // W x H RGBA32F, camera-relative (device)
    uint8_t* mask;            // W x H bytes (device)
    uint32_t W, H, rowBegin, rowEnd;
    uint32_t pieceRows;       // rows of the 2-D grid that hold the pieces of split tiles (they come first); 0 without a table
    uint32_t blocksX, blocksY;
    uint32_t rowOrder;        // dispatch order of the tile rows (speed only): 0 first to last, 1 last to first, 2 middle row outwards
    const uint32_t* skipMap;  // split tiles: bit (by * blocksX + bx) set: the tile's own wave has nothing to do (NULL: no split table)
    uint32_t bandShift;       // interleaved stripes: log2(bandRows / 8) when that is a power of two, else 0xFFFFFFFF
    uint32_t stripe;
    // ------------------------------------------------------------------------------------------------------------------
    uint32_t bandRows, nStripes;           // interleaved stripes (nStripes <= 1: plain rowBegin..rowEnd)
    const void* bvh;          // packed vec4 stream, SURVEY.md Appendix A (device)
    uint32_t bvhBytes;
    uint32_t bvhFinite;       // every float of the stream is finite -> FAST slab test is legal
    uint32_t bvhOrdered;      // ... and every inner node has bboxMin <= bboxMax -> ordered slab test is legal
    uint32_t nBlocks, gridBlocks, swizzle;
    uint32_t grid2d;            // 1: launched as a blocksX x blocksY grid in natural order (no swizzle, no order table)
    float cam[3];
    uint32_t lightType, nsamples;
    uint32_t softSplit;         // soft shadows: 4 waves per tile, samples dealt over them (option "soft_split")
    uint32_t pixelBase;         // index in the caller's frame of this dispatch's pixel 0 (host-pointer stripes travel as frames of their own)
    uint32_t lightTable;        // per-pixel jitter: entries of offsets[] a pixel starts in (0 = off); rts_light.table
    float light[3];
    // generic rays
    const void* rays;         // rts_ray[n] (device)
    uint8_t* out;
    uint64_t nrays;
    const uint32_t* tileOrder; // optional: block i works on tile tileOrder[i] (device array of nBlocks entries)
    uint64_t* waveStats;      // diagnostics (tools/wave_stats.py): 4 u64 per wave, or NULL
    uint64_t* waveRealtime;   // diagnostics: 4 u64 per wave {s_memrealtime at start, at end (100 MHz), clocks to first ray, XCC id}
    uint64_t* clockProbe;     // diagnostics that the EVERYDAY instantiation carries too: the first wave of every tile row
                              // stamps {shader clock, 100 MHz clock} at its start and end (4 u64 per row), or NULL
    uint32_t packetBudget;    // side-steps between two coherence checks of a packet (dissolve rule)
    uint32_t packetShare;     // dissolve when rays served per step < packetShare/16 of the rays alive
    const void* wide;         // private wide nodes (128 B each, root first) or NULL: V_WIDE falls back to V_PACKET
    const void* tris;         // private triangle records (48 B each, leaves in stream order)
    const uint32_t* parents;  // private: parent node of every node of the stream
    uint32_t primCount;
    uint32_t wideBytes;       // size of the private copy (wide nodes + triangle records, one allocation)
    uint32_t trisOffset;      // byte offset of the triangle records in it (tris == wide + trisOffset)
    uint32_t wideLane;        // dissolved wide packets continue lane per ray over the WIDE nodes (0: over the stream, stackless)
    // split tiles (rts_ctx_plan_splits): tiles measured to be long are walked by several one-wave workgroups ("pieces"), each
    // over one index range of the node stream; the pieces occupy the first pieceRows rows of the 2-D grid (dispatched first)
    const uint32_t* pieces;   // 8 dwords per piece: {bx | by << 16, first node, end node, state slot | pieces of the tile << 24,
                              //  byte offset of the wide node the piece starts at, 0, 0, 0}
    uint64_t* tileState;      // 2 u64 per split tile {lanes found occluded by any piece, pieces done}; zero between launches
    uint32_t* pieceLog;       // split planning only: per piece 1 + pieceLogCap dwords {count, node indices visited ...}
    uint64_t* pieceClock;     // diagnostics ("piece_stats"): per piece {100 MHz clock at its start, at its end}, or NULL
    uint32_t nPieces, pieceLogCap;
    uint32_t allInTable;      // every tile of the dispatch has a record: the grid is the records alone (no tile rows at all)
    uint32_t hasPieces;       // the table has split tiles (0: front tiles only -- the instantiation without the piece path)
    uint32_t frontStride;     // ... records per XCD in frontMap
    // One dword per record, what a record's wave reads first: bx | by << 16 of a FRONT tile, 0xFFFFFFFF for a piece (which then
    // reads its 8 dwords).  Record i lives at (i mod 8) * frontStride + i / 8: the dispatcher deals workgroups round-robin over
    // the 8 XCDs, so every XCD reads ONE contiguous run of dwords -- 32 records per 128-byte line of its own L2 -- instead of
    // every eighth 32-byte record of an array all eight L2s have to pull in (a dependent miss in front of the texel request:
    // a table in plain image order was 2-5 % slower than no table).  NULL (planning launches): every record is a piece.
    const uint32_t* frontMap;
    float offsets[64][4];
};

struct SplitCut {             // planning: one selected tile
    uint32_t tile;            // bx | by << 16 (tile coordinates of the dispatch)
    uint32_t pieces;          // S
};

const char* kernelName(int variant, bool mask);
void tileShape(int variant, int wavesPerBlock, uint32_t* blockW, uint32_t* blockH);   // pixels covered by one block
hipError_t launchShadowMask(int variant, int wavesPerBlock, const TraceParams& p, hipStream_t stream, uint32_t ldsPad = 0);
hipError_t launchTraceRays(int variant, const TraceParams& p, hipStream_t stream);
hipError_t launchReciprocalSelfTest(unsigned long long* d_counts, hipStream_t stream);      // 3 counters, zeroed by the caller
// planning: from the visit logs of `tiles` one-piece walks (p.pieceLog layout) to the piece table: tile t gets cuts[t].pieces
// records starting at record firstPiece[t], its index ranges cut at the quantiles of its log
hipError_t launchSplitQuantiles(const uint32_t* d_log, uint32_t logCap, const SplitCut* d_cuts, const uint32_t* d_firstPiece,
                                uint32_t tiles, uint32_t* d_pieces, const void* d_wide, hipStream_t stream);

} // namespace rts
