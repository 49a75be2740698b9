// C ABI of librts.so (declared in include/rts.h): thin, exception-free glue between plain
// pointers and (a) the host BVH producer, (b) the HIP traversal kernels.
#include "../../include/rts.h"
#include "bvh_builder.h"
#include "rts_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

using rts::TraceParams;

struct rts_ctx {
    int device = 0;
    void* d_bvh = nullptr;
    size_t bvhVec4 = 0;
    uint32_t P = 0;
    bool bvhFinite = false;
    bool bvhOrdered = false;
    int variant = rts::V_AUTO;
    int swizzle = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> marks;   // rts_timer_mark slots, created on first use
    // staging for the host-pointer entries
    void* d_in = nullptr; size_t inBytes = 0;
    void* d_out = nullptr; size_t outBytes = 0;
    const char* lastKernel = "";
    int packetBudget = 16;
    int packetShare = 4;
    int blockWaves = 1;
    int ldsPad = 0;              // experiment knob: dynamic LDS bytes per workgroup (throttles occupancy)
    uint32_t* d_tileOrder = nullptr; size_t tileOrderCount = 0;
    int useTileOrder = 1;                    // option "tile_order": 0 ignores an installed order
    uint32_t tileOrderSquare = 0, tileOrderBlock = 0;   // what rts_ctx_plan_tile_order planned with (0, 0: the caller's own order or none)
    bool tileOrderPlanned = false;           // installed by rts_ctx_plan_tile_order (the tuner may replace it), not by the caller
    bool tileOrderSoftOnly = false;          // planned on a dispatch of several samples: one-sample dispatches of the same size keep their everyday launch
    uint64_t* d_waveStats = nullptr; size_t waveStatsBytes = 0; size_t waveStatsUsed = 0;
    uint64_t launches = 0;
    int rowOrder = 0;                  // dispatch order of tile rows on 2-D grids: 0 top-down, 1 bottom-up, 2 middle-out
    void* d_scratch = nullptr; size_t scratchBytes = 0;    // working memory of the GPU builders, kept between builds
    // GPU-private copy for the wide packet kernel (rts_wide.hip): derived from d_bvh on the device, never visible outside
    void* d_wide = nullptr; size_t wideCap = 0;      // wide nodes (P * 128 B), triangle records (P * 64 B), parents (N * 4 B): one allocation
    void* d_tris = nullptr;
    void* d_parents = nullptr;
    uint32_t* d_word = nullptr;              // 4 bytes for the validation kernel's verdict
    uint32_t wideCount = 0, wideLevels = 0;
    bool bvhEnclosed = false;                // pre-order binary tree whose boxes enclose their children's (validateKernel)
    int wideCopy = 1;                        // option "wide_copy": build the private copy at upload
    int wideLane = 0;                        // option "wide_lane": dissolved wide packets walk the wide nodes lane per ray (LDS stacks: 28 waves per CU)
    int softSplit = 1;                       // option "soft_split": soft shadows with 4 waves per tile (samples side by side)
    uint32_t pixelBase = 0;                  // set around a host-pointer stripe (see rts_trace_shadow_mask)
    uint64_t* d_clockProbe = nullptr; size_t clockProbeRows = 0;    // option "clock_probe"
    // split table (rts_ctx_plan_splits): valid for ONE dispatch geometry (the key), used by every trace that matches it
    struct Splits {
        bool valid = false;
        uint32_t W = 0, H = 0, rowBegin = 0, rowEnd = 0, bandRows = 0, nStripes = 0, stripe = 0, blocksX = 0, blocksY = 0;
        uint32_t* d_skipMap = nullptr;       // one bit per tile of the dispatch
        uint32_t* d_pieces = nullptr;        // 8 dwords per piece
        uint32_t* d_frontMap = nullptr;      // one dword per record, XCD-major (TraceParams::frontMap)
        uint32_t frontStride = 0;
        uint32_t nPieces = 0, pieceRows = 0, nTiles = 0, nFront = 0;      // records (pieces + front tiles), split tiles, front tiles
        bool allTiles = false;
        rts_split_plan plan{};               // what the table was planned with (rts_ctx_get_split_plan)
        // {occluded lanes, pieces done} per split tile: one buffer per stream that traces with the table, so that frames in
        // flight on different streams never meet in it (frames on one stream follow each other)
        std::vector<std::pair<void*, uint64_t*>> state;
    } splits;
    int useSplits = 1;                       // option "tile_splits": 0 ignores an installed table
    int tuneForMotion = 0;                   // option "tune_for_motion": the tuner's tables must survive a camera path
    uint64_t* d_pieceClock = nullptr; size_t pieceClockCount = 0;    // option "piece_stats"
    struct Planning {                        // set by rts_ctx_plan_splits around its pieces-only launch
        const uint32_t* d_pieces; uint32_t nPieces, pieceRows; uint64_t* d_state; uint32_t* d_log; uint32_t logCap;
    };
    const Planning* planning = nullptr;
    uint32_t lastBlocksX = 0, lastBlocksY = 0; int lastVariant = 0; bool lastGrid2d = false;   // of the last mask dispatch
};

namespace {

inline int hipStatus(hipError_t e) { return e == hipSuccess ? RTS_OK : RTS_ERR_HIP + (int)e; }
}
namespace rts {   // rts_wide.hip
hipError_t validateStreamDevice(const void* d_packed, uint32_t P, uint32_t* d_word, uint32_t* flagsOut);
size_t wideScratchBytes(uint32_t P);
hipError_t buildWideDevice(const void* d_packed, uint32_t P, void* d_wide, void* d_tris, void* d_parents, void* d_scratch,
                           uint32_t maxDepth, uint32_t* wideCount, uint32_t* levels);
}
extern "C" void* rts_ctx_scratch(rts_ctx* c, size_t bytes);
namespace {
#define RTS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipStatus(e_); } while (0)

int ensure(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return RTS_OK;
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; *have = 0; if (e != hipSuccess) return hipStatus(e); }
    hipError_t e = hipMalloc(p, want ? want : 16);
    if (e != hipSuccess) { *p = nullptr; return hipStatus(e); }
    *have = want;
    return RTS_OK;
}

void clearSplits(rts_ctx* c) {
    rts_ctx::Splits& t = c->splits;
    if (t.d_skipMap) (void)hipFree(t.d_skipMap);
    if (t.d_pieces) (void)hipFree(t.d_pieces);
    if (t.d_frontMap) (void)hipFree(t.d_frontMap);
    for (auto& e : t.state) if (e.second) (void)hipFree(e.second);
    t = rts_ctx::Splits();
}

// the {occluded, done} words of the split tiles for traces on `stream` (zeroed once; every launch leaves them zero)
uint64_t* splitState(rts_ctx* c, void* stream) {
    rts_ctx::Splits& t = c->splits;
    for (auto& e : t.state) if (e.first == stream) return e.second;
    if (t.state.size() >= 8) return nullptr;                       // more streams than that trace without the table
    uint64_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)t.nTiles * 16 + 16) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // (cleared ON the stream that is about to use it: a memset on the default stream is not ordered before a launch on a
    //  non-blocking stream)
    if (hipMemsetAsync(d, 0, (size_t)t.nTiles * 16 + 16, (hipStream_t)stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
    try { t.state.emplace_back(stream, d); } catch (...) { (void)hipFree(d); return nullptr; }
    return d;
}

int fillParams(rts_ctx* ctx, TraceParams& p) {
    memset(&p, 0, sizeof(p));
    if (!ctx->d_bvh) return RTS_ERR_NO_BVH;
    p.bvh = ctx->d_bvh;
    p.bvhBytes = (uint32_t)(ctx->bvhVec4 * 16);
    p.bvhFinite = ctx->bvhFinite ? 1u : 0u;
    p.bvhOrdered = (ctx->bvhFinite && ctx->bvhOrdered) ? 1u : 0u;
    p.packetBudget = (uint32_t)ctx->packetBudget;
    p.packetShare = (uint32_t)ctx->packetShare;
    p.wide = ctx->wideCount ? ctx->d_wide : nullptr;
    p.tris = ctx->wideCount ? ctx->d_tris : nullptr;
    p.primCount = ctx->P;
    p.parents = ctx->wideCount ? (const uint32_t*)ctx->d_parents : nullptr;
    p.wideBytes = (uint32_t)((size_t)ctx->P * 192u);
    p.trisOffset = (uint32_t)((size_t)ctx->P * 128u);
    p.wideLane = (uint32_t)ctx->wideLane;
    p.softSplit = (uint32_t)ctx->softSplit;
    p.pixelBase = ctx->pixelBase;
    return RTS_OK;
}

// The stream in c->d_bvh has just been installed (uploaded or adopted): what the kernels may assume about it is decided
// on the device (one kernel over all nodes), then the private copy of the wide kernel is derived from it.  A stream that
// breaks the layout rules is refused (RTS_ERR_BAD_BVH) and the context is left without a BVH.
int finishInstall(rts_ctx* c, bool freeOnRefusal) {
    uint32_t flags = 0xF;
    hipError_t e = rts::validateStreamDevice(c->d_bvh, c->P, c->d_word, &flags);
    if (e != hipSuccess || (flags & 1u)) {
        if (freeOnRefusal) (void)hipFree(c->d_bvh);
        c->d_bvh = nullptr; c->bvhVec4 = 0; c->P = 0; c->wideCount = 0;
        return e != hipSuccess ? hipStatus(e) : RTS_ERR_BAD_BVH;
    }
    c->bvhFinite = !(flags & 2u);
    c->bvhOrdered = !(flags & 4u);
    c->bvhEnclosed = !(flags & 8u);
    c->wideCount = 0; c->wideLevels = 0;
    // (byte offsets inside the private copy are 32-bit: 192 bytes per triangle)
    if (!c->wideCopy || !c->bvhFinite || !c->bvhOrdered || !c->bvhEnclosed || c->P < 2 || c->P > (1u << 24)) return RTS_OK;
    int s = ensure(&c->d_wide, &c->wideCap, (size_t)c->P * 200 + 256);
    c->d_tris = s == RTS_OK ? (char*)c->d_wide + (size_t)c->P * 128 : nullptr;
    c->d_parents = s == RTS_OK ? (char*)c->d_wide + (size_t)c->P * 192 : nullptr;
    void* scratch = s == RTS_OK ? rts_ctx_scratch(c, rts::wideScratchBytes(c->P)) : nullptr;
    if (!scratch) return RTS_OK;                      // no memory for the copy: the stackless kernels need none
    // (trees deeper than 512 levels -- chains of single-triangle splits -- keep the stackless kernels)
    e = rts::buildWideDevice(c->d_bvh, c->P, c->d_wide, c->d_tris, c->d_parents, scratch, 512, &c->wideCount, &c->wideLevels);
    // a failure here only costs the private copy: the stream itself is installed and valid, the stackless kernels need
    // nothing else.  (Returning the error would leave the caller of the adopt path freeing a stream the context still holds.)
    if (e != hipSuccess) { c->wideCount = 0; c->wideLevels = 0; (void)hipGetLastError(); }
    return RTS_OK;
}

} // namespace

extern "C" {

const char* rts_status_string(int s) {
    switch (s) {
    case RTS_OK: return "ok";
    case RTS_ERR_INVALID_ARG: return "invalid argument";
    case RTS_ERR_CAPACITY: return "output capacity too small";
    case RTS_ERR_NONFINITE: return "non-finite vertex";
    case RTS_ERR_NO_BVH: return "no BVH set on context";
    case RTS_ERR_BAD_BVH: return "packed BVH failed validation";
    case RTS_ERR_DEGENERATE: return "degenerate input: no SAH split position (coordinate extents overflow the cost) or a chain of equal boxes deeper than 262144 levels";
    default: break;
    }
    if (s >= RTS_ERR_HIP) return hipGetErrorString((hipError_t)(s - RTS_ERR_HIP));
    return "unknown";
}

size_t rts_bvh_packed_count(uint32_t P) { return P ? (size_t)5 * P - 2 : 0; }
size_t rts_bvh_node_count(uint32_t P) { return P ? (size_t)2 * P - 1 : 0; }

int rts_bvh_build_ex(const float* vertices, uint32_t stride, const uint32_t* indices, uint32_t P,
                     uint32_t sah_prim_limit, int threads, rts_vec4u* out, size_t cap, rts_bvh_node* out_nodes) {
    if (!vertices || !indices || !out || P == 0) return RTS_ERR_INVALID_ARG;
    if (cap < rts_bvh_packed_count(P)) return RTS_ERR_CAPACITY;
    try {
        rts::BVHBuilder b;
        b.sahPrimLimit = sah_prim_limit;
        b.threads = threads;
        if (!b.build(vertices, stride, indices, P)) return b.lastError ? b.lastError : RTS_ERR_INVALID_ARG;
        memcpy(out, b.m_packedNodes.data(), b.m_packedNodes.size() * sizeof(rts_vec4u));
        if (out_nodes) memcpy(out_nodes, b.m_nodes.data(), b.m_nodes.size() * sizeof(rts_bvh_node));
    } catch (...) {
        return RTS_ERR_INVALID_ARG;
    }
    return RTS_OK;
}

int rts_bvh_build(const float* vertices, uint32_t stride, const uint32_t* indices, uint32_t P,
                  rts_vec4u* out, size_t cap, rts_bvh_node* out_nodes) {
    return rts_bvh_build_ex(vertices, stride, indices, P, 1000000u, 0, out, cap, out_nodes);
}

int rts_bvh_validate(const rts_vec4u* packed, size_t count, uint32_t* prim_count_out) {
    if (!packed || count < 3 || (count + 2) % 5 != 0) return packed ? RTS_ERR_BAD_BVH : RTS_ERR_INVALID_ARG;
    const uint64_t P = (count + 2) / 5, N = 2 * P - 1;
    if (P > 0x33333333ull) return RTS_ERR_BAD_BVH;
    for (uint64_t i = 0; i < N; ++i) {
        const rts_vec4u& a = packed[2 * i];
        const rts_vec4u& b = packed[2 * i + 1];
        if (b.d != 0xFFFFFFFFu && !(b.d > i && b.d < N)) return RTS_ERR_BAD_BVH;      // strictly forward
        if (a.d == 0xFFFFFFFFu) { if (i + 1 >= N) return RTS_ERR_BAD_BVH; }            // inner: i+1 exists
        else if (a.d < 2 * N || a.d >= 2 * N + P) return RTS_ERR_BAD_BVH;               // leaf: tail pointer
    }
    if (prim_count_out) *prim_count_out = (uint32_t)P;
    return RTS_OK;
}

int rts_device_count(int* count) {
    if (!count) return RTS_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return hipStatus(e);
}

int rts_ctx_create(int device, rts_ctx** out) {
    if (!out) return RTS_ERR_INVALID_ARG;
    *out = nullptr;
    RTS_HIP(hipSetDevice(device));
    rts_ctx* c = new (std::nothrow) rts_ctx();
    if (!c) return RTS_ERR_INVALID_ARG;
    c->device = device;
    hipError_t e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_word, 256);
    if (e != hipSuccess) { delete c; return hipStatus(e); }
    *out = c;
    return RTS_OK;
}

int rts_ctx_destroy(rts_ctx* c) {
    if (!c) return RTS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->d_bvh) (void)hipFree(c->d_bvh);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_waveStats) (void)hipFree(c->d_waveStats);
    if (c->d_tileOrder) (void)hipFree(c->d_tileOrder);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_wide) (void)hipFree(c->d_wide);
    if (c->d_word) (void)hipFree(c->d_word);
    if (c->d_clockProbe) (void)hipFree(c->d_clockProbe);
    if (c->d_pieceClock) (void)hipFree(c->d_pieceClock);
    clearSplits(c);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->marks) if (e) (void)hipEventDestroy(e);
    delete c;
    return RTS_OK;
}

int rts_ctx_set_bvh(rts_ctx* c, const rts_vec4u* packed, size_t count) {
    if (!c || !packed) return RTS_ERR_INVALID_ARG;
    uint32_t P = 0;
    int s = rts_bvh_validate(packed, count, &P);
    if (s != RTS_OK) return s;
    if (count * 16 >= 0xFFFFFF00ull) return RTS_ERR_BAD_BVH;           // 32-bit byte offsets on the device (top 256 B = "nothing")
    RTS_HIP(hipSetDevice(c->device));
    // the context only changes once the new copy is complete; a failure leaves it without a BVH, never with a torn one
    if (c->d_bvh) { void* old = c->d_bvh; c->d_bvh = nullptr; c->bvhVec4 = 0; c->P = 0; RTS_HIP(hipFree(old)); }
    void* d = nullptr;
    RTS_HIP(hipMalloc(&d, count * 16 + 64));          // + slack: the prefetching packet loop reads one node ahead
    hipError_t e = hipMemcpy(d, packed, count * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return hipStatus(e); }
    c->d_bvh = d; c->bvhVec4 = count; c->P = P;
    clearSplits(c);                      // (a split table holds node indices of the stream it was planned on)
    if (c->tileOrderPlanned) (void)rts_ctx_set_tile_order(c, nullptr, 0);   // (... a planned tile order the lives of its tiles)
    return finishInstall(c, true);       // finite / ordered / enclosed are decided on the device; private wide copy
}

int rts_ctx_set_option(rts_ctx* c, const char* key, int value) {
    if (!c || !key) return RTS_ERR_INVALID_ARG;
    if (!strcmp(key, "kernel")) { if (value < rts::V_AUTO || value >= rts::V_COUNT) return RTS_ERR_INVALID_ARG; c->variant = value; return RTS_OK; }
    if (!strcmp(key, "xcd_swizzle")) { c->swizzle = value ? 1 : 0; return RTS_OK; }
    // (1..4096: the dissolve rule multiplies budget * share * live rays in 32 bits)
    if (!strcmp(key, "packet_budget")) { if (value < 1 || value > 4096) return RTS_ERR_INVALID_ARG; c->packetBudget = value; return RTS_OK; }
    if (!strcmp(key, "block_waves")) { if (value != 1 && value != 4) return RTS_ERR_INVALID_ARG; c->blockWaves = value; return RTS_OK; }
    if (!strcmp(key, "row_order")) { if (value < 0 || value > 2) return RTS_ERR_INVALID_ARG; c->rowOrder = value; return RTS_OK; }
    if (!strcmp(key, "lds_pad")) { if (value < 0 || value > 65536) return RTS_ERR_INVALID_ARG; c->ldsPad = value; return RTS_OK; }
    if (!strcmp(key, "packet_share")) { if (value < 0 || value > 16) return RTS_ERR_INVALID_ARG; c->packetShare = value; return RTS_OK; }
    if (!strcmp(key, "wide_copy")) { c->wideCopy = value ? 1 : 0; return RTS_OK; }      // takes effect at the next upload / build
    if (!strcmp(key, "builder_scratch")) {          // 0: release the working memory the GPU builders keep between builds
        if (value != 0) return RTS_ERR_INVALID_ARG;
        RTS_HIP(hipSetDevice(c->device));
        if (c->d_scratch) { void* old = c->d_scratch; c->d_scratch = nullptr; c->scratchBytes = 0; RTS_HIP(hipFree(old)); }
        return RTS_OK;
    }
    if (!strcmp(key, "wide_lane")) { c->wideLane = value ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "soft_split")) { c->softSplit = value ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "tile_splits")) { c->useSplits = value ? 1 : 0; return RTS_OK; }     // 0: traces ignore an installed split table
    if (!strcmp(key, "tile_order")) { c->useTileOrder = value ? 1 : 0; return RTS_OK; }            // 0: traces ignore an installed tile order
    if (!strcmp(key, "tune_for_motion")) { c->tuneForMotion = value ? 1 : 0; return RTS_OK; }   // rts_ctx_autotune: only tables that keep over a camera path
    if (!strcmp(key, "piece_stats")) {          // diagnostics: value = pieces to stamp (0 = off), see rts_ctx_read_piece_stats
        RTS_HIP(hipSetDevice(c->device));
        if (c->d_pieceClock) { RTS_HIP(hipFree(c->d_pieceClock)); c->d_pieceClock = nullptr; c->pieceClockCount = 0; }
        if (value > 0) {
            RTS_HIP(hipMalloc((void**)&c->d_pieceClock, (size_t)value * 64));
            RTS_HIP(hipMemset(c->d_pieceClock, 0, (size_t)value * 64));
            c->pieceClockCount = (size_t)value;
        }
        return RTS_OK;
    }
    if (!strcmp(key, "clock_probe")) {          // value = tile rows to stamp (0 = off); packet kernels on 2-D grids
        RTS_HIP(hipSetDevice(c->device));
        if (c->d_clockProbe) { RTS_HIP(hipFree(c->d_clockProbe)); c->d_clockProbe = nullptr; c->clockProbeRows = 0; }
        if (value > 0) {
            RTS_HIP(hipMalloc((void**)&c->d_clockProbe, (size_t)value * 32));
            RTS_HIP(hipMemset(c->d_clockProbe, 0, (size_t)value * 32));
            c->clockProbeRows = (size_t)value;
        }
        return RTS_OK;
    }
    if (!strcmp(key, "wave_stats")) {            // diagnostics: value = number of waves to record (0 = off)
        RTS_HIP(hipSetDevice(c->device));
        if (c->d_waveStats) { RTS_HIP(hipFree(c->d_waveStats)); c->d_waveStats = nullptr; c->waveStatsBytes = 0; }
        if (value > 0) {
            c->waveStatsBytes = (size_t)value * 32;           // 4 u64 per wave, then 4 more per wave (realtime stamps)
            RTS_HIP(hipMalloc((void**)&c->d_waveStats, c->waveStatsBytes * 2));
            RTS_HIP(hipMemset(c->d_waveStats, 0, c->waveStatsBytes * 2));
        }
        return RTS_OK;
    }
    return RTS_ERR_INVALID_ARG;
}

int rts_ctx_get_option(rts_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return RTS_ERR_INVALID_ARG;
    if (!strcmp(key, "kernel")) { *value = c->variant; return RTS_OK; }
    if (!strcmp(key, "xcd_swizzle")) { *value = c->swizzle; return RTS_OK; }
    if (!strcmp(key, "packet_budget")) { *value = c->packetBudget; return RTS_OK; }
    if (!strcmp(key, "block_waves")) { *value = c->blockWaves; return RTS_OK; }
    if (!strcmp(key, "packet_share")) { *value = c->packetShare; return RTS_OK; }
    if (!strcmp(key, "kernel_count")) { *value = rts::V_COUNT; return RTS_OK; }
    if (!strcmp(key, "row_order")) { *value = c->rowOrder; return RTS_OK; }
    if (!strcmp(key, "bvh_finite")) { *value = c->bvhFinite ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "bvh_ordered")) { *value = c->bvhOrdered ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "bvh_enclosed")) { *value = c->bvhEnclosed ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "wide_copy")) { *value = c->wideCopy; return RTS_OK; }
    if (!strcmp(key, "builder_scratch")) { *value = (int)(c->scratchBytes >> 20); return RTS_OK; }     // MiB held
    if (!strcmp(key, "wide_lane")) { *value = c->wideLane; return RTS_OK; }
    if (!strcmp(key, "soft_split")) { *value = c->softSplit; return RTS_OK; }
    if (!strcmp(key, "wide_nodes")) { *value = (int)c->wideCount; return RTS_OK; }
    if (!strcmp(key, "tile_splits")) { *value = c->useSplits; return RTS_OK; }
    if (!strcmp(key, "tune_for_motion")) { *value = c->tuneForMotion; return RTS_OK; }
    if (!strcmp(key, "tile_order")) { *value = c->useTileOrder; return RTS_OK; }
    if (!strcmp(key, "tile_order_tiles")) { *value = (int)c->tileOrderCount; return RTS_OK; }
    if (!strcmp(key, "tile_order_planned")) { *value = c->tileOrderPlanned ? 1 : 0; return RTS_OK; }
    if (!strcmp(key, "tile_order_square")) { *value = (int)c->tileOrderSquare; return RTS_OK; }
    if (!strcmp(key, "tile_order_block")) { *value = (int)c->tileOrderBlock; return RTS_OK; }
    if (!strcmp(key, "split_tiles")) { *value = c->splits.valid ? (int)c->splits.nTiles : 0; return RTS_OK; }
    if (!strcmp(key, "front_tiles")) { *value = c->splits.valid ? (int)c->splits.nFront : 0; return RTS_OK; }
    if (!strcmp(key, "split_pieces")) { *value = c->splits.valid ? (int)c->splits.nPieces : 0; return RTS_OK; }
    if (!strcmp(key, "wide_levels")) { *value = (int)c->wideLevels; return RTS_OK; }
    return RTS_ERR_INVALID_ARG;
}

// Diagnostics (RTS_TUNE_LOG set): why the planner refused.
static int planRefused(const char* why) {
    if (getenv("RTS_TUNE_LOG")) fprintf(stderr, "[rts plan] refused: %s\n", why);
    return RTS_ERR_INVALID_ARG;
}

static int traceMaskImpl(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions,
                         uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, uint32_t band_rows,
                         uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, void* stream) {
    if (!c || !k || !d_positions || !d_mask || W == 0 || H == 0 || row_begin > row_end || row_end > H)
        return RTS_ERR_INVALID_ARG;
    if (light && (light->type > RTS_LIGHT_POINT || light->nsamples > 64)) return RTS_ERR_INVALID_ARG;
    if (light && light->table && (light->table > 64 || light->table < light->nsamples || light->nsamples < 2)) return RTS_ERR_INVALID_ARG;
    if ((uint64_t)W * H > (1ull << 31)) return RTS_ERR_INVALID_ARG;      // tile counts are 32-bit on the device
    TraceParams p;
    int s = fillParams(c, p);
    if (s != RTS_OK) return s;
    if (row_begin == row_end) return RTS_OK;
    RTS_HIP(hipSetDevice(c->device));
    p.positions = (const float4*)d_positions;
    p.mask = d_mask;
    p.W = W; p.H = H; p.rowBegin = row_begin; p.rowEnd = row_end;
    p.bandRows = band_rows; p.nStripes = n_stripes; p.stripe = stripe;
    p.bandShift = 0xFFFFFFFFu;
    if (n_stripes > 1 && band_rows % 8 == 0) {
        const uint32_t tiles = band_rows / 8;
        if ((tiles & (tiles - 1)) == 0) { uint32_t sh = 0; while ((1u << sh) < tiles) ++sh; p.bandShift = sh; }
    }
    // V_AUTO: a packet's steps are a dependent chain, so it needs several waves per SIMD to overlap them;
    // a launch with fewer than ~4 waves per SIMD is faster lane-per-ray (with in-wave work sharing: measured
    // 5-20 % ahead of the plain loop on every small frame).  (Bigger packets, V_PACKET2/4, were
    // measured slower or equal on every BASELINE config and are kept as selectable variants only.)
    int variant = c->variant;
    uint32_t rows = row_end - row_begin;
    if (n_stripes > 1) {                        // virtual rows = whole owned bands (partial last band guarded in-kernel)
        const uint32_t bands = (H + band_rows - 1) / band_rows;
        rows = ((bands - stripe + n_stripes - 1) / n_stripes) * band_rows;
    }
    const uint64_t pixels = (uint64_t)W * rows;
    // Big one-sample dispatches of a stream with a private copy take the WIDE packet (a static rule from what rts_ctx_autotune
    // picks on the 4K frames: city 0.153 against 0.164 ms, courtyard 0.562 against 0.630; at 1080p and for soft shadows the
    // stackless packet is ahead or level -- profiles/r04/tuning_robustness.log): a caller that never tunes gets the kernel the
    // tuner would have picked there, not its launch options.
    if (variant == rts::V_AUTO)
        variant = pixels < (1u << 18) ? rts::V_SHARE
                : (c->wideCount && (!light || light->nsamples <= 1) && pixels >= (1u << 22) && c->blockWaves == 1) ? rts::V_WIDE : rts::V_PACKET;
    if ((variant == rts::V_WIDE || variant == rts::V_WIDE_C) && !p.wide) variant = rts::V_PACKET;     // no private copy for this stream (see finishInstall)
    uint32_t bw, bh;
    rts::tileShape(variant, c->blockWaves, &bw, &bh);
    if (n_stripes > 1 && band_rows % bh != 0) return RTS_ERR_INVALID_ARG;   // a band is a whole number of workgroup rows (8, 16 or 32 pixel rows)
    p.blocksX = (W + bw - 1) / bw;
    p.blocksY = (rows + bh - 1) / bh;
    p.nBlocks = p.blocksX * p.blocksY;
    p.swizzle = c->swizzle ? 1u : 0u;
    p.gridBlocks = p.swizzle ? ((p.nBlocks + 7) / 8) * 8 : p.nBlocks;
    // (soft shadows in the one-tile packet forms run 4 waves per workgroup: "soft_split")
    const bool split = light && light->nsamples > 1 && c->softSplit && c->blockWaves == 1 &&
                       (variant == rts::V_PACKET || variant == rts::V_WIDE);
    const size_t statWaves = split ? 4 : (size_t)c->blockWaves;
    if (c->d_waveStats && (size_t)p.gridBlocks * statWaves * 32 <= c->waveStatsBytes) {
        p.waveStats = c->d_waveStats;
        p.waveRealtime = c->d_waveStats + c->waveStatsBytes / 8;
    }
    if (c->d_tileOrder && c->useTileOrder && c->tileOrderCount == p.nBlocks && !p.swizzle &&
        (!c->tileOrderSoftOnly || (light && light->nsamples > 1)))
        p.tileOrder = c->d_tileOrder;
    p.grid2d = (!p.swizzle && !p.tileOrder && p.blocksY <= 65535u) ? 1u : 0u;
    if (c->d_clockProbe && p.grid2d && p.blocksY <= c->clockProbeRows) p.clockProbe = c->d_clockProbe;
    p.rowOrder = (p.grid2d && n_stripes <= 1) ? (uint32_t)c->rowOrder : 0u;
    for (int i = 0; i < 3; ++i) p.cam[i] = k->cameraPosition[i];
    if (light) {
        p.lightType = light->type;
        p.nsamples = light->nsamples > 1 ? light->nsamples : 1;
        for (int i = 0; i < 3; ++i) p.light[i] = light->xyz[i];
        p.lightTable = p.nsamples > 1 ? light->table : 0u;
        if (p.nsamples > 1) memcpy(p.offsets, light->offsets, sizeof(float) * 4 * (p.lightTable ? p.lightTable : p.nsamples));
    } else {
        p.lightType = RTS_LIGHT_DIRECTIONAL;
        p.nsamples = 1;
        for (int i = 0; i < 3; ++i) p.light[i] = k->lightDirection[i];
    }
    c->lastBlocksX = p.blocksX; c->lastBlocksY = p.blocksY; c->lastVariant = variant; c->lastGrid2d = p.grid2d != 0;
    if (c->planning) {                       // pieces only: the planning walk of the selected tiles, with visit logs
        if (!(variant == rts::V_PACKET || variant == rts::V_WIDE) || c->blockWaves != 1 || p.nsamples != 1 || !p.grid2d || !p.wide ||
            (n_stripes > 1 && p.bandShift == 0xFFFFFFFFu))
            return planRefused("the planning walk needs a one-tile packet kernel, one sample, a 2-D grid, the private copy");
        p.waveStats = nullptr; p.waveRealtime = nullptr; p.clockProbe = nullptr; p.rowOrder = 0;
        p.skipMap = c->planning->d_pieces;   // (never read: no tile rows in this launch)
        p.pieces = c->planning->d_pieces; p.nPieces = c->planning->nPieces; p.pieceRows = c->planning->pieceRows;
        p.frontMap = nullptr; p.frontStride = 0; p.hasPieces = 1u;               // (every record is a piece)
        p.tileState = c->planning->d_state; p.pieceLog = c->planning->d_log; p.pieceLogCap = c->planning->logCap;
        p.blocksY = 0;
        return hipStatus(rts::launchShadowMask(variant, c->blockWaves, p, (hipStream_t)stream, 0));
    }
    // split table: only the everyday one-tile packet launches of the very dispatch it was planned for
    const rts_ctx::Splits& sp = c->splits;
    if (sp.valid && c->useSplits && sp.nPieces && (variant == rts::V_PACKET || variant == rts::V_WIDE) && c->blockWaves == 1 &&
        p.nsamples == 1 && p.grid2d && !p.waveStats && !c->wideLane && p.wide &&
        (n_stripes <= 1 || (p.bandShift != 0xFFFFFFFFu && p.rowOrder == 0)) &&
        sp.W == W && sp.H == H && sp.rowBegin == row_begin && sp.rowEnd == row_end && sp.bandRows == band_rows &&
        sp.nStripes == n_stripes && sp.stripe == stripe && sp.blocksX == p.blocksX && sp.blocksY == p.blocksY &&
        p.blocksY + sp.pieceRows <= 65535u) {
        if (uint64_t* st = splitState(c, stream)) {
            p.skipMap = sp.d_skipMap; p.pieces = sp.d_pieces; p.nPieces = sp.nPieces; p.pieceRows = sp.pieceRows;
            p.frontMap = sp.d_frontMap; p.frontStride = sp.frontStride;
            p.tileState = st;
            p.allInTable = sp.allTiles ? 1u : 0u;
            p.hasPieces = sp.nTiles ? 1u : 0u;
            if (c->d_pieceClock && c->pieceClockCount >= sp.nPieces) p.pieceClock = c->d_pieceClock;
        }
    }
    c->lastKernel = rts::kernelName(variant, true);
    ++c->launches;
    return hipStatus(rts::launchShadowMask(variant, c->blockWaves, p, (hipStream_t)stream, (uint32_t)c->ldsPad));
}

int rts_trace_shadow_mask_device(rts_ctx* c, const rts_constants* k, const rts_light* light,
                                 const float* d_positions, uint32_t W, uint32_t H,
                                 uint32_t row_begin, uint32_t row_end, uint8_t* d_mask, void* stream) {
    return traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, 0, 1, 0, d_mask, stream);
}

int rts_trace_shadow_mask_stripes_device(rts_ctx* c, const rts_constants* k, const rts_light* light,
                                         const float* d_positions, uint32_t W, uint32_t H, uint32_t band_rows,
                                         uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, void* stream) {
    if (band_rows == 0 || band_rows % 8 != 0 || n_stripes == 0 || stripe >= n_stripes) return RTS_ERR_INVALID_ARG;
    if (n_stripes == 1) return traceMaskImpl(c, k, light, d_positions, W, H, 0, H, 0, 1, 0, d_mask, stream);
    // rows this stripe owns: whole bands stripe, stripe+n, ... (the last one may be cut by H)
    const uint32_t bands = (H + band_rows - 1) / band_rows;
    uint32_t owned = 0;
    for (uint32_t b = stripe; b < bands; b += n_stripes) owned += (b + 1) * band_rows <= H ? band_rows : H - b * band_rows;
    if (owned == 0) return RTS_OK;
    // dispatch `owned` virtual rows; the kernel maps them onto the frame (ownedRow) and guards with row < H
    return traceMaskImpl(c, k, light, d_positions, W, H, 0, H, band_rows, n_stripes, stripe, d_mask, stream);
}

int rts_trace_shadow_mask(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* positions,
                          uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, uint8_t* mask) {
    if (!c || !k || !positions || !mask || W == 0 || H == 0 || row_begin > row_end || row_end > H)
        return RTS_ERR_INVALID_ARG;
    if (!c->d_bvh) return RTS_ERR_NO_BVH;
    if (row_begin == row_end) return RTS_OK;
    RTS_HIP(hipSetDevice(c->device));
    // Only the stripe travels: the device buffers hold rows [row_begin,row_end) as a frame of their own.
    const uint32_t rows = row_end - row_begin;
    const size_t inB = (size_t)rows * W * 16, outB = (size_t)rows * W;
    int s = ensure(&c->d_in, &c->inBytes, inB);
    if (s == RTS_OK) s = ensure(&c->d_out, &c->outBytes, outB);
    if (s != RTS_OK) return s;
    RTS_HIP(hipMemcpy(c->d_in, positions + (size_t)row_begin * W * 4, inB, hipMemcpyHostToDevice));
    c->pixelBase = row_begin * W;          // per-pixel jitter hashes the pixel's index in the caller's frame
    s = rts_trace_shadow_mask_device(c, k, light, (const float*)c->d_in, W, rows, 0, rows, (uint8_t*)c->d_out, nullptr);
    c->pixelBase = 0;
    if (s != RTS_OK) return s;
    RTS_HIP(hipMemcpy(mask + (size_t)row_begin * W, c->d_out, outB, hipMemcpyDeviceToHost));
    return RTS_OK;
}

int rts_trace_rays_device(rts_ctx* c, const rts_ray* d_rays, size_t n, uint8_t* d_out, void* stream) {
    if (!c || (n && (!d_rays || !d_out)) || n > (1ull << 38)) return RTS_ERR_INVALID_ARG;   // grid.x is 31-bit
    TraceParams p;
    int s = fillParams(c, p);
    if (s != RTS_OK) return s;
    if (n == 0) return RTS_OK;
    RTS_HIP(hipSetDevice(c->device));
    p.rays = d_rays; p.out = d_out; p.nrays = n;
    // generic rays carry no coherence promise: lane-per-ray with in-wave work sharing unless the caller asks for a
    // variant (random segments: 2x the plain loop; coherent rays: equal; profiles/r01/generic_rays_variants.log)
    const int variant = c->variant == rts::V_AUTO ? rts::V_SHARE : c->variant;
    c->lastKernel = rts::kernelName(variant, false);
    ++c->launches;
    return hipStatus(rts::launchTraceRays(variant, p, (hipStream_t)stream));
}

int rts_trace_rays(rts_ctx* c, const rts_ray* rays, size_t n, uint8_t* out) {
    if (!c || (n && (!rays || !out))) return RTS_ERR_INVALID_ARG;
    if (!c->d_bvh) return RTS_ERR_NO_BVH;
    if (n == 0) return RTS_OK;
    RTS_HIP(hipSetDevice(c->device));
    int s = ensure(&c->d_in, &c->inBytes, n * sizeof(rts_ray));
    if (s == RTS_OK) s = ensure(&c->d_out, &c->outBytes, n);
    if (s != RTS_OK) return s;
    RTS_HIP(hipMemcpy(c->d_in, rays, n * sizeof(rts_ray), hipMemcpyHostToDevice));
    s = rts_trace_rays_device(c, (const rts_ray*)c->d_in, n, (uint8_t*)c->d_out, nullptr);
    if (s != RTS_OK) return s;
    RTS_HIP(hipMemcpy(out, c->d_out, n, hipMemcpyDeviceToHost));
    return RTS_OK;
}

int rts_device_malloc(rts_ctx* c, void** p, size_t bytes) {
    if (!c || !p) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMalloc(p, bytes ? bytes : 16));
    return RTS_OK;
}
int rts_device_free(rts_ctx* c, void* p) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipFree(p));
    return RTS_OK;
}
int rts_memcpy_h2d(rts_ctx* c, void* d, const void* s, size_t bytes) {
    if (!c || !d || !s) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMemcpy(d, s, bytes, hipMemcpyHostToDevice));
    return RTS_OK;
}
int rts_memcpy_d2h(rts_ctx* c, void* d, const void* s, size_t bytes) {
    if (!c || !d || !s) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost));
    return RTS_OK;
}
int rts_device_mem_info(rts_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    RTS_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return RTS_OK;
}
int rts_stream_create(rts_ctx* c, void** stream) {
    if (!c || !stream) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    hipStream_t s = nullptr;
    RTS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return RTS_OK;
}
int rts_stream_destroy(rts_ctx* c, void* stream) {
    if (!c || !stream) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipStreamDestroy((hipStream_t)stream));
    return RTS_OK;
}
int rts_stream_synchronize(rts_ctx* c, void* stream) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RTS_OK;
}
int rts_timer_begin(rts_ctx* c, void* stream) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipEventRecord(c->ev0, (hipStream_t)stream));
    return RTS_OK;
}
int rts_timer_end(rts_ctx* c, void* stream) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipEventRecord(c->ev1, (hipStream_t)stream));
    return RTS_OK;
}
int rts_timer_elapsed_ms(rts_ctx* c, float* ms) {
    if (!c || !ms) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipEventSynchronize(c->ev1));
    RTS_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return RTS_OK;
}
int rts_timer_mark(rts_ctx* c, void* stream, uint32_t slot) {
    if (!c || slot >= 65536u) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    try {
        if (c->marks.size() <= slot) c->marks.resize((size_t)slot + 1, nullptr);
    } catch (...) {
        return RTS_ERR_CAPACITY;
    }
    if (!c->marks[slot]) RTS_HIP(hipEventCreate(&c->marks[slot]));
    RTS_HIP(hipEventRecord(c->marks[slot], (hipStream_t)stream));
    return RTS_OK;
}
int rts_timer_between_ms(rts_ctx* c, uint32_t a, uint32_t b, float* ms) {
    if (!c || !ms || a >= c->marks.size() || b >= c->marks.size() || !c->marks[a] || !c->marks[b]) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipEventSynchronize(c->marks[b]));
    RTS_HIP(hipEventElapsedTime(ms, c->marks[a], c->marks[b]));
    return RTS_OK;
}
const char* rts_ctx_last_kernel_name(rts_ctx* c) { return c ? c->lastKernel : ""; }

int rts_ctx_device_ordinal(rts_ctx* c) { return c ? c->device : 0; }

// used by the GPU builders (rts_lbvh.hip): one buffer the context keeps between builds (a rebuild per frame pays no
// hipMalloc / hipFree: fifty of them cost more than the build).  NULL if it cannot be had; the builders then allocate
// every buffer on their own (DeviceArena::getOwn).
void* rts_ctx_scratch(rts_ctx* c, size_t bytes) {
    if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (bytes > c->scratchBytes) {
        // grows by at least half (a scene that grows a little every frame must not pay a hipFree -- a device-wide
        // synchronisation -- per frame); option "builder_scratch" = 0 gives the memory back
        size_t want = bytes > c->scratchBytes + c->scratchBytes / 2 ? bytes : c->scratchBytes + c->scratchBytes / 2;
        if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratchBytes = 0; }
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        c->d_scratch = p; c->scratchBytes = want;
    }
    return c->d_scratch;
}

// used by the GPU builder (rts_lbvh.hip): the context takes ownership of a packed stream that is already on the device
// (RTS_OK), or refuses it and leaves it with the caller (any other status)
int rts_ctx_adopt_device_bvh(rts_ctx* c, void* d_packed, size_t count, uint32_t P) {
    if (!c || !d_packed || count != (size_t)5 * P - 2 || count * 16 >= 0xFFFFFF00ull) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    if (c->d_bvh) { void* old = c->d_bvh; c->d_bvh = nullptr; c->bvhVec4 = 0; c->P = 0; RTS_HIP(hipFree(old)); }
    c->d_bvh = d_packed;
    c->bvhVec4 = count; c->P = P;
    clearSplits(c);
    if (c->tileOrderPlanned) (void)rts_ctx_set_tile_order(c, nullptr, 0);
    // the same checks as for an uploaded stream, on the device: layout, finiteness (edges of finite vertices can
    // overflow), box order, enclosure.  A refused stream stays the caller's to free.
    return finishInstall(c, false);
}

// used by the harness (rts_primary.hip): the device copy of the packed stream, NULL before rts_ctx_set_bvh
const void* rts_ctx_device_bvh(rts_ctx* c) {
    if (!c) return nullptr;
    (void)hipSetDevice(c->device);
    return c->d_bvh;
}

// Dispatch order for the next traces whose block count equals `count` (NULL/0 = natural order).  Every tile index
// must appear exactly once (checked).  Speed only: results never depend on it.
int rts_ctx_set_tile_order(rts_ctx* c, const uint32_t* order, size_t count) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    if (c->d_tileOrder) { RTS_HIP(hipFree(c->d_tileOrder)); c->d_tileOrder = nullptr; c->tileOrderCount = 0; }
    c->tileOrderSquare = 0; c->tileOrderBlock = 0; c->tileOrderSoftOnly = false; c->tileOrderPlanned = false;
    if (!order || count == 0) return RTS_OK;
    try {
        std::vector<uint8_t> seen(count, 0);
        for (size_t i = 0; i < count; ++i) { if (order[i] >= count || seen[order[i]]) return RTS_ERR_INVALID_ARG; seen[order[i]] = 1; }
    } catch (...) {
        return RTS_ERR_CAPACITY;                   // no exception crosses the C ABI
    }
    RTS_HIP(hipMalloc((void**)&c->d_tileOrder, count * 4));
    RTS_HIP(hipMemcpy(c->d_tileOrder, order, count * 4, hipMemcpyHostToDevice));
    c->tileOrderCount = count;
    return RTS_OK;
}

int rts_ctx_read_wave_stats(rts_ctx* c, uint64_t* out, size_t waves) {
    if (!c || !out || !c->d_waveStats || waves * 32 > c->waveStatsBytes) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMemcpy(out, c->d_waveStats, waves * 32, hipMemcpyDeviceToHost));
    return RTS_OK;
}

int rts_ctx_read_clock_probe(rts_ctx* c, uint64_t* out, size_t rows) {
    if (!c || !out || !c->d_clockProbe || rows > c->clockProbeRows) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMemcpy(out, c->d_clockProbe, rows * 32, hipMemcpyDeviceToHost));
    return RTS_OK;
}

// Wave statistics of ONE dispatch (what rts_ctx_read_wave_stats / _realtime return): two launches of the diagnostic
// instantiation without a table, the second one read back.  Used by the planner and, once per tuning call, by the tuner.
static int measureDispatch(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                           uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                           std::vector<uint64_t>& stats, std::vector<uint64_t>& rt, uint32_t perTile = 1) {
    int status = RTS_OK;
    uint64_t* keep = c->d_waveStats; const size_t keepBytes = c->waveStatsBytes;
    c->d_waveStats = nullptr; c->waveStatsBytes = 0;
    uint32_t rows = row_end - row_begin;
    if (n_stripes > 1) { const uint32_t bands = (H + band_rows - 1) / band_rows; rows = ((bands - stripe + n_stripes - 1) / n_stripes) * band_rows; }
    const size_t tiles = (size_t)((W + 7) / 8) * ((rows + 7) / 8);
    const size_t waves = tiles * perTile;                             // (soft shadows, "soft_split": 4 waves per tile)
    hipError_t e = hipMalloc((void**)&c->d_waveStats, waves * 64);
    if (e == hipSuccess) e = hipMemset(c->d_waveStats, 0, waves * 64);
    if (e == hipSuccess) {
        c->waveStatsBytes = waves * 32;
        const int use = c->useSplits; c->useSplits = 0;
        for (int i = 0; i < 2 && status == RTS_OK; ++i)                   // (the second launch is the one that counts: warm caches)
            status = traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, nullptr);
        c->useSplits = use;
        if (status == RTS_OK) e = hipDeviceSynchronize();
        if (status == RTS_OK && e == hipSuccess && (size_t)c->lastBlocksX * c->lastBlocksY != tiles) status = planRefused("not a dispatch of 8x8 tiles");
        if (status == RTS_OK && e == hipSuccess) {
            stats.resize(waves * 4); rt.resize(waves * 4);
            e = hipMemcpy(stats.data(), c->d_waveStats, waves * 32, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(rt.data(), c->d_waveStats + waves * 4, waves * 32, hipMemcpyDeviceToHost);
        }
    }
    if (c->d_waveStats) (void)hipFree(c->d_waveStats);
    c->d_waveStats = keep; c->waveStatsBytes = keepBytes;
    if (status != RTS_OK) return status;
    return hipStatus(e);
}

// ---- the order of a split table's front records: host logic, also reachable without a device (rtsh_split_front_order) --------
struct SplitSel { float us; uint32_t tile; };     // a tile (bx | by << 16) and the life it is sorted by

// half-octaves of life (0.5 / 1 / 2 / 4 bands per octave and the exact order measured: profiles/r04/table_order_bands.log -- 2 and 4 are level)
static int lifeBand(float us) { return (int)std::floor(std::log2(us < 0.25f ? 0.25f : us) * 2.f); }

// life_block B: the front order knows the image only in blocks of B x B tiles -- a tile is as long as the longest tile of its
// block.  Coarser, so a little less gain on the frame it was measured on, but a camera that moves shifts what is long by whole
// tiles and leaves the blocks' order nearly as it was (profiles/r04/table_granularity.log).
static std::unordered_map<uint32_t, float> blockLives(const std::vector<SplitSel>& tiles, uint32_t B) {
    std::unordered_map<uint32_t, float> longest;
    for (const SplitSel& t : tiles) {
        float& m = longest[((t.tile & 0xFFFFu) / B) | (((t.tile >> 16) / B) << 16)];
        if (t.us > m) m = t.us;
    }
    return longest;
}

// front tiles: longest first in half-octaves of life, image order inside one (neighbouring tiles walk the same part of the
// tree: started together they share the scalar cache and the L2 as in the plain launch)
static void sortFront(std::vector<SplitSel>& front) {
    std::sort(front.begin(), front.end(), [](const SplitSel& a, const SplitSel& b) {
        const int ba = lifeBand(a.us), bb = lifeBand(b.us);
        if (ba != bb) return ba > bb;
        const uint32_t ka = ((a.tile >> 16) << 16) | (a.tile & 0xFFFFu), kb = ((b.tile >> 16) << 16) | (b.tile & 0xFFFFu);
        return ka < kb;
    });
}

// xcd_square: the workgroups of a dispatch go round-robin over the 8 XCDs (record i runs on XCD i mod 8, each with an L2 of its
// own), so inside a band record i is taken from the tiles of "its" S x S-tile squares of the image: an XCD's L2 then holds the
// part of the tree its squares see instead of every XCD holding all of it.  (As a static placement of the plain launch this
// lost -- regions differ in cost and unbalance the XCDs, EXPERIMENTS.md --; inside a band of equal measured life every XCD gets
// the same number of equally long tiles.  profiles/r04/xcd_regions_in_table_order.log)  Bands stay where they are, and the
// order inside a bucket stays the image order.
static void dealOverXcds(std::vector<SplitSel>& front, uint32_t firstRecord, uint32_t S) {
    const size_t F = front.size();
    std::vector<SplitSel> out; out.reserve(F);
    for (size_t i = 0; i < F;) {
        size_t j = i; const int b = lifeBand(front[i].us);
        while (j < F && lifeBand(front[j].us) == b) ++j;
        std::vector<SplitSel> bucket[8]; size_t head[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (size_t q = i; q < j; ++q) {
            const uint32_t rx = (front[q].tile & 0xFFFFu) / S, ry = (front[q].tile >> 16) / S;
            bucket[(rx + ry * 3u) & 7u].push_back(front[q]);
        }
        for (size_t q = i; q < j; ++q) {
            uint32_t x = (uint32_t)(firstRecord + q) & 7u;              // the XCD this record will run on
            if (head[x] == bucket[x].size()) {                           // none of its own left: from the fullest bucket
                size_t most = 0;
                for (uint32_t y = 0; y < 8; ++y) if (bucket[y].size() - head[y] > most) { most = bucket[y].size() - head[y]; x = y; }
            }
            out.push_back(bucket[x][head[x]++]);
        }
        i = j;
    }
    front.swap(out);
}

// Test hook (tests/test_host_logic.py): the front order of n tiles {life_us[i], tiles[i] = bx | by << 16} as rts_ctx_plan_splits
// would build it with front_share 1 and no splits; order_out[r] = index of the tile that becomes record first_record + r.
extern "C" int rtsh_split_front_order(const float* life_us, const uint32_t* tiles, size_t n, uint32_t first_record, uint32_t xcd_square,
                                      uint32_t life_block, uint32_t* order_out) {
    if ((n && (!life_us || !tiles || !order_out)) || xcd_square > 65535u || life_block > 65535u) return RTS_ERR_INVALID_ARG;
    try {
        std::vector<SplitSel> all(n);
        std::unordered_map<uint32_t, uint32_t> index;
        for (size_t i = 0; i < n; ++i) { all[i] = { life_us[i], tiles[i] }; index[tiles[i]] = (uint32_t)i; }
        if (index.size() != n) return RTS_ERR_INVALID_ARG;                 // a tile twice
        const uint32_t B = life_block > 1 ? life_block : 1u;
        std::vector<SplitSel> front = all;
        if (B > 1) {
            const auto longest = blockLives(all, B);
            for (SplitSel& t : front) t.us = longest.at(((t.tile & 0xFFFFu) / B) | (((t.tile >> 16) / B) << 16));
        }
        sortFront(front);
        if (xcd_square) dealOverXcds(front, first_record, xcd_square);
        for (size_t r = 0; r < n; ++r) order_out[r] = index[front[r].tile];
    } catch (...) { return RTS_ERR_CAPACITY; }
    return RTS_OK;
}

// Plans the split table for ONE dispatch geometry (see include/rts.h).  Synchronous, default stream.
static int planSplitsImpl(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                          uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                          const rts_split_plan* plan, uint32_t* tiles_out, uint32_t* pieces_out) {
    if (tiles_out) *tiles_out = 0;
    if (pieces_out) *pieces_out = 0;
    if (!c || !k || !d_positions || !d_mask || !plan || !(plan->min_life_us > 0.f) || !(plan->piece_us > 0.f) || !(plan->end_after_us >= 0.f) ||
        !(plan->front_life_us >= 0.f) || plan->front_life_us > plan->min_life_us || !(plan->front_share >= 0.f) || plan->front_share > 1.f ||
        plan->xcd_square > 65535u || plan->life_block > 65535u)
        return planRefused("arguments");
    if (light && light->nsamples > 1) return planRefused("more than one light sample");   // (soft shadows are dealt over waves by "soft_split")
    RTS_HIP(hipSetDevice(c->device));
    clearSplits(c);
    if (!c->wideCount) return RTS_OK;                                            // pieces walk the private copy: none, no table
    if (n_stripes > 1 && stripe >= (H + band_rows - 1) / band_rows) return RTS_OK;   // a stripe without a band: nothing to launch, no table
    const uint32_t maxPieces = plan->max_pieces < 2 ? 2 : (plan->max_pieces > 64 ? 64 : plan->max_pieces);
    const uint32_t maxTiles = plan->max_tiles ? (plan->max_tiles > 65536u ? 65536u : plan->max_tiles) : 4096u;
    const uint32_t logCap = 16384;
    int status = RTS_OK;
    std::vector<uint64_t> stats, rt;
    size_t waves = 0;
    try {
        if (plan->prev_stats && plan->prev_realtime && plan->prev_waves) {      // the caller's statistics of an earlier frame
            waves = plan->prev_waves;
            stats.assign(plan->prev_stats, plan->prev_stats + waves * 4);
            rt.assign(plan->prev_realtime, plan->prev_realtime + waves * 4);
        } else {                                                                 // one launch of this dispatch with wave statistics
            status = measureDispatch(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, stats, rt);
            if (status != RTS_OK) return status;
            waves = stats.size() / 4;
        }
        // the tiles whose wave lived longer than min_life_us, longest first
        using Sel = SplitSel;
        std::vector<Sel> sel, front;                                            // to be split / to be started first, unsplit
        const uint32_t blocksX = (W + 7) / 8;
        uint32_t blocksY = 0;
        uint64_t began = ~0ull;                                                 // the dispatch's first wave
        for (size_t i = 0; i < waves; ++i) if (rt[i * 4 + 1] > rt[i * 4] && rt[i * 4] < began) began = rt[i * 4];
        float frontLife = plan->front_life_us;
        if (plan->front_share >= 1.f) frontLife = 1e-6f;                          // (every tile: the whole dispatch in table order)
        else if (plan->front_share > 0.f) {                                      // the life the longest front_share of the tiles exceed
            std::vector<float> lives;
            lives.reserve(waves);
            for (size_t i = 0; i < waves; ++i) if (rt[i * 4 + 1] > rt[i * 4]) lives.push_back((float)(rt[i * 4 + 1] - rt[i * 4]) * 0.01f);
            if (!lives.empty()) {
                size_t nth = (size_t)((1.0 - (double)plan->front_share) * (double)lives.size());
                if (nth >= lives.size()) nth = lives.size() - 1;
                std::nth_element(lives.begin(), lives.begin() + nth, lives.end());
                if (lives[nth] > frontLife) frontLife = lives[nth];
            }
        }
        const uint32_t B = plan->life_block > 1 ? plan->life_block : 1u;       // (blockLives)
        std::unordered_map<uint32_t, float> blockLife;
        if (B > 1) {
            std::vector<SplitSel> all;
            all.reserve(waves);
            for (size_t i = 0; i < waves; ++i) {
                const uint64_t r0 = rt[i * 4], r1 = rt[i * 4 + 1];
                if (r1 <= r0) continue;
                all.push_back({ (float)(r1 - r0) * 0.01f, (uint32_t)(stats[i * 4 + 3] >> 48) | (((uint32_t)(stats[i * 4 + 3] >> 32) & 0xFFFFu) << 16) });
            }
            blockLife = blockLives(all, B);
        }
        for (size_t i = 0; i < waves; ++i) {
            const uint64_t r0 = rt[i * 4], r1 = rt[i * 4 + 1];
            if (r1 <= r0) continue;
            const float us = (float)(r1 - r0) * 0.01f;
            const uint32_t bx = (uint32_t)(stats[i * 4 + 3] >> 48), by = (uint32_t)(stats[i * 4 + 3] >> 32) & 0xFFFFu;
            if (by >= blocksY) blocksY = by + 1;
            if (bx >= blocksX) continue;
            if (us > plan->min_life_us && (float)(r1 - began) * 0.01f > plan->end_after_us) sel.push_back({ us, bx | (by << 16) });
            else if (frontLife > 0.f && us > frontLife) front.push_back({ B > 1 ? blockLife[(bx / B) | ((by / B) << 16)] : us, bx | (by << 16) });
        }
        if (sel.empty() && front.empty()) return RTS_OK;
        const auto longer = [](const Sel& a, const Sel& b) { return a.us > b.us || (a.us == b.us && a.tile < b.tile); };
        std::sort(sel.begin(), sel.end(), longer);
        sortFront(front);
        if (sel.size() > maxTiles) { front.insert(front.begin(), sel.begin() + maxTiles, sel.end()); sel.resize(maxTiles); }   // (what is not split starts first at least)
        if (front.size() > 262144) front.resize(262144);
        const uint32_t T = (uint32_t)sel.size(), F = (uint32_t)front.size();
        std::vector<rts::SplitCut> cuts(T);
        std::vector<uint32_t> first(T), provisional((size_t)T * 8, 0u);
        uint32_t nPieces = 0;
        for (uint32_t t = 0; t < T; ++t) {
            uint32_t S = (uint32_t)std::ceil(sel[t].us / plan->piece_us);
            S = S < 2 ? 2 : (S > maxPieces ? maxPieces : S);
            cuts[t] = { sel[t].tile, S };
            first[t] = nPieces;
            nPieces += S;
            provisional[(size_t)t * 8 + 0] = sel[t].tile; provisional[(size_t)t * 8 + 1] = 0; provisional[(size_t)t * 8 + 2] = 0xFFFFFFFFu;
            provisional[(size_t)t * 8 + 3] = t | (1u << 24);                       // (dword 4: the walk starts at the root, offset 0)
        }
        if (plan->xcd_square) dealOverXcds(front, nPieces, plan->xcd_square);
        // the dispatch this table belongs to (what traceMaskImpl will compute for the same arguments)
        uint32_t rows = row_end - row_begin;
        if (n_stripes > 1) { const uint32_t bands = (H + band_rows - 1) / band_rows; rows = ((bands - stripe + n_stripes - 1) / n_stripes) * band_rows; }
        const uint32_t keyBlocksY = (rows + 7) / 8;
        if (blocksY > keyBlocksY) return planRefused("statistics of another dispatch");
        std::vector<uint32_t> bitmap(((size_t)blocksX * keyBlocksY + 31) / 32 + 1, 0u);
        std::vector<uint32_t> frontRecords((size_t)F * 8, 0u);                   // {tile, 0, END, 0 = "front tile", 0...}
        for (uint32_t t = 0; t < T + F; ++t) {
            const uint32_t tile = t < T ? sel[t].tile : front[t - T].tile;
            const uint32_t id = (tile >> 16) * blocksX + (tile & 0xFFFFu);
            bitmap[id >> 5] |= 1u << (id & 31u);
            if (t >= T) { frontRecords[(size_t)(t - T) * 8] = tile; frontRecords[(size_t)(t - T) * 8 + 2] = 0xFFFFFFFFu; }
        }
        // device side: planning walk of the selected tiles (one piece each, everything logged), then the quantiles
        const uint32_t frontStride = (nPieces + F + 7u) / 8u;
        std::vector<uint32_t> frontMap((size_t)frontStride * 8u, 0xFFFFFFFFu);   // (TraceParams::frontMap: record i at (i mod 8) * stride + i / 8)
        for (uint32_t j = 0; j < F; ++j) { const uint32_t id = nPieces + j; frontMap[(size_t)(id & 7u) * frontStride + (id >> 3)] = front[j].tile; }
        void *d_cuts = nullptr, *d_first = nullptr, *d_prov = nullptr, *d_log = nullptr, *d_state = nullptr, *d_pieces = nullptr, *d_map = nullptr;
        void* d_front = nullptr;
        const size_t logBytes = (size_t)T * (logCap + 1) * 4 + 16;
        hipError_t e = hipMalloc(&d_cuts, (size_t)T * 8 + 16);
        if (e == hipSuccess) e = hipMalloc(&d_first, (size_t)T * 4 + 16);
        if (e == hipSuccess) e = hipMalloc(&d_prov, (size_t)T * 32 + 16);
        if (e == hipSuccess) e = hipMalloc(&d_log, logBytes);
        if (e == hipSuccess) e = hipMalloc(&d_state, (size_t)T * 16 + 16);
        if (e == hipSuccess) e = hipMalloc(&d_pieces, (size_t)(nPieces + F) * 32 + 16);
        if (e == hipSuccess && F) e = hipMemcpy((char*)d_pieces + (size_t)nPieces * 32, frontRecords.data(), (size_t)F * 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&d_map, bitmap.size() * 4);
        if (e == hipSuccess) e = hipMalloc(&d_front, frontMap.size() * 4 + 16);
        if (e == hipSuccess) e = hipMemcpy(d_front, frontMap.data(), frontMap.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cuts, cuts.data(), (size_t)T * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_first, first.data(), (size_t)T * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_prov, provisional.data(), (size_t)T * 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_map, bitmap.data(), bitmap.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(d_log, 0, logBytes);
        if (e == hipSuccess) e = hipMemset(d_state, 0, (size_t)T * 16 + 16);
        if (e == hipSuccess && T) {
            const rts_ctx::Planning pl{ (const uint32_t*)d_prov, T, (T + blocksX - 1) / blocksX, (uint64_t*)d_state, (uint32_t*)d_log, logCap };
            c->planning = &pl;
            status = traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, nullptr);
            c->planning = nullptr;
            if (status == RTS_OK) e = rts::launchSplitQuantiles((const uint32_t*)d_log, logCap, (const rts::SplitCut*)d_cuts, (const uint32_t*)d_first,
                                                                T, (uint32_t*)d_pieces, c->d_wide, nullptr);
            if (status == RTS_OK && e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (d_cuts) (void)hipFree(d_cuts);
        if (d_first) (void)hipFree(d_first);
        if (d_prov) (void)hipFree(d_prov);
        if (d_log) (void)hipFree(d_log);
        if (d_state) (void)hipFree(d_state);
        if (status != RTS_OK || e != hipSuccess) {
            if (d_pieces) (void)hipFree(d_pieces);
            if (d_map) (void)hipFree(d_map);
            if (d_front) (void)hipFree(d_front);
            (void)hipGetLastError();
            return status != RTS_OK ? status : hipStatus(e);
        }
        rts_ctx::Splits& t = c->splits;
        t.valid = true;
        t.W = W; t.H = H; t.rowBegin = row_begin; t.rowEnd = row_end; t.bandRows = band_rows; t.nStripes = n_stripes; t.stripe = stripe;
        t.blocksX = blocksX; t.blocksY = keyBlocksY;
        t.d_skipMap = (uint32_t*)d_map; t.d_pieces = (uint32_t*)d_pieces;
        t.d_frontMap = (uint32_t*)d_front; t.frontStride = frontStride;
        t.nPieces = nPieces + F; t.pieceRows = (nPieces + F + blocksX - 1) / blocksX; t.nTiles = T; t.nFront = F;
        t.allTiles = (size_t)T + F == waves && waves == (size_t)blocksX * keyBlocksY;   // every tile has a record: no tile rows are launched
        t.plan = *plan; t.plan.prev_stats = nullptr; t.plan.prev_realtime = nullptr; t.plan.prev_waves = 0;
        if (tiles_out) *tiles_out = T + F;
        if (pieces_out) *pieces_out = nPieces + F;
    } catch (...) {
        c->planning = nullptr;
        return RTS_ERR_CAPACITY;                       // no exception crosses the C ABI
    }
    return RTS_OK;
}

// rts_ctx_plan_tile_order: the dispatch order of a split table's whole-dispatch form for the launches that cannot carry a table --
// soft shadows (several samples per pixel, 4 waves per tile).  Wave statistics of this dispatch, a tile as long as its longest wave,
// the order of sortFront / dealOverXcds (bands of life, longest first, each band dealt over the XCDs by image squares), installed as
// the context's tile order (rts_ctx_set_tile_order: workgroup i walks tile order[i]).
static int planTileOrderImpl(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                             uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                             uint32_t xcd_square, uint32_t life_block, uint32_t* tiles_out) {
    if (tiles_out) *tiles_out = 0;
    if (!c || !k || !d_positions || !d_mask || xcd_square > 65535u || life_block > 65535u) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    int status = rts_ctx_set_tile_order(c, nullptr, 0);
    if (status != RTS_OK) return status;
    if (c->blockWaves != 1 || c->swizzle) return RTS_OK;                           // (one tile per workgroup only)
    const uint32_t perTile = (light && light->nsamples > 1 && c->softSplit) ? 4u : 1u;
    std::vector<uint64_t> stats, rt;
    status = measureDispatch(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, stats, rt, perTile);
    if (status == RTS_ERR_INVALID_ARG) return RTS_OK;                               // not a dispatch of 8x8 tiles: no order
    if (status != RTS_OK) return status;
    if (c->lastVariant != rts::V_PACKET && c->lastVariant != rts::V_WIDE) return RTS_OK;
    try {
        const uint32_t blocksX = c->lastBlocksX, nTiles = c->lastBlocksX * c->lastBlocksY;
        std::vector<float> life(nTiles, 0.f);
        for (size_t i = 0; i < stats.size() / 4; ++i) {
            const uint64_t r0 = rt[i * 4], r1 = rt[i * 4 + 1];
            if (r1 <= r0) continue;
            const uint32_t bx = (uint32_t)(stats[i * 4 + 3] >> 48), by = (uint32_t)(stats[i * 4 + 3] >> 32) & 0xFFFFu;
            if (bx >= blocksX || by >= c->lastBlocksY) continue;
            const float us = (float)(r1 - r0) * 0.01f;
            float& m = life[(size_t)by * blocksX + bx];
            if (us > m) m = us;                                                     // a tile is as long as its longest wave
        }
        std::vector<SplitSel> front(nTiles);
        for (uint32_t t = 0; t < nTiles; ++t) front[t] = { life[t] > 0.f ? life[t] : 0.25f, (t % blocksX) | ((t / blocksX) << 16) };
        if (life_block > 1) {
            const auto longest = blockLives(front, life_block);
            for (SplitSel& f : front) f.us = longest.at(((f.tile & 0xFFFFu) / life_block) | (((f.tile >> 16) / life_block) << 16));
        }
        sortFront(front);
        if (xcd_square) dealOverXcds(front, 0u, xcd_square);
        std::vector<uint32_t> order(nTiles);
        for (uint32_t i = 0; i < nTiles; ++i) order[i] = (front[i].tile >> 16) * blocksX + (front[i].tile & 0xFFFFu);
        status = rts_ctx_set_tile_order(c, order.data(), order.size());
        if (status != RTS_OK) return status;
        c->tileOrderSquare = xcd_square; c->tileOrderBlock = life_block;
        c->tileOrderSoftOnly = light && light->nsamples > 1;
        c->tileOrderPlanned = true;
        if (tiles_out) *tiles_out = nTiles;
    } catch (...) { return RTS_ERR_CAPACITY; }
    return RTS_OK;
}

// Picks the kernel for THIS dispatch by timing the candidates on it (what a renderer does once per scene and resolution):
// the lane-per-ray walk with work sharing, the packet kernel, the wide packet kernel (when the stream has a private copy).
// Then, for a packet kernel, two launch parameters that are worth 2-6 % on some frames and cost as much on others
// (profiles/r03/autotune_stage2_sweep.log): the dissolve threshold ("packet_share" 4 or 6) and the order in which the tile
// rows are started ("row_order" top-down or bottom-up: the rows started last are the kernel's tail).  Third stage (round 4):
// a split table for the tiles measured to be long (rts_ctx_plan_splits), kept when it gains 1.5 % -- frames whose time is a
// few long waves (atrium 1080p: -30 %), and the stripes of a multi-GPU frame; a frame that is throughput-bound to its end
// (city, courtyard at 4K) keeps the plain launch.  Leaves the options and the table of the winners installed.  Results
// never depend on any of them.
static int autotuneImpl(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                        uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                        int* chosen, float* ms_out) {
    if (!c || !k || !d_positions || !d_mask) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    clearSplits(c);
    int status = RTS_OK;
    int reps = 5;                              // (nine for the tables of a dispatch below 0.1 ms: its launches are cheap, its noise is not)
    auto median5 = [&](float* out) {          // two untimed launches, then the median of five (or nine)
        float times[9];
        for (int i = -2; i < reps; ++i) {
            hipError_t e = hipEventRecord(c->ev0, nullptr);
            if (e != hipSuccess) { status = hipStatus(e); return false; }
            status = traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, nullptr);
            if (status != RTS_OK) return false;
            float ms = 0;
            e = hipEventRecord(c->ev1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
            if (e != hipSuccess) { status = hipStatus(e); return false; }
            if (i >= 0) times[i] = ms;
        }
        for (int i = 1; i < reps; ++i) for (int j = i; j > 0 && times[j] < times[j - 1]; --j) { float t = times[j]; times[j] = times[j - 1]; times[j - 1] = t; }
        *out = times[reps / 2];
        return true;
    };
    // (the wide kernel first, and a later candidate must beat the best by 2 %: at equal frame time the wide kernel's waves
    //  are shorter, which is what a striped multi-GPU frame needs -- tools/stripe_scaling.py)
    const int candidates[3] = { rts::V_WIDE, rts::V_PACKET, rts::V_SHARE };
    const int before = c->variant, shareBefore = c->packetShare, orderBefore = c->rowOrder;
    auto giveUp = [&]() {
        c->variant = before; c->packetShare = shareBefore; c->rowOrder = orderBefore; clearSplits(c);
        if (c->tileOrderPlanned) (void)rts_ctx_set_tile_order(c, nullptr, 0);
        return status;
    };
    if (c->tileOrderPlanned) {                                // (an order an earlier tuning planned: the candidates meet the plain dispatch)
        status = rts_ctx_set_tile_order(c, nullptr, 0);
        if (status != RTS_OK) return status;
    }
    uint64_t pixels = (uint64_t)W * (row_end - row_begin);
    if (n_stripes > 1) pixels /= n_stripes;
    int best = before;
    float bestMs = 1e30f;
    for (int v : candidates) {
        if (v == rts::V_WIDE && !c->wideCount) continue;
        if (v == rts::V_SHARE && pixels > (1u << 20)) continue;                  // (never close on a big frame: skip its long launches)
        c->variant = v;
        float ms;
        if (!median5(&ms)) return giveUp();
        if (ms < bestMs * 0.98f) { bestMs = ms; best = v; }
    }
    c->variant = best;
    if (best == rts::V_WIDE || best == rts::V_PACKET) {
        // second stage, one parameter at a time; a change must gain 1.5 % to be kept (launch-to-launch noise is below 1 %)
        float ms;
        if (c->packetShare == 4) {
            c->packetShare = 6;
            if (!median5(&ms)) return giveUp();
            if (ms < bestMs * 0.985f) bestMs = ms; else c->packetShare = shareBefore;
        }
        if (c->rowOrder == 0 && !c->d_tileOrder && !c->swizzle && n_stripes <= 1) {
            c->rowOrder = 1;
            if (!median5(&ms)) return giveUp();
            if (ms < bestMs * 0.985f) bestMs = ms; else c->rowOrder = orderBefore;
        }
        // third stage: split tables.  Tiles that lived longer than a share of the dispatch and ended in its later part.
        if (c->wideCount && c->blockWaves == 1 && !c->wideLane && (!light || light->nsamples <= 1) && c->useSplits) {
            // Seven tables from ONE set of wave statistics (profiles/r04/front_tiles_sweep.log, whole_dispatch_order.log): the tiles
            // that lived longer than max(T/4, 20 us) and ended in the second half split into pieces of max(T/10, 8 us) -- or none
            // split --, with the longest 3 % / third / ALL of the tiles started first (all: the whole dispatch in table order, no
            // tile rows at all); and the splits alone with a lower threshold.  A frame whose time is a few long waves wants the
            // first kind (atrium), a throughput-bound one a short front list (city) or the table order (courtyard), the stripe of
            // a multi-GPU frame both.
            const float T = bestMs * 1000.f;                                      // us
            if (bestMs < 0.1f) { reps = 9; if (!median5(&bestMs)) return giveUp(); }   // (the plain launch once more, by the same measure)
            const int trials = 7;
            int kept = -1, installed = -1;
            rts_split_plan plan{};
            std::vector<uint64_t> tuneStats, tuneRt;
            status = measureDispatch(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, tuneStats, tuneRt);
            if (status == RTS_ERR_INVALID_ARG) { status = RTS_OK; goto tuned; }    // (not a dispatch of 8x8 tiles: no table)
            if (status != RTS_OK) return giveUp();
            {
            auto fill = [&](int i) {
                plan = rts_split_plan{};
                plan.max_pieces = 8;
                plan.max_tiles = 8192;
                plan.prev_stats = tuneStats.data(); plan.prev_realtime = tuneRt.data(); plan.prev_waves = tuneStats.size() / 4;
                static const float share[3] = { 0.03f, 1.f / 3.f, 1.f };
                if (i < 6) {
                    plan.front_share = share[i % 3];
                    if (i % 3 == 2) plan.xcd_square = 32;                       // (the whole dispatch in table order: XCD-local too)
                    if (c->tuneForMotion) plan.life_block = 16;                  // (sorted by 128 x 128-pixel blocks: see rts_split_plan)
                    if (i < 3) {
                        plan.min_life_us = 0.25f * T < 20.f ? 20.f : 0.25f * T; plan.end_after_us = 0.5f * T;
                        plan.piece_us = 0.1f * T < 8.f ? 8.f : 0.1f * T;
                    } else { plan.min_life_us = 1e9f; plan.piece_us = 1e9f; }
                } else {
                    plan.min_life_us = 0.15f * T < 8.f ? 8.f : 0.15f * T; plan.end_after_us = 0.5f * T; plan.piece_us = plan.min_life_us * 0.5f;
                }
            };
            // Every table is timed against the PLAIN launch measured right beside it (planning leaves the device idle for
            // milliseconds and its clocks drop: a figure from before the planning is not comparable; 20 ms of the launch itself
            // first).  The table with the best ratio is the candidate; it must gain 1 %.
            float bestRatio = 1e30f, tableMs = bestMs;
            for (int i = 0; i < trials; ++i) {
                // (a camera path: pieces belong to the very tiles of one camera, and a front list that is a share of the tiles is a set
                //  of tiles too -- only the whole dispatch in block order was measured to keep: profiles/r04/table_granularity.log)
                if (c->tuneForMotion && i != 5) continue;
                fill(i);
                uint32_t tiles = 0;
                status = planSplitsImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, &plan, &tiles, nullptr);
                if (status != RTS_OK) return giveUp();
                installed = i;
                if (!tiles) continue;
                for (const auto t0 = std::chrono::steady_clock::now(); status == RTS_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(20);) {
                    for (int w = 0; w < 8 && status == RTS_OK; ++w)
                        status = traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, nullptr);
                    if (hipStreamSynchronize(nullptr) != hipSuccess) status = RTS_ERR_HIP;
                }
                if (status != RTS_OK) return giveUp();
                float plainMs;
                if (!median5(&ms)) return giveUp();
                c->useSplits = 0;
                const bool ok = median5(&plainMs);
                c->useSplits = 1;
                if (!ok) return giveUp();
                if (getenv("RTS_TUNE_LOG"))           // diagnostics: what the tuner saw
                    fprintf(stderr, "rts tune: table %d (life > %.1f us, front share %.2f): %u split + %u front tiles, %.4f ms against %.4f plain beside it\n", i,
                            plan.min_life_us, plan.front_share, c->splits.nTiles, c->splits.nFront, ms, plainMs);
                if (ms / plainMs < bestRatio) { bestRatio = ms / plainMs; tableMs = ms; kept = i; }
            }
            if (kept >= 0 && bestRatio < 0.99f) bestMs = tableMs; else kept = -1;    // (paired figures: 1 % is outside their noise)
            if (kept < 0) clearSplits(c);
            else if (kept != installed) {
                fill(kept);
                status = planSplitsImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, &plan, nullptr, nullptr);
                if (status != RTS_OK) return giveUp();
            }
            }
        } else if (light && light->nsamples > 1 && c->blockWaves == 1 && !c->swizzle && c->useTileOrder) {
            // ... for soft shadows (no table: 4 waves per tile) the whole-dispatch order as a tile order: bands of measured life,
            // longest first, dealt over the XCDs by image squares (city x 16 samples - 5.6 %, courtyard - 6.3 %:
            // profiles/r04/soft_tile_order.log); sorted by blocks when the camera will move.  Timed beside the plain launch.
            // Both packet families are tried in their own order (the first stage chose between them in the plain order, where they
            // are within 2 % on these frames; in order the stackless packet gains more: city x 16 samples 2.25 against 2.39 ms).
            auto ordered = [&](int variant, float* ms, uint32_t* tiles) {
                c->variant = variant;
                status = planTileOrderImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, 32u,
                                           c->tuneForMotion ? 16u : 0u, tiles);
                if (status != RTS_OK) return false;
                if (!*tiles) return true;
                for (const auto t0 = std::chrono::steady_clock::now(); status == RTS_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(20);) {
                    status = traceMaskImpl(c, k, light, d_positions, W, H, row_begin, row_end, band_rows, n_stripes, stripe, d_mask, nullptr);
                    if (hipStreamSynchronize(nullptr) != hipSuccess) status = RTS_ERR_HIP;
                }
                return status == RTS_OK && median5(ms);
            };
            const int other = best == rts::V_WIDE ? rts::V_PACKET : (c->wideCount ? rts::V_WIDE : -1);
            uint32_t tiles = 0, tilesOther = 0;
            float ms = 0.f, msOther = 1e30f;
            if (other >= 0) { if (!ordered(other, &msOther, &tilesOther)) return giveUp(); if (!tilesOther) msOther = 1e30f; }
            if (!ordered(best, &ms, &tiles)) return giveUp();                     // (last: its order is the one left installed)
            if (tiles && msOther < ms * 0.98f) {                                   // the other family, in its order, is ahead: take it
                best = other;
                if (!ordered(best, &ms, &tiles)) return giveUp();
            }
            c->variant = best;
            if (tiles) {
                float plainMs;
                c->useTileOrder = 0;
                const bool ok = median5(&plainMs);
                c->useTileOrder = 1;
                if (!ok) return giveUp();
                if (getenv("RTS_TUNE_LOG"))
                    fprintf(stderr, "rts tune: tile order (%u tiles, kernel %d): %.4f ms against %.4f plain beside it (the other family in its order: %.4f)\n",
                            tiles, best, ms, plainMs, msOther);
                if (ms / plainMs < 0.99f) bestMs = ms;
                else { status = rts_ctx_set_tile_order(c, nullptr, 0); if (status != RTS_OK) return giveUp(); }
            }
        }
    }
tuned:
    if (chosen) *chosen = best;
    if (ms_out) *ms_out = bestMs;
    return RTS_OK;
}

int rts_ctx_plan_tile_order(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                            uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, uint32_t xcd_square, uint32_t life_block,
                            uint32_t* tiles) {
    if (W == 0 || H == 0 || n_stripes == 0 || stripe >= n_stripes || (n_stripes > 1 && (band_rows == 0 || band_rows % 8 != 0))) return RTS_ERR_INVALID_ARG;
    if (n_stripes == 1) return planTileOrderImpl(c, k, light, d_positions, W, H, 0, H, 0, 1, 0, d_mask, xcd_square, life_block, tiles);
    return planTileOrderImpl(c, k, light, d_positions, W, H, 0, H, band_rows, n_stripes, stripe, d_mask, xcd_square, life_block, tiles);
}

int rts_ctx_autotune(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W,
                     uint32_t H, uint8_t* d_mask, int* chosen, float* ms_out) {
    if (W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    return autotuneImpl(c, k, light, d_positions, W, H, 0, H, 0, 1, 0, d_mask, chosen, ms_out);
}

int rts_ctx_autotune_stripes(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                             uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, int* chosen, float* ms_out) {
    if (W == 0 || H == 0 || band_rows == 0 || band_rows % 8 != 0 || n_stripes == 0 || stripe >= n_stripes) return RTS_ERR_INVALID_ARG;
    if (n_stripes == 1) return autotuneImpl(c, k, light, d_positions, W, H, 0, H, 0, 1, 0, d_mask, chosen, ms_out);
    return autotuneImpl(c, k, light, d_positions, W, H, 0, H, band_rows, n_stripes, stripe, d_mask, chosen, ms_out);
}

int rts_ctx_get_split_plan(rts_ctx* c, rts_split_plan* out) {
    if (!c || !out || !c->splits.valid) return RTS_ERR_INVALID_ARG;
    *out = c->splits.plan;
    return RTS_OK;
}

int rts_ctx_plan_splits(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                        uint32_t row_begin, uint32_t row_end, uint8_t* d_mask, const rts_split_plan* plan, uint32_t* tiles, uint32_t* pieces) {
    if (W == 0 || H == 0 || row_begin >= row_end || row_end > H) return RTS_ERR_INVALID_ARG;
    return planSplitsImpl(c, k, light, d_positions, W, H, row_begin, row_end, 0, 1, 0, d_mask, plan, tiles, pieces);
}

int rts_ctx_plan_splits_stripes(rts_ctx* c, const rts_constants* k, const rts_light* light, const float* d_positions, uint32_t W, uint32_t H,
                                uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, const rts_split_plan* plan,
                                uint32_t* tiles, uint32_t* pieces) {
    if (W == 0 || H == 0 || band_rows == 0 || band_rows % 8 != 0 || n_stripes == 0 || stripe >= n_stripes) return RTS_ERR_INVALID_ARG;
    if (n_stripes == 1) return planSplitsImpl(c, k, light, d_positions, W, H, 0, H, 0, 1, 0, d_mask, plan, tiles, pieces);
    return planSplitsImpl(c, k, light, d_positions, W, H, 0, H, band_rows, n_stripes, stripe, d_mask, plan, tiles, pieces);
}

// diagnostics: per piece of the installed table {tile x | y << 16, first node, end node, pieces of its tile} and the 100 MHz
// stamps of its start and end in the last launch that used the table (option "piece_stats")
int rts_ctx_read_piece_stats(rts_ctx* c, uint32_t* records, uint64_t* clocks, size_t pieces) {
    if (!c || !c->splits.valid || pieces > c->splits.nPieces || (clocks && (!c->d_pieceClock || pieces > c->pieceClockCount))) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    if (records) RTS_HIP(hipMemcpy(records, c->splits.d_pieces, pieces * 32, hipMemcpyDeviceToHost));
    if (clocks) RTS_HIP(hipMemcpy(clocks, c->d_pieceClock, pieces * 64, hipMemcpyDeviceToHost));
    return RTS_OK;
}

// The kernels replace 1.0f / x by v_rcp_f32 + one Newton step where 2^-100 <= |x| <= 2^100 (rts_kernels.hip: rcpFast); that this
// is the IEEE quotient for EVERY such bit pattern is checked here, on the device it runs on: out[0] = patterns in the range,
// out[1] = patterns whose result differs from the division (must be 0), out[2] = one of them.
int rts_selftest_reciprocal(rts_ctx* c, uint64_t out[3]) {
    if (!c || !out) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    RTS_HIP(hipMalloc((void**)&d, 32));
    hipError_t e = hipMemset(d, 0, 32);
    if (e == hipSuccess) e = rts::launchReciprocalSelfTest(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return hipStatus(e);
}

int rts_ctx_clear_splits(rts_ctx* c) {
    if (!c) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    clearSplits(c);
    return RTS_OK;
}

int rts_ctx_read_wave_realtime(rts_ctx* c, uint64_t* out, size_t waves) {
    if (!c || !out || !c->d_waveStats || waves * 32 > c->waveStatsBytes) return RTS_ERR_INVALID_ARG;
    RTS_HIP(hipSetDevice(c->device));
    RTS_HIP(hipMemcpy(out, c->d_waveStats + c->waveStatsBytes / 8, waves * 32, hipMemcpyDeviceToHost));
    return RTS_OK;
}

} // extern "C"
