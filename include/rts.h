/* =============================================================================
 * rts.h -- C ABI of the MI355X-native shadow-ray path (librts.so).
 *
 * Drop-in boundary for the ONE hot path of kayru/RayTracedShadows:
 *     BVHBuilder::build  ->  packed vec4 node stream  ->  any-hit shadow kernel
 * Plain pointers and sizes only; no C++/torch types cross this line.  Every entry
 * returns an int status (RTS_OK == 0) and never throws.
 *
 * Each declaration cites the reference interface it replaces (paths relative to
 * the reference checkout).  The reference-side binding a maintainer would add is
 * shown in INTEGRATION.md.
 * ========================================================================== */
#ifndef RTS_H
#define RTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
    RTS_OK              = 0,
    RTS_ERR_INVALID_ARG = 1, /* NULL pointer, prim_count == 0, bad row range ...           */
    RTS_ERR_CAPACITY    = 2, /* output buffer too small                                    */
    RTS_ERR_NONFINITE   = 3, /* NaN/Inf vertex (reference: unbounded recursion, SURVEY E-4) */
    RTS_ERR_NO_BVH      = 4, /* trace called before rts_ctx_set_bvh                         */
    RTS_ERR_BAD_BVH     = 5, /* packed buffer fails structural validation                   */
    RTS_ERR_DEGENERATE  = 6, /* finite vertices whose extents overflow the SAH cost to +inf: no split position
                                exists (reference: unbounded recursion, SURVEY E-4/E-5); RTS_GPU_BUILD_SAH also: a
                                tree deeper than 262 144 levels (hundreds of thousands of triangles with EQUAL boxes split
                                off one per level; the reference recurses as deep as that chain is long)        */
    RTS_ERR_HIP         = 100 /* 100 + hipError_t                                           */
};
const char* rts_status_string(int status);

/* ---- data contract -------------------------------------------------------- */

/* Source/BVHBuilder.h:22-25 `struct BVHPackedNode {u32 a,b,c,d;}` == GLSL `vec4 bvhNodes[]`
 * (Source/Shaders/RayTracedShadows.comp:23-26).  Layout of the whole buffer: SURVEY.md Appendix A. */
typedef struct rts_vec4u { uint32_t a, b, c, d; } rts_vec4u;

/* Source/BVHBuilder.h:8-20 `struct BVHNode` (unpacked, DFS order; prim==0xFFFFFFFF <=> inner). */
typedef struct rts_bvh_node {
    float bboxMin[3]; uint32_t prim;
    float bboxMax[3]; uint32_t next;
} rts_bvh_node;

/* Source/RayTracedShadows.h:56-62 `struct RayTracingConstants` == UBO `Constants`
 * (RayTracedShadows.comp:3-9).  Only cameraPosition.xyz and lightDirection.xyz are read. */
typedef struct rts_constants {
    float cameraPosition[4];
    float cameraDirection[4];
    float lightDirection[4];
    float renderTargetSize[4];
} rts_constants;

/* RayTracedShadows.comp:28-32 `struct Ray { vec4 o; vec4 d; }`: o.w = tmax, d.w unused. */
typedef struct rts_ray { float o[4]; float d[4]; } rts_ray;

/* Light model.  type RTS_LIGHT_DIRECTIONAL with xyz = constants.lightDirection.xyz is exactly the
 * reference (RayTracedShadows.comp:134-146).  RTS_LIGHT_POINT and nsamples > 1 are the extensions
 * BASELINE.json's configs ask for:
 *   point : o0 = cam + rel; bias as comp:138-140; dn = (L-o0)/|L-o0|; o = o0 + dn*bias;
 *           d = L - o (un-normalised); tmax = 1
 *   nsamples in [2,64]: sample j uses L + offsets[j].xyz; the output byte is the number of
 *   UNoccluded samples (0..nsamples) instead of 0/1.  `table` != 0: per-pixel jitter, see the field. */
enum { RTS_LIGHT_DIRECTIONAL = 0, RTS_LIGHT_POINT = 1 };
typedef struct rts_light {
    uint32_t type;
    uint32_t nsamples;      /* 0 or 1 = hard shadow */
    float    xyz[3];
    uint32_t table;         /* 0: sample j uses offsets[j] in every pixel (one coherent pass per sample).
                               T in [nsamples, 64]: PER-PIXEL jitter -- offsets[] holds T entries and pixel p = y*W + x of the
                               frame uses offsets[(start(p) + j) mod T], start(p) = (hash32(p) * T) >> 32 with
                               hash32(v): v ^= v >> 16; v *= 0x7feb352d; v ^= v >> 15; v *= 0x846ca68b; v ^= v >> 16
                               (32-bit wrap-around): integer arithmetic only, the same on every device and in every
                               stripe.  Neighbouring pixels then aim at different points of the light in the same pass,
                               which is what stresses ray packets (BASELINE configs[4]).  (The field was `reserved`, 0.) */
    float    offsets[64][4];
} rts_light;

typedef struct rts_ctx rts_ctx;

/* ---- producer: replaces BVHBuilder::build (Source/BVHBuilder.h:31, BVHBuilder.cpp:248-368;
 *      call site Source/RayTracedShadows.cpp:1031-1037) ---------------------- */

/* m_packedNodes.size() for prim_count triangles = 2*(2P-1) + P = 5P-2 (BVHBuilder.cpp:308-367). */
size_t rts_bvh_packed_count(uint32_t prim_count);
/* m_nodes.size() = 2P-1. */
size_t rts_bvh_node_count(uint32_t prim_count);

/* vertices/stride/indices/prim_count: identical meaning to BVHBuilder::build -- `stride_floats`
 * is in floats (the reference passes sizeof(Vertex)/sizeof(float) == 8).  out_packed receives
 * m_packedNodes (capacity in vec4 >= rts_bvh_packed_count); out_nodes (nullable) receives m_nodes.
 * Tail .d words are 0 (reference: uninitialised). */
int rts_bvh_build(const float* vertices, uint32_t stride_floats, const uint32_t* indices,
                  uint32_t prim_count, rts_vec4u* out_packed, size_t out_capacity_vec4,
                  rts_bvh_node* out_nodes);

/* Same with the two knobs the reference hard-codes: sah_prim_limit (1000000 at BVHBuilder.cpp:83;
 * ranges larger than this use the spatial-median split) and the number of host threads
 * (0 = all; the tree does not depend on it). */
int rts_bvh_build_ex(const float* vertices, uint32_t stride_floats, const uint32_t* indices,
                     uint32_t prim_count, uint32_t sah_prim_limit, int threads,
                     rts_vec4u* out_packed, size_t out_capacity_vec4, rts_bvh_node* out_nodes);

/* Structural validation of a packed buffer from ANY producer (ours or the reference's):
 * count == 5P-2, leaf/inner tags, strictly-forward miss links, tail pointers in range. */
int rts_bvh_validate(const rts_vec4u* packed, size_t count_vec4, uint32_t* prim_count_out);

/* BVH build ON THE GPU: four producers of the same packed layout with the reference's layout rules
 * (larger-area child first, DFS numbering, miss links, tail; BVHBuilder.cpp:202-244, 308-367).
 * rts_bvh_build_device = RTS_GPU_BUILD_SAH (BVHBuilder's tree); rts_bvh_build_device_ex picks the topology:
 *   RTS_GPU_BUILD_LBVH  Karras hierarchy + bottom-up bounds (fastest build)
 *   RTS_GPU_BUILD_PLOC  parallel locally-ordered clustering, `radius` = Morton neighbours searched each way (0 = 16)
 *   RTS_GPU_BUILD_PLOC_SAH  PLOC down to <= 65 536 clusters, then the top of the tree over the clusters' boxes by the
 *                       reference's split rule (full-sweep SAH, BVHBuilder.cpp:78-156, weighted by triangle counts) on
 *                       the host: ~10x the build time of PLOC, a tree closer to BVHBuilder's
 *   RTS_GPU_BUILD_SAH   BVHBuilder's own rule for every node, level by level on the device: full-sweep SAH on three
 *                       axes (cpp:78-156), spatial median above `radius` triangles per range (cpp:157-178; 0 = the
 *                       reference's 1 000 000), larger-area child first.  Triangles with EQUAL centroids on an axis are
 *                       ordered by triangle id where the reference's std::sort leaves them unspecified: on a mesh
 *                       without such ties the stream is byte-identical to rts_bvh_build's, otherwise a tree of the
 *                       same quality.  RTS_ERR_DEGENERATE where the reference would not terminate.
 * vertex_floats = number of floats in `vertices`.  `vertices` and `indices` may be host pointers (copied to the device) or
 * device pointers on the context's device (used where they lie: a renderer's vertex and index buffers).  out_packed (host, nullable) receives the 5P-2 vec4; install != 0
 * makes the stream the context's BVH without a host round trip.  build_ms (nullable): device time of the build.
 * The builders' working memory (about 0.6 KB per triangle) stays with the context until rts_ctx_destroy, so that a rebuild
 * per frame allocates nothing. */
enum { RTS_GPU_BUILD_LBVH = 0, RTS_GPU_BUILD_PLOC = 1, RTS_GPU_BUILD_PLOC_SAH = 2, RTS_GPU_BUILD_SAH = 3 };
int rts_bvh_build_device(rts_ctx* ctx, const float* vertices, size_t vertex_floats, uint32_t stride_floats,
                         const uint32_t* indices, uint32_t prim_count, rts_vec4u* out_packed,
                         size_t out_capacity_vec4, int install, float* build_ms);
int rts_bvh_build_device_ex(rts_ctx* ctx, const float* vertices, size_t vertex_floats, uint32_t stride_floats,
                            const uint32_t* indices, uint32_t prim_count, int algorithm, uint32_t radius,
                            rts_vec4u* out_packed, size_t out_capacity_vec4, int install, float* build_ms);

/* ---- consumer: replaces the bind-group + dispatch of
 *      RayTracedShadowsApp::renderShadowMaskCompute (Source/RayTracedShadows.cpp:570-595) and the
 *      BVH upload (Source/RayTracedShadows.cpp:1039-1044) ---------------------- */

/* One context per device; not thread-safe; owns the device copy of the BVH. */
int rts_ctx_create(int device_ordinal, rts_ctx** out);
int rts_ctx_destroy(rts_ctx* ctx);

/* == Gfx_CreateBuffer(Storage, stride 16, count, m_packedNodes.data()) (cpp:1039-1044).
 * Validates the structure on the host, copies H2D once (the device copy is the same Appendix-A bytes), then decides on
 * the device what the kernels may assume (finite, ordered, enclosing boxes) and derives the private copy of kernel 8. */
int rts_ctx_set_bvh(rts_ctx* ctx, const rts_vec4u* packed, size_t count_vec4);

/* Tuning knobs.  Results never depend on any of them (tests/test_gpu_parity.py).  Unknown key or bad value ->
 * RTS_ERR_INVALID_ARG.
 *   "kernel"        -1 = auto (default: variant 7 below 256 K pixels, the packet kernel 3 from there, the wide packet 8 for
 *                   one-sample dispatches of >= 4 M pixels when the stream has a private copy); 0 straight,
 *                   1 while-while, 2 postpone, 3 packet (8x8 px / wave), 4 packet2 (16x8), 5 packet4 (16x16),
 *                   6 packet + successor prefetch, 7 lane-per-ray with work sharing, 8 WIDE packet (a private copy of the
 *                   stream with four boxes per node: one dependent fetch decides two levels of the reference's walk; a
 *                   stream without a private copy runs 3), 9 the same with the loop compiled instead of hand-written.
 *                   get "kernel_count" = 10.  rts_ctx_autotune picks between 3, 8 and 7 by timing them on the frame.
 *   "wide_copy"     1 (default): derive the private copy for kernel 8 whenever a stream is installed (on the device,
 *                   about 0.2 KB per triangle; needs a finite stream of ordered, enclosing boxes, at most 512 levels deep,
 *                   at most 2^24 triangles -- otherwise there simply is none); 0: never
 *   "wide_lane"     0 (default): a dissolved wide packet continues lane per ray over the stream, stackless; 1: over the wide
 *                   nodes with a 16-entry stack per lane in LDS (4 KB per wave: a CU then holds 28 instead of 32 waves)
 *   "soft_split"    1 (default): soft shadows (nsamples > 1) with kernel 3 or 8 run 4 waves per tile, each a quarter of the
 *                   samples (the counts meet in LDS); 0: one wave walks a pixel's samples one after the other
 *   "packet_budget" side-steps between two coherence checks of a packet (default 16)
 *   "packet_share"  a packet dissolves when it picks up fewer than share/16 of its live rays per side-step (default 4)
 *   "block_waves"   waves per workgroup of the packet kernels: 1 (default) or 4
 *   "xcd_swizzle"   1 = contiguous image chunk per XCD (default 0: measured slower)
 *   "row_order"     order in which the tile rows of a frame are dispatched: 0 first to last (default), 1 last to first,
 *                   2 middle row outwards (the rows dispatched last are the kernel's tail; which order wins depends on
 *                   where the scene's long rays are: profiles/r02/row_order_sweep.log)
 *   "tile_splits"   1 (default): traces use an installed split table (rts_ctx_plan_splits); 0: they ignore it
 *   "tune_for_motion" 0 (default): rts_ctx_autotune picks the fastest table for THIS frame; 1: only a table that keeps over a
 *                   camera path -- the whole dispatch in table order sorted by blocks of 16 x 16 tiles (rts_split_plan:
 *                   life_block, xcd_square), no pieces, no front lists
 *   "piece_stats"   diagnostics, see rts_ctx_read_piece_stats;  get only: "split_tiles", "front_tiles", "split_pieces"
 *   "lds_pad"       experiment: extra dynamic LDS bytes per one-wave packet workgroup (throttles occupancy; default 0)
 *   "wave_stats"    diagnostics, see rts_ctx_read_wave_stats
 *   "clock_probe"   diagnostics, see rts_ctx_read_clock_probe
 *   "builder_scratch" set 0: release the working memory the GPU builders keep between builds; get: MiB held
 *   get only: "bvh_finite", "bvh_ordered", "bvh_enclosed" (what the installed stream allows: decided by one kernel over all
 *   nodes at upload / adoption), "wide_nodes", "wide_levels" (size of the private copy, 0 = none) */
int rts_ctx_set_option(rts_ctx* ctx, const char* key, int value);
int rts_ctx_get_option(rts_ctx* ctx, const char* key, int* value);

/* == Gfx_Dispatch(divUp(W,8), divUp(H,8), 1) of RayTracedShadows.comp (cpp:576-592) restricted to
 * rows [row_begin,row_end) (row stripes for multi-GPU).
 *   constants : the 64-byte UBO; cameraPosition.xyz is read.  light == NULL means the reference's
 *               directional light taken from constants->lightDirection.xyz.
 *   positions : binding 2, RGBA32F W x H, row-major, camera-relative world position (Model.frag:35)
 *   mask      : binding 3, W x H bytes; 1 = lit, 0 = occluded (comp:148; polarity of the
 *               reference); rows outside the range are not touched.
 * HOST pointers; copies in, traces, copies out, synchronises. */
int rts_trace_shadow_mask(rts_ctx* ctx, const rts_constants* constants, const rts_light* light,
                          const float* positions, uint32_t W, uint32_t H,
                          uint32_t row_begin, uint32_t row_end, uint8_t* mask);

/* Same with DEVICE pointers, asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int rts_trace_shadow_mask_device(rts_ctx* ctx, const rts_constants* constants, const rts_light* light,
                                 const float* d_positions, uint32_t W, uint32_t H,
                                 uint32_t row_begin, uint32_t row_end, uint8_t* d_mask, void* stream);

/* Interleaved row stripes in ONE dispatch (multi-GPU strong scaling): the frame is cut
 * into bands of band_rows rows (a multiple of the kernel's workgroup height: 8 for the default packet kernel, 16 or 32 for
 * the others -- 32 always works) dealt round-robin to n_stripes devices; this call
 * traces the bands stripe, stripe + n_stripes, ... and touches no other row of d_mask. */
int rts_trace_shadow_mask_stripes_device(rts_ctx* ctx, const rts_constants* constants, const rts_light* light,
                                         const float* d_positions, uint32_t W, uint32_t H, uint32_t band_rows,
                                         uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask, void* stream);

/* Generic rays (the shader's `Ray`): out[i] = 1 if ray i is NOT occluded.  Host / device forms. */
int rts_trace_rays(rts_ctx* ctx, const rts_ray* rays, size_t n, uint8_t* out);
int rts_trace_rays_device(rts_ctx* ctx, const rts_ray* d_rays, size_t n, uint8_t* d_out, void* stream);

/* ---- device-memory and timing plumbing (so callers need no HIP headers) ------ */
int rts_device_count(int* count);
int rts_device_malloc(rts_ctx* ctx, void** d_ptr, size_t bytes);
int rts_device_free(rts_ctx* ctx, void* d_ptr);
int rts_memcpy_h2d(rts_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int rts_memcpy_d2h(rts_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int rts_stream_synchronize(rts_ctx* ctx, void* stream);
/* A stream of the context's device for the `stream` arguments above (a renderer passes its own hipStream_t).  Dispatches on
 * different streams may overlap: with two frames in flight the tail of one frame's dispatch runs beside the next one's. */
int rts_stream_create(rts_ctx* ctx, void** stream);
int rts_stream_destroy(rts_ctx* ctx, void* stream);
/* hipEvent pair on `stream` == Gfx_BeginTimer/EndTimer(Timestamp_Shadows) (cpp:572,594).
 * rts_timer_end records the stop event; rts_timer_elapsed_ms synchronises on it. */
int rts_timer_begin(rts_ctx* ctx, void* stream);
int rts_timer_end(rts_ctx* ctx, void* stream);
int rts_timer_elapsed_ms(rts_ctx* ctx, float* ms);
/* Per-launch timing: hipEvents in numbered slots (0..65535, created on first use).  rts_timer_mark records slot `slot`
 * on `stream`; rts_timer_between_ms synchronises on slot_b and returns the time from slot_a to slot_b.  A mark before
 * every dispatch of a frame loop gives the per-frame GPU times whose median the reference shows
 * (120-frame average of Timestamp_Shadows, cpp:263-265). */
int rts_timer_mark(rts_ctx* ctx, void* stream, uint32_t slot);
int rts_timer_between_ms(rts_ctx* ctx, uint32_t slot_a, uint32_t slot_b, float* ms);
/* Name of the kernel the last trace launched (for matching rocprofv3 rows). */
const char* rts_ctx_last_kernel_name(rts_ctx* ctx);
/* Dispatch order of the image tiles for traces whose workgroup count equals `count`: workgroup i works on tile
 * order[i] (a permutation of 0..count-1; NULL or 0 restores the natural order).  Speed only. */
int rts_ctx_set_tile_order(rts_ctx* ctx, const uint32_t* order, size_t count);
/* ... planned from one measured launch of THIS dispatch (any light, any number of samples; stripes as in
 * rts_trace_shadow_mask_stripes_device, n_stripes 1 = the whole frame): the order a whole-dispatch split table runs in -- half-
 * octave bands of measured tile life, longest first, each band dealt over the 8 XCDs by xcd_square x xcd_square-tile image squares
 * (0: not), a tile counted as long as the longest of its life_block x life_block block (0 / 1: itself) --, for the launches that
 * carry no table: soft shadows (16 samples on the city - 5.6 %, on the courtyard - 6.3 %).  rts_ctx_autotune does this for
 * dispatches of more than one sample and keeps it when it gains 1 %.  *tiles = tiles ordered (0: none -- not a dispatch of one
 * 8x8 tile per workgroup).  An order planned on a dispatch of several samples is used by such dispatches only: one-sample traces
 * of the same size keep their everyday launch (and their split table).  Options: "tile_order" 0 makes traces ignore the
 * installed order; get "tile_order_tiles", "tile_order_planned", "tile_order_square", "tile_order_block".  Synchronous, default
 * stream.  Speed only. */
int rts_ctx_plan_tile_order(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                            uint32_t W, uint32_t H, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                            uint32_t xcd_square, uint32_t life_block, uint32_t* tiles);
/* Free and total device memory in bytes (hipMemGetInfo on the context's device); either pointer may be NULL. */
int rts_device_mem_info(rts_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Diagnostics (tools/wave_stats.py): after rts_ctx_set_option(ctx, "wave_stats", n_waves) the packet
 * kernels record 4 x u64 per wave: start clock, end clock, {dissolved flag (bit 0) | lane-per-ray iterations after the
 * dissolve (bits 8-31) | clocks from start to the dissolve (32-63)}, {tile x (48-63) | tile y (32-47) | lane-steps in
 * those iterations (0-31)};
 * this copies them out. */
int rts_ctx_read_wave_stats(rts_ctx* ctx, uint64_t* out, size_t waves);
/* After rts_ctx_set_option(ctx, "clock_probe", tile_rows) every launch of a packet kernel on a 2-D grid -- the everyday
 * instantiation included, so the launches that are TIMED -- stamps, for the first wave of each tile row, {shader clock at
 * start, at end, 100 MHz clock at start, at end} (4 x u64 per row): clock held = sum(d shader) / sum(d 100 MHz) * 100 MHz. */
int rts_ctx_read_clock_probe(rts_ctx* ctx, uint64_t* out, size_t rows);
/* Picks the kernel for this frame by timing the candidates on it (lane-per-ray with work sharing for small frames, the
 * packet kernel, the wide packet kernel) -- what a renderer does once per scene and resolution; then, for a packet kernel,
 * the dissolve threshold ("packet_share" 4 or 6) and the order in which the tile rows are started ("row_order" 0 or 1), each
 * kept only if it gains 1.5 %; then seven split tables (rts_ctx_plan_splits below) planned from one set of wave statistics -- the
 * tiles that lived longer than a quarter of the frame split or not, the longest 3 % / third / all of the tiles started first --,
 * the fastest kept on the same condition (any table installed before is dropped).  Leaves the options "kernel", "packet_share" and "row_order" set to the winners and the winning table installed
 * (*chosen = the kernel, median of five launches in *ms; both nullable).  Device pointers, default stream, synchronous.
 * Results never depend on any of it. */
int rts_ctx_autotune(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                     uint32_t W, uint32_t H, uint8_t* d_mask, int* chosen, float* ms);
/* The same for the dispatch of rts_trace_shadow_mask_stripes_device with these band_rows / n_stripes / stripe: what rank
 * `stripe` of an n_stripes-GPU frame launches is what it tunes -- kernel, dissolve threshold, and the split
 * table for its own rows (one eighth of a 4K frame is two rounds of the chip's wave slots: its time is its longest waves). */
int rts_ctx_autotune_stripes(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                             uint32_t W, uint32_t H, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                             int* chosen, float* ms);
/* ---- split tiles: the few tiles that are measured to be LONG walked by several waves -------------------------------------
 * The reference maps one 8x8 tile to one 64-thread group (RayTracedShadows.comp:127, dispatch RayTracedShadows.cpp:590-592) and
 * so does every kernel here; a frame's time is then often its few longest waves.  A split table breaks that mapping for
 * exactly those tiles: each is walked by S one-wave workgroups ("pieces"), every piece over ONE index range of the node
 * stream with all 64 rays (any-hit is an OR over the ranges; exactness: rts_kernels.hip, "SPLIT TILES").  The pieces are
 * the first workgroups of the SAME dispatch (the longest work starts first), the split tiles' own waves end in their
 * prologue, and launches without a table run the unchanged everyday kernels.
 *
 * rts_ctx_plan_splits measures and installs the table for ONE dispatch geometry (frame size and row range, or stripe):
 *   1. wave statistics of the dispatch (one launch with "wave_stats"), or the caller's statistics of an EARLIER frame
 *      (prev_stats / prev_realtime as rts_ctx_read_wave_stats / rts_ctx_read_wave_realtime return them, prev_waves entries);
 *   2. the tiles whose wave lived longer than min_life_us and ended later than end_after_us (the longest max_tiles of
 *      them) are walked once more, alone, with
 *      their visited node indices logged; tile t gets S = ceil(life / piece_us) pieces (2 .. max_pieces), its ranges cut at
 *      the j/S quantiles of its log.
 * Every later trace with the same geometry, one sample per pixel and kernel 3 or 8 uses the table (option "tile_splits" 0
 * switches that off; get "split_tiles" / "split_pieces" = the table's size); rts_ctx_set_bvh and rts_ctx_clear_splits drop
 * it.  The table only holds node indices and tile coordinates: a camera or light that moves makes it less well balanced,
 * never wrong.  Needs the private copy of kernel 8 (without one no table is made: *tiles = 0).  Synchronous, default stream;
 * device pointers.  Results never depend on any of it (tests/test_gpu_parity.py). */
typedef struct rts_split_plan {
    float    min_life_us;        /* > 0 */
    float    end_after_us;       /* >= 0: ... and only the tiles whose wave ENDED later than this after the dispatch's first wave
                                    started (the waves that end last are the dispatch's tail; 0 = every long tile) */
    float    piece_us;           /* > 0 */
    float    front_life_us;      /* 0, or < min_life_us: the tiles that lived longer than this but are not split are FRONT tiles -- walked
                                    by their own wave, unchanged, but dispatched at the head of the grid (after the pieces, longest
                                    first): what is long starts early, and the dispatch ends with short waves */
    float    front_share;        /* 0..1: ... or, given as a share: the longest front_share of all tiles of the dispatch start first (the
                                    larger of the two thresholds counts when both are given).  1 = every tile: the WHOLE dispatch runs
                                    in table order -- half-octaves of measured life, longest first, image order inside one -- and no
                                    tile rows are launched at all.  rts_ctx_autotune tries 0.03, 1/3 and 1 */
    uint32_t max_pieces;         /* 2..64 */
    uint32_t max_tiles;          /* 0 = 4096 */
    uint32_t xcd_square;         /* 0, or S: inside a band of the front order, record i (which the dispatcher places on XCD i mod 8) is taken
                                    from the S x S-tile squares of the image that belong to that XCD -- each XCD's L2 then holds the part of
                                    the tree its squares see.  Balanced by construction (equal numbers of equally long tiles per XCD);
                                    rts_ctx_autotune uses 32 with front_share 1 */
    uint32_t life_block;         /* 0 / 1, or B: the front order takes a tile to be as long as the longest tile of its block of B x B tiles.
                                    A table sorted by single tiles fits ONE camera (a step of 0.1 % of the view distance moves what is long by
                                    a tile, and a stale order is slower than none); sorted by blocks of 16 it gives up a third of its gain
                                    and keeps the rest over a camera path (rts_ctx option "tune_for_motion") */
    uint32_t reserved_;          /* 0 */
    const uint64_t* prev_stats;  /* all three NULL / 0: measure now */
    const uint64_t* prev_realtime;
    size_t   prev_waves;
} rts_split_plan;
int rts_ctx_plan_splits(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                        uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, uint8_t* d_mask,
                        const rts_split_plan* plan, uint32_t* tiles, uint32_t* pieces);
/* ... for the dispatch of rts_trace_shadow_mask_stripes_device with the same band_rows / n_stripes / stripe */
int rts_ctx_plan_splits_stripes(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                                uint32_t W, uint32_t H, uint32_t band_rows, uint32_t n_stripes, uint32_t stripe, uint8_t* d_mask,
                                const rts_split_plan* plan, uint32_t* tiles, uint32_t* pieces);
int rts_ctx_clear_splits(rts_ctx* ctx);
/* The parameters the installed table was planned with (prev_* NULL): what a caller that tunes in one process and renders in
 * another hands to rts_ctx_plan_splits there.  RTS_ERR_INVALID_ARG without a table. */
int rts_ctx_get_split_plan(rts_ctx* ctx, rts_split_plan* out);
/* Diagnostics (tools/piece_stats.py): the first `pieces` records of the installed table -- 8 x u32 {tile x | tile y << 16, first
 * node, end node, state slot | pieces of the tile << 24, byte offset of the wide node the piece starts at, 0, 0, 0} -- and, after rts_ctx_set_option(ctx, "piece_stats", n), 8 x u64 per
 * piece of the last launch that used the table: 100 MHz clock at {start, end, rays ready, end of the packet phase, start of the
 * lane-per-ray phase, end of the walk}, {wide nodes entered | stack entries at the dissolve << 32}, lanes found occluded.
 * Either pointer may be NULL. */
int rts_ctx_read_piece_stats(rts_ctx* ctx, uint32_t* records, uint64_t* clocks, size_t pieces);
/* Self-test behind one of the kernels' shortcuts: 1.0f / x (comp:44, comp:77) is computed as v_rcp_f32 + one Newton step in
 * fma arithmetic when 2^-100 <= |x| <= 2^100 in a whole wave.  That this is the correctly rounded quotient is checked HERE for
 * every bit pattern of the range on the context's device: out[0] = patterns checked (3 355 443 200), out[1] = patterns whose
 * result differs from the IEEE division -- 0 on gfx950 --, out[2] = one such pattern.  A fraction of a second. */
int rts_selftest_reciprocal(rts_ctx* ctx, uint64_t out[3]);
/* Same launch, 4 x u64 per wave: s_memrealtime (the constant 100 MHz counter) at the wave's start and end, shader clocks
 * from the wave's start to its first ray being ready (G-buffer texel in, ray set up), XCC id.  With the start/end shader
 * clocks above: clock held under load = sum(end - start clocks) / sum(end - start realtime) * 100 MHz. */
int rts_ctx_read_wave_realtime(rts_ctx* ctx, uint64_t* out, size_t waves);

#ifdef __cplusplus
}
#endif
#endif /* RTS_H */
