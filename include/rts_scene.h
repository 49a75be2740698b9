/* =============================================================================
 * rts_scene.h -- HARNESS entry points of librts.so (not part of the drop-in path).
 *
 * The reference feeds its shadow kernel from a Vulkan G-buffer pass and an OBJ file; neither
 * exists here, so the harness synthesises the same inputs:
 *   * rtsh_primary_positions : the RGBA32F "camera-relative world position" target that
 *     Source/Shaders/Model.frag:35,39 writes (closest hit per pixel centre through the same packed
 *     BVH; background pixels = (0,0,0,0), i.e. the clear value).
 *   * rtsh_obj_* : OBJ reader with the semantics of External/zeux_objparser/objparser.cpp and the
 *     flat-vertex expansion of RayTracedShadowsApp::loadModel (Source/RayTracedShadows.cpp:783-824).
 * ========================================================================== */
#ifndef RTS_SCENE_H
#define RTS_SCENE_H

#include <stddef.h>
#include <stdint.h>
#include "rts.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Pinhole camera as set up at Source/RayTracedShadows.cpp:238-242 (vertical fov in radians, lookAt
 * eye -> target, +Y up).  positions: W*H*4 floats.  hit_count (nullable) receives the number of
 * pixels that hit geometry.  threads: 0 = all host threads. */
int rtsh_primary_positions(const rts_vec4u* packed, size_t count_vec4, const float eye[3],
                           const float target[3], float fovy, uint32_t W, uint32_t H,
                           float* positions, uint64_t* hit_count, int threads);

/* Same plus the normal target (Model.frag:35,38: RGBA, face normal turned towards the viewer; 0 = background).
 * normals may be NULL. */
int rtsh_primary_gbuffer(const rts_vec4u* packed, size_t count_vec4, const float eye[3], const float target[3],
                         float fovy, uint32_t W, uint32_t H, float* positions, float* normals,
                         uint64_t* hit_count, int threads);

/* The same pass on the GPU: traces through the BVH already uploaded to `ctx`, writes DEVICE
 * buffers (W*H*4 floats each, d_normals may be NULL), asynchronous on `stream`.  Produces the same bits as the host
 * version (shared code, same FP rules). */
int rtsh_primary_gbuffer_device(rts_ctx* ctx, const float eye[3], const float target[3], float fovy,
                                uint32_t W, uint32_t H, float* d_positions, float* d_normals, void* stream);

/* Combine pass (Source/Shaders/Combine.frag:18-37 with the default white material):
 * rgb[W*H*3] = 255 * (1.25*max(0,N.L)*mask/samples + 0.15 + 0.05*(1 - max(0, N.-cameraDirection))), 0 where the
 * normal is 0.  light == NULL: directional light from constants->lightDirection; positions needed for point lights. */
int rtsh_combine(const rts_constants* constants, const rts_light* light, const float* positions, const float* normals,
                 const uint8_t* mask, uint32_t W, uint32_t H, uint8_t* rgb);

/* The combine pass on the GPU (same per-pixel arithmetic, shared source): DEVICE buffers, d_rgb = W*H*3 bytes,
 * asynchronous on `stream`.  Together with rtsh_primary_gbuffer_device and rts_trace_shadow_mask_device the whole
 * frame -- G-buffer, shadow mask, lighting -- stays on the device (tools/render.py). */
int rtsh_combine_device(rts_ctx* ctx, const rts_constants* constants, const rts_light* light, const float* d_positions,
                        const float* d_normals, const uint8_t* d_mask, uint32_t W, uint32_t H, uint8_t* d_rgb,
                        void* stream);

/* OBJ ingest.  rtsh_obj_load parses `path` and expands it to the reference's flat
 * Vertex stream: 8 floats per vertex (position.xyz, normal.xyz, texcoord.uv), indices[i] = i.
 * Call with vertices == NULL to query *vertex_count (3 per triangle) first.  Returns RTS_OK,
 * RTS_ERR_INVALID_ARG (cannot open / fails objValidate) or RTS_ERR_CAPACITY. */
int rtsh_obj_load(const char* path, float* vertices, size_t vertex_capacity, uint32_t* vertex_count,
                  float bbox_min[3], float bbox_max[3]);

/* The parser's number reader (objparser.cpp:62-131), exposed so tests can pin it. */
float rtsh_obj_parse_float(const char* text, int* consumed);

/* Host logic of rts_ctx_plan_splits, reachable without a device (tests): the order of a split table's front records for n tiles
 * {life_us[i], tiles[i] = bx | by << 16} with front_share 1 and no splits -- half-octave bands of life, longest first, image order
 * inside a band; life_block B > 1: a tile counts as long as the longest tile of its B x B block; xcd_square S > 0: inside a band,
 * record first_record + r is taken from the tiles of XCD ((first_record + r) mod 8)'s S x S squares while it has any (include/rts.h,
 * rts_split_plan).  order_out[r] = index of the tile that becomes record first_record + r. */
int rtsh_split_front_order(const float* life_us, const uint32_t* tiles, size_t n, uint32_t first_record, uint32_t xcd_square,
                           uint32_t life_block, uint32_t* order_out);

#ifdef __cplusplus
}
#endif
#endif /* RTS_SCENE_H */
