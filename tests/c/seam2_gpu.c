/* Plain C99 caller of the CONSUMER seam on the GPU (include/rts.h): what a maintainer's replacement of
 * RayTracedShadowsApp::renderShadowMaskCompute (Source/RayTracedShadows.cpp:570-595) does -- create the context, upload
 * the packed stream (cpp:1039-1044), bind positions and mask as device buffers, dispatch, read the mask back -- checked
 * against a committed golden mask (tests/golden/cornell_128.npz, flattened by the test into the file named on the command
 * line).  No Python, no C++ in this program.  Exit code 0 = every byte equal for every kernel variant.
 *
 * File layout (little endian): u32 W, H, count_vec4; packed[count_vec4][4] u32; constants[16] f32; positions[W*H][4] f32;
 * light_point[3] f32; mask_dir[W*H] u8 (the reference's directional light from the constants block); mask_point[W*H] u8. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rts.h"

#define CHECK(call) do { int st_ = (call); if (st_ != RTS_OK) { fprintf(stderr, "%s: %s\n", #call, rts_status_string(st_)); return 10; } } while (0)

static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: seam2_gpu <golden.bin>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    uint32_t hdr[3];
    if (read_exact(f, hdr, sizeof hdr)) return 3;
    const uint32_t W = hdr[0], H = hdr[1];
    const size_t count = hdr[2], pixels = (size_t)W * H;
    rts_vec4u* packed = (rts_vec4u*)malloc(count * sizeof(rts_vec4u));
    rts_constants k;
    float* positions = (float*)malloc(pixels * 16);
    float light_point[3];
    uint8_t* want_dir = (uint8_t*)malloc(pixels);
    uint8_t* want_point = (uint8_t*)malloc(pixels);
    uint8_t* got = (uint8_t*)malloc(pixels);
    if (!packed || !positions || !want_dir || !want_point || !got) return 3;
    if (read_exact(f, packed, count * sizeof(rts_vec4u)) || read_exact(f, &k, sizeof k) || read_exact(f, positions, pixels * 16) ||
        read_exact(f, light_point, sizeof light_point) || read_exact(f, want_dir, pixels) || read_exact(f, want_point, pixels)) return 3;
    fclose(f);

    int devices = 0;
    CHECK(rts_device_count(&devices));
    if (devices < 1) { fprintf(stderr, "no GPU\n"); return 4; }
    rts_ctx* ctx = NULL;
    CHECK(rts_ctx_create(0, &ctx));
    if (rts_trace_shadow_mask(ctx, &k, NULL, positions, W, H, 0, H, got) != RTS_ERR_NO_BVH) return 5;   /* no stream yet */
    CHECK(rts_ctx_set_bvh(ctx, packed, count));                      /* == Gfx_CreateBuffer(Storage, 16, count, data) */

    void *d_positions = NULL, *d_mask = NULL;
    CHECK(rts_device_malloc(ctx, &d_positions, pixels * 16));         /* binding 2: RGBA32F W x H */
    CHECK(rts_device_malloc(ctx, &d_mask, pixels));                  /* binding 3: R8 W x H */
    CHECK(rts_memcpy_h2d(ctx, d_positions, positions, pixels * 16));

    rts_light* point = (rts_light*)calloc(1, sizeof(rts_light));
    if (!point) return 3;
    point->type = RTS_LIGHT_POINT;
    point->nsamples = 1;
    memcpy(point->xyz, light_point, sizeof light_point);

    int kernels = 0;
    CHECK(rts_ctx_get_option(ctx, "kernel_count", &kernels));
    int failures = 0;
    for (int kernel = -1; kernel < kernels; ++kernel) {
        CHECK(rts_ctx_set_option(ctx, "kernel", kernel));
        for (int pass = 0; pass < 2; ++pass) {                       /* 0: the reference's light (constants), 1: point light */
            const rts_light* light = pass ? point : NULL;
            const uint8_t* want = pass ? want_point : want_dir;
            memset(got, 0xAB, pixels);
            CHECK(rts_memcpy_h2d(ctx, d_mask, got, pixels));
            CHECK(rts_timer_begin(ctx, NULL));
            CHECK(rts_trace_shadow_mask_device(ctx, &k, light, (const float*)d_positions, W, H, 0, H, (uint8_t*)d_mask, NULL));
            CHECK(rts_timer_end(ctx, NULL));                         /* == Gfx_BeginTimer / EndTimer(Timestamp_Shadows) */
            float ms = 0.0f;
            CHECK(rts_timer_elapsed_ms(ctx, &ms));
            CHECK(rts_memcpy_d2h(ctx, got, d_mask, pixels));
            size_t bad = 0;
            for (size_t i = 0; i < pixels; ++i) bad += got[i] != want[i];
            if (bad) { fprintf(stderr, "kernel %d light %d: %zu of %zu bytes differ\n", kernel, pass, bad, pixels); ++failures; }
            else printf("kernel %2d %-11s %s ok (%.3f ms)\n", kernel, pass ? "point" : "directional", rts_ctx_last_kernel_name(ctx), ms);
        }
    }
    /* the host-pointer form of the same dispatch, lower half of the frame only */
    CHECK(rts_ctx_set_option(ctx, "kernel", -1));
    memset(got, 0xAB, pixels);
    CHECK(rts_trace_shadow_mask(ctx, &k, NULL, positions, W, H, H / 2, H, got));
    for (size_t i = 0; i < pixels; ++i) {
        const int own = i >= (size_t)(H / 2) * W;
        if (got[i] != (own ? want_dir[i] : 0xAB)) { fprintf(stderr, "host stripe: byte %zu\n", i); ++failures; break; }
    }
    CHECK(rts_device_free(ctx, d_positions));
    CHECK(rts_device_free(ctx, d_mask));
    CHECK(rts_ctx_destroy(ctx));
    free(packed); free(positions); free(want_dir); free(want_point); free(got); free(point);
    return failures ? 1 : 0;
}
