// Harness: G-buffer generation on the GPU -- the same closest-hit code as the host version
// (rts_closest_hit.h), one primary ray per lane, 8x8 pixel tiles.  Replaces the reference's raster pass
// (Source/RayTracedShadows.cpp:512-568, Model.vert/.frag) as the producer of the position target the shadow kernel
// reads; not part of the timed shadow path.
#include <hip/hip_runtime.h>
#include "../../include/rts_scene.h"
#include "rts_closest_hit.h"

namespace rts_harness {
Camera makeCamera(const float eye[3], const float target[3], float fovy, uint32_t W, uint32_t H);

__global__ __launch_bounds__(256) void gbufferKernel(const uint32_t* bvh, Camera cam, uint32_t W, uint32_t H,
                                                     float* positions, float* normals) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    if (x >= W || y >= H) return;
    const size_t i = ((size_t)y * W + x) * 4;
    shadePixel(bvh, cam, x, y, W, H, positions + i, normals ? normals + i : nullptr);
}
int makeCombineParams(const rts_constants* k, const rts_light* light, bool havePositions, CombineParams* out);

// Combine pass (Combine.frag:18-37), one pixel per lane; rgb is 3 bytes per pixel like the host version.
__global__ __launch_bounds__(256) void combineKernel(CombineParams c, const float* positions, const float* normals,
                                                     const uint8_t* mask, uint64_t n, uint8_t* rgb) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float zero[4] = { 0.f, 0.f, 0.f, 0.f };
    const uint8_t q = combinePixel(c, positions ? positions + i * 4 : zero, normals + i * 4, mask[i]);
    rgb[i * 3] = q; rgb[i * 3 + 1] = q; rgb[i * 3 + 2] = q;
}
} // namespace rts_harness

extern "C" const void* rts_ctx_device_bvh(rts_ctx* ctx);   // rts_api.cpp
extern "C" int rts_ctx_device_ordinal(rts_ctx* ctx);

extern "C" int rtsh_combine_device(rts_ctx* ctx, const rts_constants* k, const rts_light* light, const float* d_positions,
                                   const float* d_normals, const uint8_t* d_mask, uint32_t W, uint32_t H, uint8_t* d_rgb,
                                   void* stream) {
    if (!ctx || !k || !d_normals || !d_mask || !d_rgb || W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    rts_harness::CombineParams c;
    int s = rts_harness::makeCombineParams(k, light, d_positions != nullptr, &c);
    if (s != RTS_OK) return s;
    hipError_t e = hipSetDevice(rts_ctx_device_ordinal(ctx));
    if (e != hipSuccess) return RTS_ERR_HIP + (int)e;
    const uint64_t n = (uint64_t)W * H;
    hipLaunchKernelGGL(rts_harness::combineKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c,
                       d_positions, d_normals, d_mask, n, d_rgb);
    e = hipGetLastError();
    return e == hipSuccess ? RTS_OK : RTS_ERR_HIP + (int)e;
}

extern "C" int rtsh_primary_gbuffer_device(rts_ctx* ctx, const float eye[3], const float target[3], float fovy,
                                           uint32_t W, uint32_t H, float* d_positions, float* d_normals, void* stream) {
    if (!ctx || !eye || !target || !d_positions || W == 0 || H == 0) return RTS_ERR_INVALID_ARG;
    const void* bvh = rts_ctx_device_bvh(ctx);
    if (!bvh) return RTS_ERR_NO_BVH;
    const rts_harness::Camera cam = rts_harness::makeCamera(eye, target, fovy, W, H);
    dim3 grid((W + 15) / 16, (H + 15) / 16), block(256);
    hipLaunchKernelGGL(rts_harness::gbufferKernel, grid, block, 0, (hipStream_t)stream, (const uint32_t*)bvh, cam, W, H,
                       d_positions, d_normals);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RTS_OK : RTS_ERR_HIP + (int)e;
}
