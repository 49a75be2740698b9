// Harness: G-buffer generation on the GPU -- the same closest-hit tests as the host version
// (rts_closest_hit.h), one primary ray per lane, one wave per 8x8 pixel tile, walked as a packet.  Replaces the reference's raster pass
// (Source/RayTracedShadows.cpp:512-568, Model.vert/.frag) as the producer of the position target the shadow kernel
// reads; not part of the timed shadow path.
#include <hip/hip_runtime.h>
#include "../../include/rts_scene.h"
#include "rts_closest_hit.h"

namespace rts_harness {
Camera makeCamera(const float eye[3], const float target[3], float fovy, uint32_t W, uint32_t H);

// One wave = one 8x8 tile of primary rays, walked as a PACKET: the rays of a tile are coherent, every index only grows, and
// every way out of a subtree leads to its root's miss link -- so the smallest index any ray stands on is wave-uniform
// (the argument of the shadow kernel): the node comes through the scalar cache once per wave, and each ray
// takes part only where it stands.  Every ray performs exactly the tests, in the order, that closestHit() performs for it
// alone (same helpers, no contraction), so the texels are the same bits as the host pass and the oracle's.
__global__ __launch_bounds__(64) void gbufferKernel(const uint32_t* __restrict__ bvh, Camera cam, uint32_t W, uint32_t H,
                                                    float* __restrict__ positions, float* __restrict__ normals) {
    const uint32_t END = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x;
    const uint32_t x = blockIdx.x * 8u + (lane & 7u), y = blockIdx.y * 8u + (lane >> 3);
    const bool live = x < W && y < H;
    const V3 d = primaryDirection(cam, live ? x : 0u, live ? y : 0u, W, H);
    const V3 o = cam.eye;
    const V3 inv{ 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    Hit best{ asFloat(0x7F800000u), END };
    uint32_t pos = live ? 0u : END;                                     // the node this ray tests next
    uint32_t cur = 0;
    while (cur != END) {
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        const uint32_t* a = bvh + (size_t)cur * 8;
        const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = a[4], b1 = a[5], b2 = a[6], b3 = a[7];
        const uint32_t na[4] = { a0, a1, a2, a3 }, nb[4] = { b0, b1, b2, b3 };
        const bool here = pos == cur;
        bool entered = false;
        if (a3 != END) {                                                // (wave-uniform branch)
            if (here) {
                leafTest(na, nb, bvh + (size_t)a3 * 4, cur, o, d, &best);
                pos = b3;
            }
        } else if (here) {
            entered = boxTest(na, nb, o, inv, best.t);
            pos = entered ? cur + 1u : b3;
        }
        cur = __builtin_amdgcn_ballot_w64(entered) != 0 ? cur + 1u : b3;   // nobody inside: everyone waits at or beyond the miss link
    }
    if (!live) return;
    const size_t i = ((size_t)y * W + x) * 4;
    writeTexel(bvh, d, best, positions + i, normals ? normals + i : nullptr);
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
    dim3 grid((W + 7) / 8, (H + 7) / 8), block(64);
    hipLaunchKernelGGL(rts_harness::gbufferKernel, grid, block, 0, (hipStream_t)stream, (const uint32_t*)bvh, cam, W, H,
                       d_positions, d_normals);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RTS_OK : RTS_ERR_HIP + (int)e;
}
