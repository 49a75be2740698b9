// Internal interface between the C-ABI layer (rts_api.cpp) and the HIP kernels (rts_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rts {

enum Variant {
    V_STRAIGHT = 0,    // loop as the shader spells it
    V_WHILEWHILE = 1,  // inner-node descent loop + batched leaf test
    V_POSTPONE = 2,    // leaves parked in registers, tested wave-wide
    V_PACKET = 3,      // wave walks the union of its rays' paths; nodes via scalar loads
    V_COUNT
};

// Kernel argument block (passed by value; lives in the kernarg segment, read with scalar loads).
struct TraceParams {
    const void* bvh;          // packed vec4 stream, SURVEY.md Appendix A (device)
    uint32_t bvhBytes;
    uint32_t bvhFinite;       // every float of the stream is finite -> FAST slab test is legal
    // mask dispatch
    const float4* positions;  // W x H RGBA32F, camera-relative (device)
    uint8_t* mask;            // W x H bytes (device)
    uint32_t W, H, rowBegin, rowEnd;
    uint32_t blocksX, blocksY, nBlocks, gridBlocks, swizzle;
    float cam[3];
    uint32_t lightType, nsamples;
    float light[3];
    // generic rays
    const void* rays;         // rts_ray[n] (device)
    uint8_t* out;
    uint64_t nrays;
    float offsets[64][4];
};

const char* kernelName(int variant, bool mask);
hipError_t launchShadowMask(int variant, const TraceParams& p, hipStream_t stream);
hipError_t launchTraceRays(int variant, const TraceParams& p, hipStream_t stream);

} // namespace rts
