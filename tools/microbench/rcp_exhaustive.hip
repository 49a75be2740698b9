// Is v_rcp_f32 + one Newton step the correctly rounded 1.0f / x on gfx950?  All 2^32 inputs against the IEEE division.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
//         tools/microbench/rcp_exhaustive.hip -o tools/microbench/rcp_exhaustive && tools/microbench/rcp_exhaustive
// Prints, per class of input (|x| in the guarded range [2^-100, 2^100] or outside, mantissa all ones or not), how many inputs
// give a different bit pattern, with examples.  (Markstein: one Newton step from a 1-ulp estimate rounds correctly unless the
// divisor's mantissa is all ones; the range guard keeps every intermediate normal.)  Same for sqrt: v_sqrt_f32 + the usual
// two-fma correction against the IEEE sqrtf.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>

__device__ __forceinline__ float fastRcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float fastSqrt(float x) {              // Newton on s = sqrt(x) with the residual in one fma
    float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

struct Counts { unsigned long long bad[8]; uint32_t example[8]; };

__global__ void check(Counts* out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = t; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i;
        const float x = __uint_as_float(u);
        const float ax = __builtin_fabsf(x);
        const bool inRange = ax >= 7.888609052e-31f && ax <= 1.2676506e30f;          // 2^-100 .. 2^100
        const bool ones = (u & 0x7FFFFFu) == 0x7FFFFFu;
        {
            const float want = 1.0f / x, got = fastRcp(x);
            const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
            if (!same) { const int c = (inRange ? 0 : 2) + (ones ? 1 : 0); atomicAdd(&out->bad[c], 1ull); out->example[c] = u; }
        }
        if (!(u >> 31)) {
            const float want = __builtin_sqrtf(x), got = fastSqrt(x);
            const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
            const bool sIn = x >= 7.888609052e-31f && x <= 1.2676506e30f;
            if (!same) { const int c = 4 + (sIn ? 0 : 1); atomicAdd(&out->bad[c], 1ull); out->example[c] = u; }
        }
    }
}

int main() {
    Counts* d; Counts h{};
    if (hipMalloc(&d, sizeof(Counts)) != hipSuccess) return 1;
    (void)hipMemset(d, 0, sizeof(Counts));
    hipLaunchKernelGGL(check, dim3(256 * 32), dim3(256), 0, nullptr, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    (void)hipMemcpy(&h, d, sizeof(Counts), hipMemcpyDeviceToHost);
    const char* names[6] = { "rcp, |x| in [2^-100, 2^100], mantissa not all ones", "rcp, in range, mantissa all ones", "rcp, out of range",
                             "rcp, out of range, mantissa all ones", "sqrt, x in [2^-100, 2^100]", "sqrt, out of range" };
    for (int c = 0; c < 6; ++c) {
        float ex; uint32_t u = h.example[c]; memcpy(&ex, &u, 4);
        printf("%-52s: %llu inputs differ%s", names[c], h.bad[c], h.bad[c] ? "" : "\n");
        if (h.bad[c]) printf(" (e.g. 0x%08x = %g)\n", u, ex);
    }
    return 0;
}
