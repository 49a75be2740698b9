// Can a second kernel start while the first one of the SAME stream is still running?
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/any_order.hip -o tools/microbench/any_order && tools/microbench/any_order
// Kernel A: few long workgroups (spin `usA` microseconds on the 100 MHz counter); kernel B: many short ones.  Every workgroup
// stamps s_memrealtime at its start and end; the host prints, per mode, when B's first workgroup started relative to A's
// last end.  Modes: plain launches on one stream; B with hipExtAnyOrderLaunch on the same stream; B on a second stream
// (fork/join with events, as a library call that must leave everything on the caller's stream would do it).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ void spin(uint64_t* stamps, uint32_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const uint32_t nA = 64, nB = 4096, usA = 200, usB = 5;
    uint64_t *dA, *dB, *dC;
    CK(hipMalloc(&dA, nA * 16)); CK(hipMalloc(&dB, nB * 16)); CK(hipMalloc(&dC, nA * 16));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    std::vector<uint64_t> hA(2 * nA), hB(2 * nB), hC(2 * nA);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(t0, s1));
            if (mode == 2) {                     // B first, on the second stream, so that it is in flight when A starts
                CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
                hipLaunchKernelGGL(spin, dim3(nA), dim3(64), 0, s2, dA, usA * 100);
                hipLaunchKernelGGL(spin, dim3(nB), dim3(64), 0, s1, dB, usB * 100);
                CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0));
            } else {
                hipLaunchKernelGGL(spin, dim3(nA), dim3(64), 0, s1, dA, usA * 100);
                if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(nB), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, dB, usB * 100);
                else hipLaunchKernelGGL(spin, dim3(nB), dim3(64), 0, s1, dB, usB * 100);
            }
            hipLaunchKernelGGL(spin, dim3(nA), dim3(64), 0, s1, dC, 100);     // the "next frame": must start after A and B
            CK(hipEventRecord(t1, s1));
            CK(hipStreamSynchronize(s1));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            CK(hipMemcpy(hA.data(), dA, nA * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), dB, nB * 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hC.data(), dC, nA * 16, hipMemcpyDeviceToHost));
            uint64_t a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0, c0 = ~0ull;
            for (uint32_t i = 0; i < nA; ++i) { a0 = std::min(a0, hA[2 * i]); a1 = std::max(a1, hA[2 * i + 1]); c0 = std::min(c0, hC[2 * i]); }
            for (uint32_t i = 0; i < nB; ++i) { b0 = std::min(b0, hB[2 * i]); b1 = std::max(b1, hB[2 * i + 1]); }
            const char* names[3] = { "same stream, plain", "same stream, B any-order", "A on a second stream (fork/join events)" };
            printf("%-42s rep %d: A [0, %.1f] us  B [%.1f, %.1f] us  next kernel starts at %.1f us  events %.1f us  -> %s\n", names[mode], rep,
                   (a1 - a0) / 100.0, ((double)b0 - (double)a0) / 100.0, ((double)b1 - (double)a0) / 100.0, ((double)c0 - (double)a0) / 100.0, ms * 1000.0,
                   b0 < a1 ? "OVERLAP" : "serial");
        }
    }
    return 0;
}
