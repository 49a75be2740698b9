// Microbenchmark (GPU box): where should the hot near-root nodes of the WIDE walk come from?  (north_star: "LDS staging of
// hot near-root nodes"; SURVEY.md Appendix C sizes the treelet: the top levels, ~127 KB.)
//
//   hipcc --offload-arch=gfx950 -O2 -o node_fetch node_fetch.hip && ./node_fetch
//
// One "step" = what the wide packet kernel does per node: fetch a 128-byte wide node that every lane of the wave needs
// (wave-uniform address), run the four cheap slab tests (40 VALU, rts_kernels.hip: cheapBox), derive the next node's index
// from the outcome (a dependent chain, as in the walk).  The node table holds K nodes; the walk hops pseudo-randomly in it.
// Three fetch paths, same arithmetic:
//   scalar   two s_load_dwordx16 through the scalar cache; planes are SGPR operands of the VALU instructions (what the
//            shipped kernel does); one-wave workgroups, 8 waves per SIMD
//   lds      the K nodes are STAGED IN LDS by the workgroup (16 waves, 128 KB for K = 1024: one workgroup per CU, 4 waves
//            per SIMD); a step reads its node with seven ds_read_b128 (every lane the same address: a broadcast), planes
//            are VGPR operands
//   vector   seven global_load_dwordx4 from the wave-uniform address (vector L1 instead of the scalar cache), planes in
//            VGPRs; one-wave workgroups, 8 waves per SIMD (the registers for the node fit)
// Table sizes: 64 nodes (8 KB: always in the 16 KB scalar cache), 1024 nodes (128 KB: the treelet of Appendix C; beyond the
// scalar cache, inside L2), 65536 nodes (8 MB: beyond one XCD's L2).  `lds` only exists for K <= 1024.
// Output: shader clocks per step per wave, and steps per microsecond chip-wide (from the realtime span of the launch).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) u32x16* ConstWide;

struct RayC { float ix, iy, iz, ux, uy, uz, dx, dy, dz; };

__device__ __forceinline__ bool cheapBox(float lx, float ly, float lz, float hx, float hy, float hz, const RayC& r) {
    const float fx = __builtin_fmaf(hx, r.ix, -r.ux), fy = __builtin_fmaf(hy, r.iy, -r.uy), fz = __builtin_fmaf(hz, r.iz, -r.uz);
    const float nx = __builtin_fmaf(lx, r.ix, -r.dx), ny = __builtin_fmaf(ly, r.iy, -r.dy), nz = __builtin_fmaf(lz, r.iz, -r.dz);
    const float t1 = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), 0.0f);
    return t1 >= t0;
}

__device__ __forceinline__ RayC makeRay() {
    const float l = (float)(threadIdx.x & 63);
    RayC r;
    r.ix = 1.0f + l * 0.01f; r.iy = 1.3f + l * 0.02f; r.iz = 0.9f + l * 0.015f;
    r.ux = 0.2f * r.ix - 1e-6f; r.uy = 0.3f * r.iy - 1e-6f; r.uz = 0.1f * r.iz - 1e-6f;
    r.dx = 0.2f * r.ix + 1e-6f; r.dy = 0.3f * r.iy + 1e-6f; r.dz = 0.1f * r.iz + 1e-6f;
    return r;
}

enum { SCALAR = 0, LDS = 1, VECTOR = 2 };

template <int MODE>
__global__ void walk(const unsigned* table, unsigned K, int iters, unsigned long long* stamps, unsigned* sink) {
    extern __shared__ unsigned lds[];                    // LDS mode: the K nodes (32 dwords each)
    const RayC r = makeRay();
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (MODE == LDS) {
        for (unsigned i = threadIdx.x; i < K * 8; i += blockDim.x)          // staging: 16 bytes per thread per turn
            ((u32x4*)lds)[i] = ((const u32x4*)table)[i];
        __syncthreads();
    }
    unsigned idx = (unsigned)__builtin_amdgcn_readfirstlane((int)((wave * 2654435761u) & (K - 1)));
    unsigned acc = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        float p[24];
        if (MODE == SCALAR) {
            const ConstWide np = (ConstWide)(table + (size_t)idx * 32);
            const u32x16 a = np[0], b = np[1];
#pragma unroll
            for (int d = 0; d < 16; ++d) p[d] = __uint_as_float(a[d]);
#pragma unroll
            for (int d = 0; d < 8; ++d) p[16 + d] = __uint_as_float(b[d]);
        } else {
            u32x4 q[6];
            if (MODE == LDS) {
#pragma unroll
                for (int j = 0; j < 6; ++j) q[j] = ((const u32x4*)lds)[idx * 8 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) q[j] = ((const u32x4*)table)[(size_t)idx * 8 + j];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                p[4 * j] = __uint_as_float(q[j].x); p[4 * j + 1] = __uint_as_float(q[j].y);
                p[4 * j + 2] = __uint_as_float(q[j].z); p[4 * j + 3] = __uint_as_float(q[j].w);
            }
        }
        unsigned h = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            h |= (__builtin_amdgcn_ballot_w64(cheapBox(p[6 * k], p[6 * k + 1], p[6 * k + 2], p[6 * k + 3], p[6 * k + 4], p[6 * k + 5], r)) != 0) ? (1u << k) : 0u;
        acc += h;
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)((idx * 1664525u + 1013904223u + h * 97u) & (K - 1)));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        stamps[wave * 4] = c0; stamps[wave * 4 + 1] = c1; stamps[wave * 4 + 2] = r0; stamps[wave * 4 + 3] = r1;
        sink[wave] = acc;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
static void run(const char* name, const unsigned* d_table, unsigned K, int iters, unsigned long long* d_stamps, unsigned* d_sink) {
    const unsigned cus = 256;
    unsigned threads, blocks;
    size_t shmem;
    if (MODE == LDS) { threads = 1024; blocks = cus * 8; shmem = (size_t)K * 128; }        // 8 rounds of one workgroup per CU
    else { threads = 64; blocks = cus * 32 * 8; shmem = 0; }                              // 8 rounds of 32 one-wave workgroups per CU
    const unsigned waves = blocks * (threads / 64);
    if (MODE == LDS) CK(hipFuncSetAttribute((const void*)walk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    for (int warm = 0; warm < 40; ++warm) hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(threads), shmem, nullptr, d_table, K, iters, d_stamps, d_sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(threads), shmem, nullptr, d_table, K, iters, d_stamps, d_sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)waves * 4);
    CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double clk = 0, rt = 0;
    unsigned long long first = ~0ull, last = 0;
    for (unsigned w = 0; w < waves; ++w) {
        clk += (double)(st[w * 4 + 1] - st[w * 4]); rt += (double)(st[w * 4 + 3] - st[w * 4 + 2]);
        first = std::min(first, st[w * 4 + 2]); last = std::max(last, st[w * 4 + 3]);
    }
    const double mhz = clk / rt * 100.0, spanUs = (double)(last - first) / 100.0;
    printf("%-7s K=%6u (%8.0f KB): %7.1f clk per step per wave, %8.1f steps/us chip-wide (%u waves x %d steps in %.1f us, %.0f MHz%s)\n",
           name, K, K * 128.0 / 1024.0, clk / waves / iters, (double)waves * iters / spanUs, waves, iters, spanUs, mhz,
           MODE == LDS ? "; staging included in the span, not in clk per step" : "");
}

int main() {
    const unsigned KMAX = 65536;
    std::vector<unsigned> host((size_t)KMAX * 32);
    unsigned seed = 12345;
    for (size_t n = 0; n < KMAX; ++n)
        for (int k = 0; k < 4; ++k) {
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                seed = seed * 1664525u + 1013904223u;
                lo[a] = (float)(seed >> 8) / 16777216.0f;
                seed = seed * 1664525u + 1013904223u;
                hi[a] = lo[a] + (float)(seed >> 8) / 16777216.0f * 0.7f;
            }
            for (int a = 0; a < 3; ++a) { memcpy(&host[n * 32 + 6 * k + a], &lo[a], 4); memcpy(&host[n * 32 + 6 * k + 3 + a], &hi[a], 4); }
        }
    unsigned* d_table; unsigned long long* d_stamps; unsigned* d_sink;
    CK(hipMalloc(&d_table, host.size() * 4));
    CK(hipMemcpy(d_table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const size_t maxWaves = 256 * 32 * 8;
    CK(hipMalloc(&d_stamps, maxWaves * 32)); CK(hipMalloc(&d_sink, maxWaves * 4));
    const int iters = 400;
    for (unsigned K : { 64u, 1024u, 65536u }) {
        run<SCALAR>("scalar", d_table, K, iters, d_stamps, d_sink);
        if (K <= 1024) run<LDS>("lds", d_table, K, iters, d_stamps, d_sink);
        run<VECTOR>("vector", d_table, K, iters, d_stamps, d_sink);
    }
    return 0;
}
