// Microbenchmark (GPU box): sustained wave64 instruction issue per SIMD on gfx950, measured in SHADER CLOCKS.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
//
// Method (round 2; replaces the wall-clock version that assumed 2.4 GHz):
//   * every wave stamps s_memtime (shader cycles) and s_memrealtime (the chip-wide 100 MHz counter) around its loop.
//     clock held = sum(shader cycles) / sum(realtime ticks) x 100 MHz; span of the launch = last end - first start on the
//     realtime counter, converted to shader cycles with that clock;
//       instructions per clock per SIMD = waves x iterations x instructions per iteration / (span cycles x SIMDs)
//     so the chip-wide rate is measured, whatever the placement of the waves, and no clock frequency is assumed;
//   * the grid holds 16 x as many one-wave workgroups as fit at once (like the real launch, slots are refilled as waves
//     end); occupancy is set with dynamic LDS: 160 KB per CU / (4 x waves per SIMD) workgroups;
//   * 1.5 s of back-to-back launches of the same kernel before the measured one (clock and power state settled).
// Modes:
//   fma      16 independent v_fma_f32 per iteration                     -- calibration: the guide's 0.5 / clk / SIMD
//   slab     the packet loop's ordered slab test, 16 VALU (6 v_sub with an SGPR operand, 6 v_mul, v_min3, v_max,
//            v_max3, v_cmp_ge into an SGPR pair), no scalar work
//   step     slab + the scalar side of a down-step of the real loop: s_add, s_load_dwordx8 of a 32-byte node (always
//            the same few nodes: scalar-cache hits), s_waitcnt, s_cmp, s_cbranch, s_andn2, s_cbranch (7 scalar)
//   stepmiss step with the node address striding through a 64 MB table (scalar-cache and mostly L2 misses)
//   pkfma    16 independent v_pk_fma_f32 (two fp32 FMAs per lane and instruction) on VGPR pairs   (round 3: could the wide
//   pkfmas   the same with an SGPR pair as first source and the second one broadcast (op_sel)      node test be packed?)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define SLAB16(O, I)                                                                                                     \
    "v_sub_f32 %[t0], s44, %[" O "x]\n\t v_sub_f32 %[t1], s45, %[" O "y]\n\t v_sub_f32 %[t2], s46, %[" O "z]\n\t"        \
    "v_sub_f32 %[t3], s40, %[" O "x]\n\t v_sub_f32 %[t4], s41, %[" O "y]\n\t v_sub_f32 %[t5], s42, %[" O "z]\n\t"        \
    "v_mul_f32 %[t0], %[t0], %[" I "x]\n\t v_mul_f32 %[t1], %[t1], %[" I "y]\n\t v_mul_f32 %[t2], %[t2], %[" I "z]\n\t"  \
    "v_mul_f32 %[t3], %[t3], %[" I "x]\n\t v_mul_f32 %[t4], %[t4], %[" I "y]\n\t v_mul_f32 %[t5], %[t5], %[" I "z]\n\t"  \
    "v_min3_f32 %[t0], %[t0], %[t1], %[t2]\n\t v_max_f32 %[t3], %[t3], %[t4]\n\t v_max3_f32 %[t3], %[t3], %[t5], 0\n\t"  \
    "v_cmp_ge_f32 s[54:55], %[t0], %[t3]\n\t"

enum { FMA = 0, SLAB = 1, STEP = 2, STEPMISS = 3, PKFMA = 4, PKFMAS = 5 };

template <int MODE>
__global__ __launch_bounds__(64) void spin(const void* nodes, unsigned mask, int iters, unsigned long long* stamps, float* sink) {
    extern __shared__ unsigned char occupancyPad[];     // never touched: sets the number of workgroups per CU
    float ox = threadIdx.x * 1e-3f + 0.1f, oy = ox + 0.2f, oz = ox + 0.3f, ix = 1.5f, iy = 2.5f, iz = 3.5f;
    float t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    float a[16];
    for (int k = 0; k < 16; ++k) a[k] = ox + k;
    double b[16];                                         // (64-bit VGPR pairs for the packed forms)
    for (int k = 0; k < 16; ++k) b[k] = (double)ox + k;
    double ixy = (double)ix;
    unsigned long long members = ~0ull;
    unsigned off = (blockIdx.x * 2654435761u) & mask & ~31u;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == FMA) {
            asm volatile(
                "v_fma_f32 %0, %0, %16, %0\n\t v_fma_f32 %1, %1, %16, %1\n\t v_fma_f32 %2, %2, %16, %2\n\t v_fma_f32 %3, %3, %16, %3\n\t"
                "v_fma_f32 %4, %4, %16, %4\n\t v_fma_f32 %5, %5, %16, %5\n\t v_fma_f32 %6, %6, %16, %6\n\t v_fma_f32 %7, %7, %16, %7\n\t"
                "v_fma_f32 %8, %8, %16, %8\n\t v_fma_f32 %9, %9, %16, %9\n\t v_fma_f32 %10, %10, %16, %10\n\t v_fma_f32 %11, %11, %16, %11\n\t"
                "v_fma_f32 %12, %12, %16, %12\n\t v_fma_f32 %13, %13, %16, %13\n\t v_fma_f32 %14, %14, %16, %14\n\t v_fma_f32 %15, %15, %16, %15\n\t"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                  "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                : "v"(ix));
        } else if (MODE == PKFMA || MODE == PKFMAS) {
#define PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %16, %" #n "\n\t"
#define PKS(n) "v_pk_fma_f32 %" #n ", s[40:41], %16, %" #n " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
            if (MODE == PKFMA)
                asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(8) PK(9) PK(10) PK(11) PK(12) PK(13) PK(14) PK(15)
                             : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]),
                               "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15])
                             : "v"(ixy));
            else
                asm volatile("s_mov_b32 s40, 0x3f000000\n\t s_mov_b32 s41, 0x3f400000\n\t"
                             PKS(0) PKS(1) PKS(2) PKS(3) PKS(4) PKS(5) PKS(6) PKS(7) PKS(8) PKS(9) PKS(10) PKS(11) PKS(12) PKS(13) PKS(14) PKS(15)
                             : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]),
                               "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15])
                             : "v"(ixy)
                             : "s40", "s41");
        } else if (MODE == SLAB) {
            asm volatile(SLAB16("o", "i")
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5)
                         : [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz)
                         : "s40", "s41", "s42", "s44", "s45", "s46", "s54", "s55");
        } else {
            // one down-step of the packet loop (tools/gen_packet_asm.py, loop_leaf): node fetch, leaf check, slab test,
            // "nobody leaves" check.  The node never is a leaf and every lane always hits (boxes of +-1e30).
            asm volatile(
                "s_add_u32 %[off], %[off], %[stride]\n\t"
                "s_and_b32 %[off], %[off], %[mask]\n\t"
                "s_load_dwordx8 s[40:47], %[base], %[off]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_cmp_lg_u32 s43, -1\n\t"
                "s_cbranch_scc1 9f\n\t"
                SLAB16("o", "i")
                "s_andn2_b64 s[50:51], %[m], s[54:55]\n\t"
                "s_cbranch_scc0 9f\n\t"
                "s_mov_b64 %[m], -1\n\t"
                "9:\n\t"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [off] "+s"(off), [m] "+s"(members)
                : [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [base] "s"(nodes),
                  [stride] "s"(MODE == STEPMISS ? 0x9E3780u : 32u), [mask] "s"(mask & ~31u)
                : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s50", "s51", "s54", "s55", "scc");
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 3] = c1 - c0; stamps[blockIdx.x * 3 + 1] = r0; stamps[blockIdx.x * 3 + 2] = r1; }
    float s = t0 + t3 + (float)members;
    for (int k = 0; k < 16; ++k) s += a[k] + (float)b[k];
    if (s == 123.456f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, int wavesPerSimd, int valuPerIter, int scalarPerIter, const void* nodes, unsigned mask,
                unsigned long long* d_stamps, float* d_sink, int cus) {
    const int over = 16;
    const int waves = cus * 4 * wavesPerSimd * over;
    const int iters = (MODE == STEPMISS ? 4000 : 40000) / wavesPerSimd / over;
    // occupancy limit through LDS: 4 x wavesPerSimd workgroups per CU (8 per SIMD is the hardware's own limit)
    const size_t lds = wavesPerSimd >= 8 ? 0 : (size_t)(160 * 1024) / (4 * wavesPerSimd + 1) + 1024;
    hipFuncSetAttribute((const void*)spin<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    float ms = 0;
    do {                                                        // warm-up: at least 1.5 s of the same work
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(spin<MODE>, dim3(waves), dim3(64), lds, 0, nodes, mask, iters, d_stamps, d_sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 1500.f);
    hipLaunchKernelGGL(spin<MODE>, dim3(waves), dim3(64), lds, 0, nodes, mask, iters, d_stamps, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st((size_t)waves * 3);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> per(waves);
    for (int w = 0; w < waves; ++w) {
        cyc += (double)st[3 * w]; ticks += (double)(st[3 * w + 2] - st[3 * w + 1]);
        first = std::min(first, st[3 * w + 1]); last = std::max(last, st[3 * w + 2]);
        per[w] = (double)st[3 * w] / iters;
    }
    std::sort(per.begin(), per.end());
    const double mhz = cyc / ticks * 100.0;
    const double spanCycles = (double)(last - first) * mhz / 100.0;
    const double inFlight = ticks / (double)(last - first);
    const double simds = cus * 4.0;
    printf("%-9s %d waves/SIMD: %6.1f clk per iteration per wave (p5 %.1f p95 %.1f); waves in flight %.0f of %d; "
           "VALU %.3f / clk / SIMD, all instr %.3f / clk / SIMD; clock %.0f MHz; span %.0f us\n",
           name, wavesPerSimd, per[waves / 2], per[waves / 20], per[waves - 1 - waves / 20], inFlight, cus * 4 * wavesPerSimd,
           (double)waves * valuPerIter * iters / (spanCycles * simds),
           (double)waves * (valuPerIter + scalarPerIter + 3) * iters / (spanCycles * simds), mhz, (double)(last - first) / 100.0);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t bytes = 64u << 20;
    std::vector<unsigned> host(bytes / 4);
    for (size_t n = 0; n < bytes / 32; ++n) {                    // inner nodes whose box every ray hits
        float lo = -1e30f, hi = 1e30f;
        unsigned* w = &host[n * 8];
        for (int k = 0; k < 3; ++k) { memcpy(&w[k], &lo, 4); memcpy(&w[4 + k], &hi, 4); }
        w[3] = 0xFFFFFFFFu; w[7] = 0xFFFFFFFFu;
    }
    void* d_nodes; hipMalloc(&d_nodes, bytes); hipMemcpy(d_nodes, host.data(), bytes, hipMemcpyHostToDevice);
    unsigned long long* d_stamps; hipMalloc(&d_stamps, (size_t)cus * 4 * 8 * 16 * 3 * 8);
    float* d_sink; hipMalloc(&d_sink, 64);
    printf("%s, %d CUs\n", prop.name, cus);
    for (int w : {1, 2, 4, 8}) run<FMA>("fma", w, 16, 0, d_nodes, 0, d_stamps, d_sink, cus);
    for (int w : {1, 4, 8}) run<PKFMA>("pkfma", w, 16, 0, d_nodes, 0, d_stamps, d_sink, cus);
    for (int w : {1, 4, 8}) run<PKFMAS>("pkfmas", w, 16, 2, d_nodes, 0, d_stamps, d_sink, cus);
    for (int w : {1, 2, 4, 8}) run<SLAB>("slab", w, 16, 0, d_nodes, 0, d_stamps, d_sink, cus);
    for (int w : {1, 2, 4, 8}) run<STEP>("step", w, 16, 9, d_nodes, 1023u, d_stamps, d_sink, cus);          // 1 KB of nodes: K$ hits
    for (int w : {4, 8}) run<STEPMISS>("stepmiss", w, 16, 9, d_nodes, (unsigned)(bytes - 1), d_stamps, d_sink, cus);
    return 0;
}
