// Microbenchmark (GPU box): sustained wave64 instruction issue per SIMD on gfx950 for the instruction mix of the
// packet loop.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 16 independent v_mul/v_sub/v_min3 (no memory).  MODE 1: the same + 9 SALU + 1 taken branch per 16 VALU.
template <int MODE>
__global__ __launch_bounds__(64) void spin(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f, e = a * 0.3f, f = a * 0.7f;
    unsigned s0 = blockIdx.x, s1 = 3, s2 = 5;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_sub_f32 %0, 0x3f800000, %0\n\t v_sub_f32 %1, 0x3f800000, %1\n\t v_sub_f32 %2, 0x3f800000, %2\n\t"
            "v_sub_f32 %3, 0x3f800000, %3\n\t v_sub_f32 %4, 0x3f800000, %4\n\t v_sub_f32 %5, 0x3f800000, %5\n\t"
            "v_mul_f32 %0, %0, %1\n\t v_mul_f32 %2, %2, %3\n\t v_mul_f32 %4, %4, %5\n\t"
            "v_mul_f32 %1, %1, %2\n\t v_mul_f32 %3, %3, %4\n\t v_mul_f32 %5, %5, %0\n\t"
            "v_min3_f32 %0, %0, %1, %2\n\t v_max_f32 %3, %3, %4\n\t v_max3_f32 %3, %3, %5, 0\n\t"
            "v_cmp_ge_f32 vcc, %0, %3\n\t"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "vcc");
        if (MODE == 1)
            asm volatile(
                "s_add_u32 %0, %0, 32\n\t s_and_b64 vcc, vcc, exec\n\t s_cmp_lg_u32 %1, -1\n\t s_add_u32 %1, %1, 1\n\t"
                "s_andn2_b64 vcc, exec, vcc\n\t s_lshl_b32 %2, %0, 5\n\t s_sub_u32 %2, %2, 1\n\t s_cmp_eq_u32 %2, -1\n\t"
                "s_and_b64 vcc, vcc, exec\n\t"
                : "+s"(s0), "+s"(s1), "+s"(s2) : : "vcc", "scc");
    }
    if (a + b + c + d + e + f + (float)(s0 + s1 + s2) == 123.456f) out[0] = a;
}

// MODE 2: the same slab test with packed f32: 4 v_pk_add + 4 v_pk_mul instead of 6 v_sub + 6 v_mul (12 VALU / iter).
__global__ __launch_bounds__(64) void spinPk(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = { threadIdx.x * 1e-3f + 1.0f, 2.0f }, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f, o = a * 0.3f, iv = a * 0.7f;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t v_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t v_pk_add_f32 %3, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 %0, %0, %5\n\t v_pk_mul_f32 %1, %1, %5\n\t v_pk_mul_f32 %2, %2, %5\n\t v_pk_mul_f32 %3, %3, %5\n\t"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(o), "v"(iv));
        asm volatile(
            "v_min3_f32 %0, %0, %1, %2\n\t v_max_f32 %3, %3, %4\n\t v_max3_f32 %3, %3, %5, 0\n\t v_cmp_ge_f32 vcc, %0, %3\n\t"
            : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(c.x), "+v"(c.y), "+v"(d.x) : : "vcc");
    }
    if (a.x + b.x + c.x + d.x + a.y == 123.456f) out[0] = a.x;
}

static void runPk(int wavesPerSimd, float* d_out) {
    const int iters = 20000, waves = 256 * 4 * wavesPerSimd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spinPk, dim3(waves), dim3(64), 0, 0, d_out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spinPk, dim3(waves), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double simdCycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("8 v_pk + 4 VALU / iter        %d waves/SIMD: %.3f ms  -> %.3f slab tests per clk per SIMD (16-VALU form: see above x 1/16)\n",
           wavesPerSimd, ms, (double)iters * waves / simdCycles);
}

template <int MODE>
static void run(const char* name, int wavesPerSimd, float* d_out) {
    const int iters = 20000, waves = 256 * 4 * wavesPerSimd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<MODE>, dim3(waves), dim3(64), 0, 0, d_out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<MODE>, dim3(waves), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double valu = 16.0 * iters * waves, other = (MODE == 1 ? 9.0 : 0.0) * iters * waves + 3.0 * iters * waves;  // + loop ctrl
    const double simdCycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-28s %d waves/SIMD: %.3f ms  VALU %.3f per clk per SIMD (at 2.4 GHz), all instr %.3f per clk per SIMD\n", name,
           wavesPerSimd, ms, valu / simdCycles, (valu + other) / simdCycles);
}

int main() {
    float* d_out; hipMalloc(&d_out, 64);
    for (int w : {1, 2, 4, 8}) run<0>("16 VALU / iter", w, d_out);
    for (int w : {1, 2, 4, 8}) run<1>("16 VALU + 9 SALU / iter", w, d_out);
    for (int w : {2, 4, 8}) runPk(w, d_out);
    return 0;
}
