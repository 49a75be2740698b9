// Probe (GPU box): which XCD (XCC) a workgroup lands on, and whether two waves on the same XCC see each other's plain
// stores through L2-scope atomics (no sc1: the atomic executes in that XCC's L2) -- the coherence a per-XCD hand-over
// queue would rely on.  Also times same-address atomics at L2 scope against agent scope.
// Build: hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip ; run: ./xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xccId() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x;
}

// Read a 64-bit word where this XCC's L2 holds it: a compare-and-swap that can never succeed.  (An atomic add/or of 0
// is folded into a plain atomic load by the compiler, and a workgroup-scope load may be served from the CU's own L1.)
__device__ __forceinline__ unsigned long long readAtL2(unsigned long long* p) {
    unsigned long long expected = 0xFFFFFFFFFFFFFFFEull;
    __hip_atomic_compare_exchange_strong(p, &expected, 0xFFFFFFFFFFFFFFFEull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    return expected;
}

__global__ __launch_bounds__(64) void whereAmI(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xccId();
}

// every wave: one fetch_add on counter[xcc] (L2 scope) or on counter[0] (agent scope)
template <int AGENT>
__global__ __launch_bounds__(64) void hammer(unsigned long long* counters, int reps) {
    if (threadIdx.x != 0) return;
    unsigned long long* c = AGENT ? counters : counters + 16 * (xccId() & 7u);
    for (int i = 0; i < reps; ++i) {
        if (AGENT) __hip_atomic_fetch_add(c, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(c, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// mailbox per XCC: the first wave to arrive on an XCC (ticket 0) publishes payload then flag with plain stores
// (s_waitcnt between them); every other wave of that XCC polls the flag with an L2 atomic (bounded) and then reads the
// payload with an L2 atomic.  ok[blockIdx] = 1 if it saw the payload, 2 if it gave up.
__global__ __launch_bounds__(64) void mailbox(unsigned long long* box, unsigned* ok) {
    if (threadIdx.x != 0) return;
    const unsigned x = xccId() & 7u;
    unsigned long long* b = box + 32 * x;            // [0] arrival counter, [8] payload, [16] flag (separate 64-B lines)
    const unsigned long long mine = __hip_atomic_fetch_add(b, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (mine == 0) {
        for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);          // let the readers start polling first
        b[8] = 0xC0FFEE00ull + x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                // s_waitcnt vmcnt(0)
        b[16] = 1;
        ok[blockIdx.x] = 3;
        return;
    }
    unsigned spins = 0;
    while (readAtL2(b + 16) != 1) {
        if (++spins > (1u << 16)) { ok[blockIdx.x] = 2; return; }
        __builtin_amdgcn_s_sleep(4);
    }
    const unsigned long long p = readAtL2(b + 8);
    ok[blockIdx.x] = p == 0xC0FFEE00ull + x ? 1 : 4;
}

int main() {
    const int N = 4096;
    unsigned* d; hipMalloc(&d, N * 4);
    hipLaunchKernelGGL(whereAmI, dim3(N), dim3(64), 0, 0, d);
    std::vector<unsigned> h(N); hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0}, rr = 0;
    for (int i = 0; i < N; ++i) { hist[h[i] & 15]++; rr += ((h[i] & 15) == (unsigned)(i % 8)); }
    printf("raw XCC_ID register of block 0..15:");
    for (int i = 0; i < 16; ++i) printf(" 0x%x", h[i]);
    printf("\nhistogram of (reg & 15):");
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\nblocks with (reg & 15) == blockIdx %% 8: %d of %d\n", rr, N);

    unsigned long long* c; hipMalloc(&c, 8 * 16 * 8 + 128); hipMemset(c, 0, 8 * 16 * 8 + 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int agent = 0; agent < 2; ++agent) {
        for (int waves : {8192, 65536}) {
            hipMemset(c, 0, 8 * 16 * 8);
            hipEventRecord(e0);
            if (agent) hipLaunchKernelGGL(hammer<1>, dim3(waves), dim3(64), 0, 0, c, 4);
            else hipLaunchKernelGGL(hammer<0>, dim3(waves), dim3(64), 0, 0, c, 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[128]; hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
            unsigned long long sum = 0; for (int i = 0; i < 8; ++i) sum += hc[16 * i];
            printf("%s scope: %d waves x 4 fetch_add on %s: %.3f ms = %.1f ns per atomic (all), counted %llu of %llu\n",
                   agent ? "agent" : "L2 (workgroup)", waves, agent ? "one word" : "one word per XCC", ms,
                   ms * 1e6 / (waves * 4.0), sum, (unsigned long long)waves * 4);
        }
    }
    unsigned long long* box; hipMalloc(&box, 8 * 32 * 8); hipMemset(box, 0, 8 * 32 * 8);
    unsigned* ok; hipMalloc(&ok, 2048 * 4); hipMemset(ok, 0, 2048 * 4);
    hipLaunchKernelGGL(mailbox, dim3(2048), dim3(64), 0, 0, box, ok);
    std::vector<unsigned> o(2048); hipMemcpy(o.data(), ok, 2048 * 4, hipMemcpyDeviceToHost);
    int cnt[5] = {0}; for (unsigned v : o) cnt[v < 5 ? v : 0]++;
    printf("mailbox through L2-scope atomics: %d readers saw the payload, %d gave up, %d saw a wrong payload, %d writers, %d unset\n",
           cnt[1], cnt[2], cnt[4], cnt[3], cnt[0]);
    return 0;
}
