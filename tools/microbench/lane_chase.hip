// Microbenchmark (GPU box): what does a LANE-PER-RAY iteration cost as a function of the vector loads in it?
//
//   hipcc --offload-arch=gfx950 -O2 -o lane_chase lane_chase.hip && ./lane_chase
//
// Every lane chases its own pointer chain through a table of NODES 128-byte records (80 MB: beyond L2, inside the Infinity
// Cache, like the node stream): the next index is read from the record just fetched, so an iteration is one dependent
// fetch.  Variants: how many 16-byte loads an iteration issues from its record (1, 2, 3 = the stackless step, 7 = a wide
// node), and whether the extra loads are real or masked out of range (buffer loads beyond num_records: "a lane with nothing
// to fetch").  Few waves (the tail of a frame) and many waves.  Output: microseconds per iteration.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, bool OOB>
__global__ __launch_bounds__(64) void chase(const void* table, unsigned bytes, int iters, unsigned long long* stamps, unsigned* sink) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00020000);
    unsigned idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % (bytes / 128u);
    unsigned acc = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        u32x4 q[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const unsigned off = (OOB && j > 0) ? 0xFFFFFF00u : idx * 128u + 16u * j;
            q[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
        }
#pragma unroll
        for (int j = 1; j < LOADS; ++j) acc += q[j].y;
        idx = q[0].x;
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = r0; stamps[blockIdx.x * 2 + 1] = r1; }
    sink[blockIdx.x * 64 + threadIdx.x] = acc + idx;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LOADS, bool OOB>
static void run(const void* d_table, unsigned bytes, unsigned waves, unsigned long long* d_stamps, unsigned* d_sink) {
    const int iters = 300;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((chase<LOADS, OOB>), dim3(waves), dim3(64), 0, nullptr, d_table, bytes, iters, d_stamps, d_sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(waves * 2);
    CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (unsigned w = 0; w < waves; ++w) sum += (double)(st[2 * w + 1] - st[2 * w]);
    printf("  %d load(s) per iteration%s, %5u waves: %.3f us per iteration\n", LOADS, OOB ? " (all but the first out of range)" : "", waves,
           sum / waves / iters / 100.0);
}

int main() {
    const unsigned nodes = 640 * 1024, bytes = nodes * 128u;       // 80 MB
    std::vector<unsigned> perm(nodes);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<unsigned> host((size_t)nodes * 32, 1u);
    for (unsigned i = 0; i < nodes; ++i) host[(size_t)perm[i] * 32] = perm[(i + 1) % nodes];     // one big cycle
    void* d_table; unsigned long long* d_stamps; unsigned* d_sink;
    CK(hipMalloc(&d_table, bytes)); CK(hipMemcpy(d_table, host.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_stamps, 8192 * 16)); CK(hipMalloc(&d_sink, 8192 * 64 * 4));
    for (unsigned waves : { 16u, 256u, 8192u }) {
        printf("%u waves in flight:\n", waves);
        run<1, false>(d_table, bytes, waves, d_stamps, d_sink);
        run<2, false>(d_table, bytes, waves, d_stamps, d_sink);
        run<3, false>(d_table, bytes, waves, d_stamps, d_sink);
        run<3, true>(d_table, bytes, waves, d_stamps, d_sink);
        run<7, false>(d_table, bytes, waves, d_stamps, d_sink);
        run<7, true>(d_table, bytes, waves, d_stamps, d_sink);
    }
    return 0;
}
