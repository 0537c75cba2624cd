// Developer probe (GPU box): throughput of returning device-scope atomicAdd on ONE address from many single-wave
// workgroups spread over the 8 XCDs (the persistent search kernel's work counter).
// hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void take(unsigned* counter, unsigned total, unsigned chunk, unsigned* sink, int spin) {
    unsigned n = 0, acc = 0;
    float a = threadIdx.x;
    for (;;) {
        if (threadIdx.x == 0) n = atomicAdd(counter, chunk);
        n = __builtin_amdgcn_readfirstlane(n);
        if (n >= total) break;
        acc += n;
        for (int i = 0; i < spin; ++i) a = fmaf(a, 0.999f, 0.5f);     // stand-in for the unit's work
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc + (a == 1.2345f);
}

int main() {
    unsigned *counter, *sink;
    (void)hipMalloc(&counter, 256); (void)hipMalloc(&sink, 4 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int spin : {0, 2000}) for (unsigned waves : {256u, 2048u, 3072u}) for (unsigned total : {8192u, 131072u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(counter, 0, 4);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(take, dim3(waves), dim3(64), 0, 0, counter, total, 1u, sink, spin);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("spin=%d waves=%u units=%u: %.3f ms -> %.1f ns per atomic\n", spin, waves, total, best, best * 1e6 / (total + waves));
    }
    return 0;
}
