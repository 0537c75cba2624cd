// Developer probe (GPU box): where does the dispatcher put single-wave workgroups, and when do they start?
// hipcc --offload-arch=gfx950 -O3 tools/placement_probe.hip -o /tmp/placement && /tmp/placement
// Each wave records HW_ID / XCC_ID and its start/end clock, spinning ~40 us in between (168-VGPR footprint is
// imitated with amdgpu_waves_per_eu(3,3) so that at most 3 waves fit a SIMD, like the search kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void probe(unsigned* hw, unsigned long long* t, int spin) {
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = fmaf(a, 0.999f, 0.5f);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        hw[blockIdx.x * 2] = id; hw[blockIdx.x * 2 + 1] = xcc;
        t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = t1;
    }
    if (a == 12345.678f) hw[0] = 0;
}

int main() {
    for (int n : {1024, 2048, 3072, 8192}) {
        unsigned* hw; unsigned long long* t;
        hipMalloc(&hw, n * 8); hipMalloc(&t, n * 16);
        hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, hw, t, 20000);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, hw, t, 20000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(n * 2); std::vector<unsigned long long> tt(n * 2);
        hipMemcpy(h.data(), hw, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(tt.data(), t, n * 16, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_simd, per_cu, per_xcc;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int i = 0; i < n; ++i) { tmin = std::min(tmin, tt[2 * i]); tmax = std::max(tmax, tt[2 * i + 1]); }
        // waves that started within the first 10 % of the kernel = first residency set
        int first = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned id = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            if (tt[2 * i] - tmin > (tmax - tmin) / 10) continue;
            ++first;
            const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++; per_simd[(cukey << 2) | simd]++; per_xcc[xcc]++;
        }
        std::map<int, int> hist_simd, hist_cu;
        for (auto& kv : per_simd) hist_simd[kv.second]++;
        for (auto& kv : per_cu) hist_cu[kv.second]++;
        printf("n=%d: kernel %.1f us (clock ticks %llu), first-wave set %d, CUs used %zu, SIMDs used %zu\n", n,
               0.0, (unsigned long long)(tmax - tmin), first, per_cu.size(), per_simd.size());
        printf("   waves per SIMD histogram:"); for (auto& kv : hist_simd) printf(" %d:%d", kv.first, kv.second); printf("\n");
        printf("   waves per CU histogram:"); for (auto& kv : hist_cu) printf(" %d:%d", kv.first, kv.second); printf("\n");
        printf("   per XCC:"); for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second); printf("\n");
        printf("   first ids: "); for (int i = 0; i < 12; ++i) printf("%08x/%x ", h[2 * i], h[2 * i + 1]); printf("\n");
        hipFree(hw); hipFree(t);
    }
    return 0;
}
