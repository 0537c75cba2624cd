// Developer probe: issue cost of v_mfma_f64_16x16x4_f64 on gfx950 -- one dependent accumulator chain per wave vs
// 2 / 4 / 8 independent chains, at 1, 2, 4 and 8 waves per SIMD, with the shader clock the kernel actually ran at
// (s_memtime ticks of wave 0 against the constant 100 MHz s_memrealtime).   hipcc --offload-arch=gfx950 -O3 tools/mfma64_microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
    const long long w0 = wall_clock64();
    f64x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = wall_clock64() - w0; }
}

template <int CHAINS>
void run(int waves_per_simd) {
    double* out; long long* cyc;
    hipMalloc(&out, 1 << 23); hipMalloc(&cyc, 16);
    const int iters = 4096;
    // one workgroup of 256 threads = 4 waves = one per SIMD of a CU; waves_per_simd workgroups per CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, cyc, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cc[2]; hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost);
    const long long c = cc[0];
    const double ghz = (double)cc[0] / (double)cc[1] * 0.1;          // shader ticks per 100 MHz tick
    const double mfmas_per_simd = (double)iters * CHAINS * waves_per_simd;
    const double flop = (double)blocks * 4 * iters * CHAINS * 2048.0;
    printf("chains %d, waves/SIMD %d: %.3f ms, %.1f TFLOP/s, shader clock %.2f GHz (data-sheet rate at that clock: %.1f TFLOP/s), "
           "shader cycles per MFMA per SIMD %.1f, clock64 ticks per MFMA in wave 0: %.1f\n",
           CHAINS, waves_per_simd, ms, flop / ms / 1e9, ghz, 78.6 * ghz / 2.4, ms * 1e-3 * ghz * 1e9 / mfmas_per_simd,
           (double)c / (iters * CHAINS));
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
