// Developer microbenchmark (GPU box): issue rates of the instructions the rollout is made of.
// hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valu_mb && /tmp/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 m = {0.999f, 1.001f}, c = {0.5f, 0.25f};
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) {  // 8 independent v_fma_f32
            a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f); a2 = fmaf(a2, 0.999f, 0.5f); a3 = fmaf(a3, 0.999f, 0.5f);
            a4 = fmaf(a4, 0.999f, 0.5f); a5 = fmaf(a5, 0.999f, 0.5f); a6 = fmaf(a6, 0.999f, 0.5f); a7 = fmaf(a7, 0.999f, 0.5f);
        } else if (OP == 1) {  // 4 v_pk_fma_f32 (8 lanes-worth of fma)
            p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
            p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
        } else if (OP == 2) {  // 4 v_fma_f64
            d0 = fma(d0, 0.999, 0.5); d1 = fma(d1, 0.999, 0.5); d2 = fma(d2, 0.999, 0.5); d3 = fma(d3, 0.999, 0.5);
        } else if (OP == 3) {  // 8 v_rcp_f32
            a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
            a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
        } else if (OP == 4) {  // 2 sincosf (OCML)
            float s, cc; sincosf(a0, &s, &cc); a0 = s + cc; sincosf(a1, &s, &cc); a1 = s + cc;
        } else if (OP == 5) {  // 4 v_add_f64
            d0 = d0 + 0.5; d1 = d1 + 0.5; d2 = d2 + 0.5; d3 = d3 + 0.5;
        } else if (OP == 6) {  // 4 cvt f32->f64 + add
            d0 = d0 + (double)a0; d1 = d1 + (double)a1; d2 = d2 + (double)a2; d3 = d3 + (double)a3;
        } else if (OP == 7) {  // 8 dependent v_fma_f32 (one chain)
            a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f);
            a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f);
        } else if (OP == 8) {  // 8 v_cndmask via compare
            a0 = a0 > 0.5f ? a1 : a2; a1 = a1 > 0.5f ? a2 : a3; a2 = a2 > 0.5f ? a3 : a4; a3 = a3 > 0.5f ? a4 : a5;
            a4 = a4 > 0.5f ? a5 : a6; a5 = a5 > 0.5f ? a6 : a7; a6 = a6 > 0.5f ? a7 : a0; a7 = a7 > 0.5f ? a0 : a1;
        } else if (OP == 9) {  // 2 sin via v_sin_f32 (native)
            a0 = __sinf(a0); a1 = __sinf(a1);
        } else if (OP == 10) {  // ONE dependent chain of 8 v_pk_fma_f32
            p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c);
            p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p0 = __builtin_elementwise_fma(p0, m, c);
        } else if (OP == 11) {  // TWO interleaved dependent chains of v_fma_f32 (4 + 4)
            a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f);
            a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f); a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f);
        } else if (OP == 12) {  // TWO interleaved dependent chains of v_pk_fma_f32 (4 + 4)
            p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
            p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c); p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
        } else if (OP == 13) {  // FOUR interleaved dependent chains of v_fma_f32 (2 each)
            a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f); a2 = fmaf(a2, 0.999f, 0.5f); a3 = fmaf(a3, 0.999f, 0.5f);
            a0 = fmaf(a0, 0.999f, 0.5f); a1 = fmaf(a1, 0.999f, 0.5f); a2 = fmaf(a2, 0.999f, 0.5f); a3 = fmaf(a3, 0.999f, 0.5f);
        } else if (OP == 14) {  // v_fma_f32 with clamp, 8 independent
            a0 = __builtin_amdgcn_fmed3f(fmaf(a0, 0.999f, 0.5f), 0.f, 1.f); a1 = __builtin_amdgcn_fmed3f(fmaf(a1, 0.999f, 0.5f), 0.f, 1.f);
            a2 = __builtin_amdgcn_fmed3f(fmaf(a2, 0.999f, 0.5f), 0.f, 1.f); a3 = __builtin_amdgcn_fmed3f(fmaf(a3, 0.999f, 0.5f), 0.f, 1.f);
            a4 = __builtin_amdgcn_fmed3f(fmaf(a4, 0.999f, 0.5f), 0.f, 1.f); a5 = __builtin_amdgcn_fmed3f(fmaf(a5, 0.999f, 0.5f), 0.f, 1.f);
            a6 = __builtin_amdgcn_fmed3f(fmaf(a6, 0.999f, 0.5f), 0.f, 1.f); a7 = __builtin_amdgcn_fmed3f(fmaf(a7, 0.999f, 0.5f), 0.f, 1.f);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
int bench(const char* name, double ops_per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 wave per SIMD per block per CU
    float* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * ITERS * ops_per_iter;  // wave-instructions
    const double per_simd_cycle = wave_instr / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %.3f wave-instr/cycle/SIMD @2.4GHz  (%.2f cycles per wave-instr)\n", name,
           waves_per_simd, ms, per_simd_cycle, 1.0 / per_simd_cycle);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    for (int w : {1, 2, 3, 4, 8}) {
        bench<0>("v_fma_f32 x8 indep", 8, w);
        bench<7>("v_fma_f32 x8 dependent", 8, w);
        bench<1>("v_pk_fma_f32 x4", 4, w);
        bench<2>("v_fma_f64 x4", 4, w);
        bench<5>("v_add_f64 x4", 4, w);
        bench<6>("cvt_f64_f32+add_f64 x4", 8, w);
        bench<3>("v_rcp_f32 x8", 8, w);
        bench<8>("cmp+cndmask x8", 16, w);
        bench<10>("v_pk_fma x8 ONE dep chain", 8, w);
        bench<11>("v_fma x8 TWO dep chains", 8, w);
        bench<12>("v_pk_fma x8 TWO dep chains", 8, w);
        bench<13>("v_fma x8 FOUR dep chains", 8, w);
        bench<14>("v_fma clamp x8 indep", 8, w);
    }
    return 0;
}
