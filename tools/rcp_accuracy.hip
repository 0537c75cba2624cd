// Developer probe: relative error of v_rcp_f64 / v_rsq_f64 on gfx950 and of the estimate refined by 0, 1, 2 Newton steps.
//   hipcc --offload-arch=gfx950 -O3 tools/rcp_accuracy.hip -o /tmp/rcp && /tmp/rcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    out[i * 6 + 0] = r;
    r = fma(fma(-v, r, 1.0), r, r); out[i * 6 + 1] = r;
    r = fma(fma(-v, r, 1.0), r, r); out[i * 6 + 2] = r;
    double y = __builtin_amdgcn_rsq(v);
    out[i * 6 + 3] = y;
    const double hx = 0.5 * v;
    y = y * fma(-(hx * y), y, 1.5); out[i * 6 + 4] = y;
    y = y * fma(-(hx * y), y, 1.5); out[i * 6 + 5] = y;
}
int main() {
    const int n = 1 << 20;
    double* hx = new double[n]; double* ho = new double[n * 6];
    for (int i = 0; i < n; ++i) hx[i] = 0.25 + 3.75 * (i + 0.5) / n;          // [0.25, 4]
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 48);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, n * 48, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double v = hx[i], tr = 1.0L / v, ts = 1.0L / sqrtl(v);
        for (int j = 0; j < 3; ++j) { const double d = fabsl((ho[i * 6 + j] - tr) / tr); if (d > e[j]) e[j] = d; }
        for (int j = 3; j < 6; ++j) { const double d = fabsl((ho[i * 6 + j] - ts) / ts); if (d > e[j]) e[j] = d; }
    }
    printf("max relative error on [0.25, 4] (2^-53 = 1.11e-16)\n  v_rcp_f64: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n  v_rsq_f64: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n",
           e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
