// igt_math64.h -- double-precision elementary functions of the device code (gfx950): the pieces the float64 rollout
// (igt_fast64.h) and the candidate generators (igt_device.h) evaluate once per control step, written out so that their
// cost is known: a libm call here is hundreds of instructions (OCML's tan and atan carry range reductions for arguments
// these kernels never see), the forms below are tens.
#pragma once
#include <hip/hip_runtime.h>

namespace igt {
namespace m64 {

// ---- libm-grade sin/cos, used once per control step ------------------------------------------------------------
// minimax kernels on |r| <= pi/4 (the classical fdlibm coefficient sets), ~1 ulp
__device__ __forceinline__ void sincos_kernel(double r, double& s, double& c) {
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    c = fma(z * z, pc, fma(z, -0.5, 1.0));
}
// any |x| up to ~1e6: two-term Cody-Waite reduction to |r| <= pi/4
__device__ __forceinline__ void sincos_reduced(double x, double& s, double& c) {
    const double kd = __builtin_rint(x * 0.63661977236758134);                   // 2/pi
    double r = fma(-kd, 1.5707963267948966, x);
    r = fma(-kd, 6.123233995736766e-17, r);
    const int q = (int)kd;
    double sr, cr;
    sincos_kernel(r, sr, cr);
    const double a = (q & 1) ? cr : sr;
    const double b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
constexpr double QUADRANT0 = 0.78;     // < pi/4: sincos_reduced() has kd = 0 there, so the kernel alone agrees bit for bit

// 1/x: hardware estimate + two Newton steps (x = 1 gives exactly 1, which the K == 0 identities rely on)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1/sqrt(x), x in (0, 1]: hardware estimate + two Newton steps
__device__ __forceinline__ double rsq_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-(hx * y), y, 1.5);
    y = y * fma(-(hx * y), y, 1.5);
    return y;
}


// atan2(y, x) for x > 0.  fdlibm's atan (interval reduction to |u| <= 7/16, 11-term polynomial, < 1 ulp) with each
// interval's transform written for the pair: with t = |y|/x the reduced argument is
//     id -1: t          id 0: (2t-1)/(2+t)      id 1: (t-1)/(t+1)      id 2: (t-1.5)/(1+1.5t)      id 3: -1/t
//   = (p |y| + q x) / (p x - q |y|)   with (p, q) = (1,0), (2,-1), (1,-1), (1,-1.5), (0,-1),
// so one reciprocal serves the quotient t and the transform.  The low words of fdlibm's atan(0.5), pi/4, atan(1.5), pi/2
// are dropped (< 7e-17 absolute); the reciprocal is rcp_nr (1 ulp): measured <= 4e-16 against libm on the steering
// feedback's range (tests/test_gpu_parity.py, tracking candidates against the numpy oracle).
__device__ __forceinline__ double atan2_xpos(double y, double x) {
    const double ay = fabs(y);
    const double a16 = 16.0 * ay;
    const bool g0 = a16 >= 7.0 * x, g1 = a16 >= 11.0 * x, g2 = a16 >= 19.0 * x, g3 = a16 >= 39.0 * x;
    const double p = g3 ? 0.0 : (g1 ? 1.0 : (g0 ? 2.0 : 1.0));
    const double q = g3 ? -1.0 : (g2 ? -1.5 : (g0 ? -1.0 : 0.0));
    const double hi = g3 ? 1.57079632679489655800e+00
                         : (g2 ? 9.82793723247329054082e-01
                               : (g1 ? 7.85398163397448278999e-01 : (g0 ? 4.63647609000806093515e-01 : 0.0)));
    const double num = fma(p, ay, q * x), den = fma(p, x, -(q * ay));
    const double u = num * rcp_nr(den);
    const double z = u * u, w = z * z;
    double s1 = fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    double s2 = fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s1 = fma(w, s1, 6.66107313738753120669e-02);
    s2 = fma(w, s2, -7.69187620504482999495e-02);
    s1 = fma(w, s1, 9.09088713343650656196e-02);
    s2 = fma(w, s2, -1.11111104054623557880e-01);
    s1 = fma(w, s1, 1.42857142725034663711e-01);
    s2 = fma(w, s2, -1.99999999998764832476e-01);
    s1 = fma(w, s1, 3.33333333333329318027e-01);
    const double S = fma(z, s1, w * s2);
    const double r = hi + fma(-u, S, u);
    return copysign(r, y);
}

}  // namespace m64
}  // namespace igt
