// igt_fast.h -- the float32 rollout of the search / emit / rollout-all kernels (gfx950).
//
// Two candidates of one lane are rolled together as the two halves of packed float2 values:
// on gfx950 a wave issues one VALU instruction per ~2.3 cycles only when the next instruction
// is independent of it (4.2 when it depends on it; tools/valu_microbench.hip), so a single
// dependent chain leaves half the issue slots empty.  The pair gives every instruction an
// independent neighbour (or packs both into one v_pk_* instruction).
//
// Arithmetic (same RK4 stages as kinematic_bicycle_model_frenet.py:107-119, float derivatives):
//   * across control steps the state lives in double; inside a control step all sub-step
//     arithmetic is float on "base + small offset" quantities:
//       - K(s'): decided on d = float(s - b) + offset, relative to the break-point, with a
//         clamped fma (step(d) = clamp(d*2^100 + 1, 0, 1)) instead of compare + select;
//       - sin/cos of (beta+epsi') and (psi'+beta): rotation of the sub-step base pair by the stage
//         offset (short polynomial); the (beta+epsi) base is rebuilt from the double epsi once per
//         control step, the (psi+beta) base is carried by rotation and re-normalised once per step;
//       - v and psi do not feed back: v_j and the psi offsets are closed-form, and the x,y rows
//         collapse to one rotation of (A,B) = sum_j w_j v_j (cos,sin)(offset_j), including the
//         reference's quirk that stage 4 sees psi + h/2*k3[6] (:111).
//     increments are summed in float over the n_rk4 sub-steps and added to the double state once
//     per control step (sum <= 0.5 m: rounding ~3e-8, far below the float-accumulator random walk
//     of plain float32).
//   * cost terms and verdicts are formed in float from the double step-boundary state and summed
//     in double; controls are generated in double (u_out is the oracle's candidate to rounding).
#pragma once
#include <type_traits>
#include "igt_device.h"

namespace igt {

// sin/cos of a double angle with float polynomials: reduce in double to r in [-pi/4, pi/4]
__device__ __forceinline__ void sincos_reduced(double ang, float& s, float& c) {
    const double kd = __builtin_rint(ang * 0.63661977236758134);       // 2/pi
    const float r = (float)fma(-kd, 1.5707963267948966, ang);
    const int q = (int)kd;
    const float r2 = r * r;
    float ps = fmaf(r2, 2.7557319e-6f, -1.9841270e-4f);                // 1/9!, -1/7!
    ps = fmaf(r2, ps, 8.3333333e-3f);
    ps = fmaf(r2, ps, -1.6666667e-1f);
    const float sr = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.4801587e-5f, -1.3888889e-3f);               // 1/8!, -1/6!
    pc = fmaf(r2, pc, 4.1666667e-2f);
    pc = fmaf(r2, pc, -0.5f);
    const float cr = fmaf(r2, pc, 1.0f);
    const float a = (q & 1) ? cr : sr;
    const float b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// the same polynomials for an angle already known to lie in (-pi/4, pi/4): sincos_reduced() has kd = 0 there
// and r = (float)ang, so the two agree bit for bit
__device__ __forceinline__ void sincos_quadrant0(float r, float& s, float& c) {
    const float r2 = r * r;
    float ps = fmaf(r2, 2.7557319e-6f, -1.9841270e-4f);
    ps = fmaf(r2, ps, 8.3333333e-3f);
    ps = fmaf(r2, ps, -1.6666667e-1f);
    s = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.4801587e-5f, -1.3888889e-3f);
    pc = fmaf(r2, pc, 4.1666667e-2f);
    pc = fmaf(r2, pc, -0.5f);
    c = fmaf(r2, pc, 1.0f);
}
constexpr float QUADRANT0 = 0.78f;     // < pi/4

namespace wide {
#define IGT_NV 2
#include "igt_fast_impl.inc"
#undef IGT_NV
}  // namespace wide
namespace single {
#define IGT_NV 1
#include "igt_fast_impl.inc"
#undef IGT_NV
}  // namespace single
using wide::rollout_pair;      // the two-candidates-per-lane build is the default

}  // namespace igt
