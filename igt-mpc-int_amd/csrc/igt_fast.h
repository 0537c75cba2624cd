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

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float x) { return (f2){x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 rcp2(f2 x) { return (f2){__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f2 clamp01(f2 x) {
    return (f2){__builtin_amdgcn_fmed3f(x.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(x.y, 0.0f, 1.0f)};
}

// |d| <= ~0.12: sin to d^5, cos to d^4 (truncation 7e-11 / 4e-9)
__device__ __forceinline__ void small_sincos2(f2 d, f2& sd, f2& cd) {
    const f2 d2 = d * d;
    const f2 p = fma2(d2, splat(1.0f / 120.0f), splat(-1.0f / 6.0f));
    sd = fma2(d * d2, p, d);
    const f2 q = fma2(d2, splat(1.0f / 24.0f), splat(-0.5f));
    cd = fma2(d2, q, splat(1.0f));
}
// half-step offsets |d| <= h/2 * |rate| (<= 0.03 for any feasible state at h = 0.025): sin to d^3, cos to d^2;
// the dropped terms d^5/120 and d^4/24 are below 3e-8 there
__device__ __forceinline__ void tiny_sincos2(f2 d, f2& sd, f2& cd) {
    const f2 d2 = d * d;
    sd = fma2(d * d2, splat(-1.0f / 6.0f), d);
    cd = fma2(d2, splat(-0.5f), splat(1.0f));
}
// |d| <= 0.5 variant (coarser discretisations): two more terms
__device__ __forceinline__ void small_sincos2_hi(f2 d, f2& sd, f2& cd) {
    const f2 d2 = d * d;
    f2 p = fma2(d2, splat(-1.0f / 5040.0f), splat(1.0f / 120.0f));
    p = fma2(d2, p, splat(-1.0f / 6.0f));
    sd = fma2(d * d2, p, d);
    f2 q = fma2(d2, splat(1.0f / 40320.0f), splat(-1.0f / 720.0f));
    q = fma2(d2, q, splat(1.0f / 24.0f));
    q = fma2(d2, q, splat(-0.5f));
    cd = fma2(d2, q, splat(1.0f));
}

__device__ __forceinline__ void rotate2(f2& s, f2& c, f2 sd, f2 cd) {
    const f2 s_ = fma2(s, cd, c * sd);
    const f2 c_ = fma2(c, cd, -(s * sd));
    s = s_; c = c_;
}

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

template <bool HI_ORDER>
struct FastPair {
    // per-scenario constants
    float h, hh, h6, kv, inv_lr, lr_ratio, big;
    double b0, b1, dt;
    int n_rk4;

    __device__ __forceinline__ void init(const KP& P, double b0_, double b1_, double kv_) {
        h = (float)P.h; hh = (float)(P.h / 2); h6 = (float)(P.h / 6);
        kv = (float)kv_; inv_lr = (float)(1.0 / P.l_r); lr_ratio = (float)P.lr_ratio;
        big = 1.2676506e30f;      // 2^100
        b0 = b0_; b1 = b1_; dt = P.dt; n_rk4 = P.n_rk4;
    }
    static __device__ __forceinline__ void ssc(f2 d, f2& sd, f2& cd) {
        if (HI_ORDER) small_sincos2_hi(d, sd, cd); else small_sincos2(d, sd, cd);
    }
    // half-step (h/2-scaled) offsets
    static __device__ __forceinline__ void ssc_half(f2 d, f2& sd, f2& cd) {
        if (HI_ORDER) small_sincos2(d, sd, cd); else tiny_sincos2(d, sd, cd);
    }
    // curvature of both candidates at break-point-relative arguments e + o*2^100
    __device__ __forceinline__ f2 curv(f2 e0, f2 e1, f2 o) const {
        // scalar v_fma_f32 ... clamp (the packed form cannot carry the clamp the compiler folds in)
        const f2 c0 = {__builtin_amdgcn_fmed3f(fmaf(o.x, big, e0.x), 0.0f, 1.0f),
                       __builtin_amdgcn_fmed3f(fmaf(o.y, big, e0.y), 0.0f, 1.0f)};
        const f2 c1 = {__builtin_amdgcn_fmed3f(fmaf(o.x, big, e1.x), 0.0f, 1.0f),
                       __builtin_amdgcn_fmed3f(fmaf(o.y, big, e1.y), 0.0f, 1.0f)};
        return (c0 - c1) * splat(kv);
    }

    // One RK4 sub-step, general curvature.  In/out: base pairs (s1,c1)=(sin,cos)(beta+epsi),
    // (s2,c2)=(sin,cos)(psi+beta); v1, ey, d0, d1 working values; acc_* increments of this control step.
    struct Work {
        f2 s1, c1, s2, c2, v1, ey, d0, d1, acc_s, acc_ey, acc_ep, acc_x, acc_y, acc_psi;
    };

    // Weighted stage sums in factorised form.  With per-stage offsets (sdA,cdA), (sdB,cdB), (sdC,cdC) of the
    // (beta+epsi) angle and per-stage gains g_j:
    //     sum_j w_j g_j cos(theta1 + d_j) = c1 A - s1 B,   sum_j w_j g_j sin(theta1 + d_j) = s1 A + c1 B,
    //     A = g1 + 2 g2 cdA + 2 g3 cdB + g4 cdC,           B = 2 g2 sdA + 2 g3 sdB + g4 sdC          (w = 1,2,2,1)
    // Both branches of substep() evaluate these through the same two helpers, so they agree bit for bit
    // whenever K == 0 (then g_j = v_j * 1 and the offsets coincide with the psi offsets).
    static __device__ __forceinline__ void stage_sums(f2 g1, f2 g2, f2 g3, f2 g4, f2 sdA, f2 cdA, f2 sdB, f2 cdB,
                                                      f2 sdC, f2 cdC, f2& A, f2& B) {
        const f2 t2 = splat(2.0f) * g2, t3 = splat(2.0f) * g3;
        A = fma2(g4, cdC, fma2(t3, cdB, fma2(t2, cdA, g1)));
        B = fma2(g4, sdC, fma2(t3, sdB, t2 * sdA));
    }

    // MODE 0: general (K decided per stage argument); 1: K == 0 at every stage argument of every lane; 2: every stage
    // argument of every lane lies strictly inside the arc, K == kv.  All three give identical bits where they apply.
    template <int MODE>
    __device__ __forceinline__ void substep(f2 a, f2 ha, f2 sblr, Work& w) const {
        constexpr bool K0 = MODE == 1, KC = MODE == 2;
        const f2 H = splat(h), HH = splat(hh), H6 = splat(h6), BIG = splat(big), ONE = splat(1.0f), TWO = splat(2.0f);
        const f2 v1 = w.v1;
        const f2 v2 = v1 + ha;            // stages 2,3
        const f2 v4 = v2 + ha;            // stage 4
        const f2 s1 = w.s1, c1 = w.c1;
        const f2 w1 = v1 * sblr, w2 = v2 * sblr, w4 = v4 * sblr;
        // psi offsets h/2 w1 (stage 2), h/2 w2 (stages 3 AND 4, frenet.py:111)
        f2 sd2, cd2, sd3, cd3;
        ssc_half(HH * w1, sd2, cd2);
        ssc_half(HH * w2, sd3, cd3);
        // psi advances by h w2 per sub-step in closed form (w is linear in v: w1 + 4 w2 + w4 = 6 w2); epsi advances by the
        // same amount minus the curvature term  corr = h/6 (K1 ds1 + 2 K2 ds2 + 2 K3 ds3 + K4 ds4), which is exactly 0
        // on the K == 0 branch -- there both base pairs are rotated by (sin,cos)(h w2) and nothing else.
        f2 As, Bs, Ae, Be, ip, sdC, cdC, sdP, cdP, corr = splat(0.0f);
        if (K0) {
            // K == 0 at every stage argument of every lane: 1 - K ey = 1 and depsi = dpsi, so the
            // (beta+epsi) stage offsets are the psi offsets and the gains are the speeds.
            ssc(H * w2, sdP, cdP);
            sdC = sdP; cdC = cdP;
            stage_sums(v1, v2, v2, v4, sd2, cd2, sd3, cd3, sdC, cdC, Ae, Be);
            As = Ae; Bs = Be;
            ip = H * w2;
        } else {
            const f2 e0 = fma2(w.d0, BIG, ONE), e1 = fma2(w.d1, BIG, ONE);
            const f2 ey = w.ey;
            f2 sdA, cdA, sdB, cdB, sa, ca;
            // ---- stage 1
            f2 K = KC ? splat(kv) : (clamp01(e0) - clamp01(e1)) * splat(kv);
            const f2 g1 = v1 * rcp2(fma2(-K, ey, ONE));
            const f2 ds1 = g1 * c1;
            const f2 de1 = v1 * s1;
            const f2 kd1 = ds1 * K;
            const f2 dp1 = w1 - kd1;
            // ---- stage 2: arguments base + h/2 k1
            ssc_half(HH * dp1, sdA, cdA);
            sa = s1; ca = c1; rotate2(sa, ca, sdA, cdA);
            if (!KC) K = curv(e0, e1, HH * ds1);
            const f2 g2 = v2 * rcp2(fma2(-K, fma2(HH, de1, ey), ONE));
            const f2 ds2 = g2 * ca;
            const f2 de2 = v2 * sa;
            const f2 kd2 = ds2 * K;
            const f2 dp2 = w2 - kd2;
            // ---- stage 3: base + h/2 k2
            ssc_half(HH * dp2, sdB, cdB);
            sa = s1; ca = c1; rotate2(sa, ca, sdB, cdB);
            if (!KC) K = curv(e0, e1, HH * ds2);
            const f2 g3 = v2 * rcp2(fma2(-K, fma2(HH, de2, ey), ONE));
            const f2 ds3 = g3 * ca;
            const f2 de3 = v2 * sa;
            const f2 kd3 = ds3 * K;
            const f2 dp3 = w2 - kd3;
            // ---- stage 4: base + h k3 (only its cosine is needed individually, for depsi)
            ssc(H * dp3, sdC, cdC);
            if (!KC) K = curv(e0, e1, H * ds3);
            const f2 g4 = v4 * rcp2(fma2(-K, fma2(H, de3, ey), ONE));
            const f2 ds4 = g4 * fma2(c1, cdC, -(s1 * sdC));
            const f2 kd4 = ds4 * K;
            stage_sums(g1, g2, g3, g4, sdA, cdA, sdB, cdB, sdC, cdC, As, Bs);
            stage_sums(v1, v2, v2, v4, sdA, cdA, sdB, cdB, sdC, cdC, Ae, Be);
            ssc(H * w2, sdP, cdP);
            corr = H6 * (kd1 + TWO * kd2 + TWO * kd3 + kd4);
            ip = H * w2 - corr;
        }
        // ---- Frenet increments (frenet.py:113-115)
        const f2 is = H6 * fma2(c1, As, -(s1 * Bs));
        const f2 ie = H6 * fma2(s1, Ae, c1 * Be);
        // ---- Cartesian rows collapse to one rotation of (A,B) as well (stage 4 shares stage 3's offset)
        const f2 v34 = fma2(TWO, v2, v4);
        const f2 tv2 = TWO * v2;
        const f2 Ac = fma2(v34, cd3, fma2(tv2, cd2, v1));
        const f2 Bc = fma2(v34, sd3, tv2 * sd2);
        w.acc_x = fma2(H6, fma2(w.c2, Ac, -(w.s2 * Bc)), w.acc_x);
        w.acc_y = fma2(H6, fma2(w.s2, Ac, w.c2 * Bc), w.acc_y);
        w.acc_psi = fma2(H6, w1 + splat(4.0f) * w2 + w4, w.acc_psi);
        w.acc_s += is; w.acc_ey += ie; w.acc_ep += ip;
        w.d0 += is; w.d1 += is; w.ey += ie; w.v1 = v4;
        // ---- base pairs for the next sub-step: both advance by h w2, (beta+epsi) additionally by -corr
        rotate2(w.s1, w.c1, sdP, cdP);
        rotate2(w.s2, w.c2, sdP, cdP);
        if (!K0) {
            f2 sd, cd;
            ssc(-corr, sd, cd);
            rotate2(w.s1, w.c1, sd, cd);
        }
    }

    // n_rk4 sub-steps of one control step.  The K == 0 branch (or the K == kv one) is taken when it is provably
    // exact for every active lane of the wave: straight route, or every lane's stage arguments stay outside (inside)
    // the arc [b0, b1] by the travel bound |ds| <= 2 |v| (1/(1 - K ey) < 2 for any state near the road).
    // The lanes of a wave may belong to different scenarios (emit uses UNIFORM = false: general branch only).
    template <bool UNIFORM>
    __device__ __forceinline__ void substeps(f2 a, f2 sblr, Work& w) const {
        const f2 ha = splat(hh) * a;
        if (!UNIFORM) {
            for (int j = 0; j < n_rk4; ++j) substep<0>(a, ha, sblr, w);
            return;
        }
        if (kv == 0.0f) {                      // straight route: scalar condition, hoisted
            for (int j = 0; j < n_rk4; ++j) substep<1>(a, ha, sblr, w);
            return;
        }
        {   // the whole control step: |travel| <= 2 dt (|v| + dt |a|)
            const f2 m = splat(2.0f * (float)dt) * (__builtin_elementwise_abs(w.v1) + splat((float)dt) * __builtin_elementwise_abs(a));
            const f2 lo = w.d0 + m, hi = w.d1 - m, in0 = w.d0 - m, in1 = w.d1 + m;
            const bool clear = ((lo.x < 0.0f) | (hi.x > 0.0f)) & ((lo.y < 0.0f) | (hi.y > 0.0f));
            const bool inside = (in0.x > 0.0f) & (in1.x < 0.0f) & (in0.y > 0.0f) & (in1.y < 0.0f);
            if (__all(clear)) {
                for (int j = 0; j < n_rk4; ++j) substep<1>(a, ha, sblr, w);
                return;
            }
            if (__all(inside)) {
                for (int j = 0; j < n_rk4; ++j) substep<2>(a, ha, sblr, w);
                return;
            }
        }
        for (int j = 0; j < n_rk4; ++j) {
            // travel bound of this sub-step: |o| <= h |ds| <= 2 h (|v| + |h a|)
            const f2 m = splat(2.0f * h) * (__builtin_elementwise_abs(w.v1) + splat(2.0f) * __builtin_elementwise_abs(ha));
            const f2 lo = w.d0 + m, hi = w.d1 - m, in0 = w.d0 - m, in1 = w.d1 + m;
            const bool clear = ((lo.x < 0.0f) | (hi.x > 0.0f)) & ((lo.y < 0.0f) | (hi.y > 0.0f));
            const bool inside = (in0.x > 0.0f) & (in1.x < 0.0f) & (in0.y > 0.0f) & (in1.y < 0.0f);
            if (__all(clear)) substep<1>(a, ha, sblr, w);
            else if (__all(inside)) substep<2>(a, ha, sblr, w);
            else substep<0>(a, ha, sblr, w);
        }
    }
};

// ---------------------------------------------------------------------------------------
// one pass over a PAIR of candidates of one scenario
// ---------------------------------------------------------------------------------------
// CAND = CAND_LATTICE / CAND_RAMP_HOLD: both candidates share the steering profile (c and c+64 have the same j)
// and satisfy the input box / rate limits by construction; CAND_TABLE: controls come from the table and are checked.
// BOOK = false (emit): cost and verdicts are skipped, only the trajectory is produced.
template <int CAND, bool HI_ORDER, bool BOOK, bool UNIFORM, typename T, class Sink, bool EARLY_EXIT = false>
__device__ __forceinline__ void rollout_pair(const KP& P, const Scenario<T>& S, const int (&cidx)[2],
                                             const double* __restrict__ table,
                                             const double* __restrict__ cinf, Sink& sink, double (&Jout)[2],
                                             unsigned (&vout)[2], double (&sN)[2], double (&vN)[2]) {
    constexpr bool LATTICE = CAND != CAND_TABLE;   // generated, steering shared by the pair
    typedef FastPair<HI_ORDER> FP;
    FP fp;
    fp.init(P, S.b0, S.b1, S.kv);
    double s[2], ey[2], ep[2], v[2], x[2], y[2], psi[2], J[2], a_d[2], df_d[2], da[2], ddf = 0.0;
    unsigned viol[2];
    bool dead = false;     // EARLY_EXIT: the whole slice is already infeasible
    const float ey_lim = (float)P.ey_lim, tol = (float)P.tol, vmin = (float)P.v_min, vmax = (float)P.v_max;
    const float w_u = (float)P.w_u, dmin2 = (float)P.dmin2, ratio2 = fp.lr_ratio * fp.lr_ratio;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        x[q] = S.x0[0]; y[q] = S.x0[1]; s[q] = S.x0[2]; ey[q] = S.x0[3]; ep[q] = S.x0[4]; v[q] = S.x0[5]; psi[q] = S.x0[6];
        a_d[q] = S.a_prev; df_d[q] = S.df_prev;
        if (CAND == CAND_LATTICE) {
            // da_i = -ra + (2 ra) i/(G-1), ddf_j likewise (oracle candidates_lattice)
            const int i = cidx[q] / P.G, j = cidx[q] - i * P.G;
            da[q] = -P.rate_a + (2 * P.rate_a) * (double)i / (double)(P.G - 1);
            if (q == 0) ddf = -P.rate_df + (2 * P.rate_df) * (double)j / (double)(P.G - 1);
        } else if (CAND == CAND_RAMP_HOLD) {
            // da / ddf hold the TARGETS here (igt_device.h cand_m)
            const int i = cidx[q] / P.G, j = cidx[q] - i * P.G;
            da[q] = clampd(S.cpar[0] + cand_m(i, P.G, P.refine_it == 0) * S.cpar[2], P.a_min, P.a_max);
            if (q == 0) ddf = clampd(S.cpar[1] + cand_m(j, P.G, P.refine_it == 0) * S.cpar[3], -P.df_max, P.df_max);
        }
        J[q] = 0.0; viol[q] = 0;
        sink.state(q, 0, S.x0);
    }
    // (sin,cos)(psi_0): the same for every candidate of the scenario
    float sp0, cp0;
    sincos_reduced(S.x0[6], sp0, cp0);
    typename FP::Work w;
    w.d0 = w.d1 = splat(0.0f);
    w.s2 = splat(sp0); w.c2 = splat(cp0);      // carried as (sin,cos)(psi + beta_k); beta_{-1} = 0
    f2 cb_prev = splat(1.0f), sb_prev = splat(0.0f);
    float ox_next = 0.0f, oy_next = 0.0f;      // obstacle 0 at the state the next trip books

    for (int k = 0; k < P.N; ++k) {
        // ---- controls of step k (double)
        f2 a, cb, sb, tu;
        if (CAND == CAND_LATTICE) {
            df_d[0] = clampd(df_d[0] + ddf, -P.df_max, P.df_max);
            df_d[1] = df_d[0];
            a_d[0] = clampd(a_d[0] + da[0], P.a_min, P.a_max);
            a_d[1] = clampd(a_d[1] + da[1], P.a_min, P.a_max);
        } else if (CAND == CAND_RAMP_HOLD) {
            df_d[0] = clampd(df_d[0] + clampd(ddf - df_d[0], -P.rate_df, P.rate_df), -P.df_max, P.df_max);
            df_d[1] = df_d[0];
            a_d[0] = clampd(a_d[0] + clampd(da[0] - a_d[0], -P.rate_a, P.rate_a), P.a_min, P.a_max);
            a_d[1] = clampd(a_d[1] + clampd(da[1] - a_d[1], -P.rate_a, P.rate_a), P.a_min, P.a_max);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double an = table[((size_t)cidx[q] * 2 + 0) * P.N + k];
                const double dn = table[((size_t)cidx[q] * 2 + 1) * P.N + k];
                // input-rate (mpc.py:301-312, u_{-1} = u_prev) and input box (mpc.py:318-321)
                if (BOOK) {
                    if (fmax(fabs(an - a_d[q]) - P.rate_a, fabs(dn - df_d[q]) - P.rate_df) > P.tol) viol[q] |= VIOL_RATE;
                    if (fmax(fmax(P.a_min - an, an - P.a_max), fmax(-P.df_max - dn, dn - P.df_max)) > P.tol)
                        viol[q] |= VIOL_BOX_U;
                }
                a_d[q] = an; df_d[q] = dn;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            sink.ctrl(q, k, a_d[q], df_d[q]);
            a[q] = (float)a_d[q];
            if (!LATTICE || q == 0) {
                // beta = atan(r tan df), r = l_r/(l_f+l_r):  cos(beta) = c/n, sin(beta) = r s/n,
                // n = sqrt(c^2 + r^2 s^2), (s,c) = (sin,cos)(df)   (|df| < pi/2)
                float sdf, cdf;
                if (CAND != CAND_TABLE && P.df_small) sincos_quadrant0((float)df_d[q], sdf, cdf);   // |df| <= df_max < pi/4
                else sincos_reduced(df_d[q], sdf, cdf);
                const float n = __builtin_amdgcn_rsqf(fmaf(ratio2 * sdf, sdf, cdf * cdf));   // argument in [r^2, 1]
                cb[q] = cdf * n;
                sb[q] = fp.lr_ratio * sdf * n;
                const float dff = (float)df_d[q];
                tu[q] = dff * dff;
            } else {
                cb[q] = cb[0]; sb[q] = sb[0]; tu[q] = tu[0];
            }
        }
        const f2 sblr = sb * splat(fp.inv_lr);
        tu = splat(w_u) * fma2(a, a, tu);                              // mpc.py:362
        // ---- bookkeeping of state k + float working set of this control step
        const float epf[2] = {(float)ep[0], (float)ep[1]};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w.ey[q] = (float)ey[q];
            w.v1[q] = (float)v[q];
            if (BOOK) {
                const float t = fmaf(w.ey[q], w.ey[q], fmaf(epf[q], epf[q], tu[q]));    // mpc.py:362-364
                J[q] += (double)t;
                if (fabsf(w.ey[q]) - ey_lim > tol) viol[q] |= VIOL_EY;          // mpc.py:296-299
                if (fmaxf(vmin - w.v1[q], w.v1[q] - vmax) > tol) viol[q] |= VIOL_BOX_V;   // mpc.py:316-317 (k < N)
            }
            if (fp.kv != 0.0f) {               // break-point-relative arc length (unused on straight routes)
                w.d0[q] = (float)(s[q] - fp.b0);
                w.d1[q] = (float)(s[q] - fp.b1);
            }
        }
        if (BOOK && k == P.N - 1) terminal_viol2(P, v, a_d, cinf, viol);                  // mpc.py:177-180
        // (sin,cos)(epsi): heading errors beyond pi/4 are rare, so the range reduction is skipped when no lane needs it
        if (__all((fabsf(epf[0]) < QUADRANT0) & (fabsf(epf[1]) < QUADRANT0))) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { float se, ce; sincos_quadrant0(epf[q], se, ce); w.s1[q] = se; w.c1[q] = ce; }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) { float se, ce; sincos_reduced(ep[q], se, ce); w.s1[q] = se; w.c1[q] = ce; }
        }
        if (BOOK && UNIFORM && EARLY_EXIT) {
            // search only: once every candidate of the slice has failed a verdict, nothing rolled further can win
            if (__all((viol[0] != 0) & (viol[1] != 0)) && !(P.dev & 2)) { dead = true; break; }
        }
        if (BOOK) {                                                    // collision, mpc.py:223-226 (k >= 1)
            // obstacle 0 of the NEXT state is requested now and used one trip later: a scalar load's latency is
            // longer than the bookkeeping between here and the sub-steps, and few waves share a SIMD
            const float ox0 = ox_next, oy0 = oy_next;
            if (P.n_obs > 0) { ox_next = S.obs[k + 1]; oy_next = S.obs[(P.N + 1) + k + 1]; }
            for (int o = 0; o < P.n_obs && k >= 1; ++o) {
                const double ox = o == 0 ? (double)ox0 : (double)S.obs[(o * 2 + 0) * (P.N + 1) + k];
                const double oy = o == 0 ? (double)oy0 : (double)S.obs[(o * 2 + 1) * (P.N + 1) + k];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float dx = (float)(x[q] - ox), dy = (float)(y[q] - oy);
                    if (dmin2 - fmaf(dx, dx, dy * dy) > tol) viol[q] |= VIOL_COLLISION;
                }
            }
        }
        rotate2(w.s1, w.c1, sb, cb);                                   // (sin,cos)(beta + epsi)
        // (sin,cos)(psi + beta_k) from (psi + beta_{k-1}): rotate by beta_k - beta_{k-1}, re-normalise
        {
            const f2 sdb = fma2(sb, cb_prev, -(cb * sb_prev));
            const f2 cdb = fma2(cb, cb_prev, sb * sb_prev);
            rotate2(w.s2, w.c2, sdb, cdb);
            const f2 n = fma2(w.s2, w.s2, w.c2 * w.c2);
            const f2 r = fma2(n, splat(-0.5f), splat(1.5f));
            w.s2 *= r; w.c2 *= r;
            cb_prev = cb; sb_prev = sb;
        }
        w.acc_s = splat(0.f); w.acc_ey = splat(0.f); w.acc_ep = splat(0.f);
        w.acc_x = splat(0.f); w.acc_y = splat(0.f); w.acc_psi = splat(0.f);
        fp.template substeps<UNIFORM>(a, sblr, w);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            s[q] += (double)w.acc_s[q]; ey[q] += (double)w.acc_ey[q]; ep[q] += (double)w.acc_ep[q];
            x[q] += (double)w.acc_x[q]; y[q] += (double)w.acc_y[q];
            if (Sink::kKeepsStates) psi[q] += (double)w.acc_psi[q];      // psi feeds nothing back (search: dead)
            v[q] = fma(fp.dt, a_d[q], v[q]);
            const double nxt[7] = {x[q], y[q], s[q], ey[q], ep[q], v[q], psi[q]};
            sink.state(q, k + 1, nxt);
        }
    }
    if (dead) {            // costs are meaningless; the verdict bits that ended the roll are what is reported
        Jout[0] = Jout[1] = 0.0; vout[0] = viol[0]; vout[1] = viol[1]; sN[0] = sN[1] = 0.0; vN[0] = vN[1] = 0.0;
        return;
    }
#pragma unroll
    for (int q = 0; q < 2 && BOOK; ++q) {
        const float epf = (float)ep[q], eyq = (float)ey[q];
        J[q] += (double)fmaf(eyq, eyq, epf * epf);
        if (fabsf(eyq) - ey_lim > tol) viol[q] |= VIOL_EY;
        for (int o = 0; o < P.n_obs; ++o) {
            const float dx = (float)(x[q] - (o == 0 ? (double)ox_next : (double)S.obs[(o * 2 + 0) * (P.N + 1) + P.N]));
            const float dy = (float)(y[q] - (o == 0 ? (double)oy_next : (double)S.obs[(o * 2 + 1) * (P.N + 1) + P.N]));
            if (dmin2 - fmaf(dx, dx, dy * dy) > tol) viol[q] |= VIOL_COLLISION;
        }
        if (!(fabs(x[q]) < 1e300 && fabs(y[q]) < 1e300 && fabs(s[q]) < 1e300 && fabs(ey[q]) < 1e300 &&
              fabs(ep[q]) < 1e300 && fabs(psi[q]) < 1e300))
            viol[q] |= VIOL_NONFINITE;
        sN[q] = s[q]; vN[q] = v[q];
        Jout[q] = J[q]; vout[q] = viol[q];
    }
    if (!BOOK) { Jout[0] = Jout[1] = 0.0; vout[0] = vout[1] = 0; sN[0] = sN[1] = 0.0; vN[0] = vN[1] = 0.0; }
}

}  // namespace igt
