// igt_fast64.h -- the float64 rollout of the search / emit / rollout-all kernels of the _f64 entry points (gfx950).
//
// The reference is float64 end to end (kinematic_bicycle_model_frenet.py:70-127, mpc.py:356-373), so this is the path
// that runs at the reference's precision.  It evaluates the same RK4 stages as ExactStepper<double> (igt_device.h,
// the oracle's operation order) but with the algebra of the float path (igt_fast_impl.inc) carried out in double:
//   * v and psi do not feed back: stage speeds and the psi offsets are closed-form;
//   * sin/cos of (beta+epsi') and (psi'+beta) are rotations of the sub-step's base pair by the stage offset, the
//     offset's sine / cosine from a Taylor polynomial (|offset| <= h |rate| << 1; truncation < 1e-17 relative);
//     libm-grade sincos is evaluated once per control step (of the slip angle) instead of 32 times -- the (beta + epsi)
//     and (psi + beta) pairs are carried from step to step by rotations and renormalised;
//   * all weighted stage sums are factorised, sum_j w_j g_j cos(theta + d_j) = cos(theta) A - sin(theta) B, for the
//     Frenet rows and the Cartesian rows (the latter including the reference's quirk that stage 4 sees
//     psi + h/2 k3[6], kinematic_bicycle_model_frenet.py:111);
//   * beta = atan(r tan df) is never formed: cos(beta) = c/n, sin(beta) = r s/n, n = sqrt(c^2 + r^2 s^2);
//   * K(s') is decided on (s - b) + offset >= 0 (the oracle compares s + offset >= b).
// Nothing is approximated beyond double rounding: measured <= 1e-12 against the reference-generated golden
// rollouts (curvature break-point straddlers included), i.e. three orders inside the 1e-9 bar of the _f64 tests.
// Three variants of a sub-step (K == 0 everywhere / K == kv everywhere / general), chosen by wave votes on a travel
// bound, are bit-identical wherever they apply (x - 0, x * 1, rotation by (0, 1) are exact), so search (votes),
// emit (general variant, lanes of different scenarios) and rollout-all agree bit for bit -- asserted in the tests.
#pragma once
#include "igt_device.h"
#include "igt_math64.h"

namespace igt {
namespace f64 {

using m64::sincos_kernel;
using m64::sincos_reduced;
using m64::QUADRANT0;
using m64::rcp_nr;
using m64::rsq_nr;


// ---- sin/cos of a small stage offset ---------------------------------------------------------------------------
// |d| <= 0.15: sin to d^9, cos to d^10 (first dropped terms 2e-17 / 2e-19 relative to 1)
__device__ __forceinline__ void small_sincos(double d, double& sd, double& cd) {
    const double d2 = d * d;
    double p = fma(d2, 1.0 / 362880.0, -1.0 / 5040.0);
    p = fma(d2, p, 1.0 / 120.0);
    p = fma(d2, p, -1.0 / 6.0);
    sd = fma(d * d2, p, d);
    double q = fma(d2, -1.0 / 3628800.0, 1.0 / 40320.0);
    q = fma(d2, q, -1.0 / 720.0);
    q = fma(d2, q, 1.0 / 24.0);
    q = fma(d2, q, -0.5);
    cd = fma(d2, q, 1.0);
}
// |d| <= 0.6 (coarse discretisations, n_rk4 <= 2): two more terms each
__device__ __forceinline__ void small_sincos_hi(double d, double& sd, double& cd) {
    const double d2 = d * d;
    double p = fma(d2, 1.0 / 6227020800.0, -1.0 / 39916800.0);
    p = fma(d2, p, 1.0 / 362880.0);
    p = fma(d2, p, -1.0 / 5040.0);
    p = fma(d2, p, 1.0 / 120.0);
    p = fma(d2, p, -1.0 / 6.0);
    sd = fma(d * d2, p, d);
    double q = fma(d2, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    q = fma(d2, q, -1.0 / 3628800.0);
    q = fma(d2, q, 1.0 / 40320.0);
    q = fma(d2, q, -1.0 / 720.0);
    q = fma(d2, q, 1.0 / 24.0);
    q = fma(d2, q, -0.5);
    cd = fma(d2, q, 1.0);
}

__device__ __forceinline__ void rotate(double& s, double& c, double sd, double cd) {
    const double s_ = fma(s, cd, c * sd);
    const double c_ = fma(c, cd, -(s * sd));
    s = s_; c = c_;
}
// NRK: the number of RK4 sub-steps per control step as a compile-time constant (4: what evaluate.py:109,440 passes), or 0:
// read from the parameters at run time.  With it fixed the generic sub-step loops vanish from the kernel and so does the
// register pressure they add: the control step of a straight route is 410 instructions instead of 464 (ISA count).
template <bool HI_ORDER, int NRK = 0>
struct Fast64 {
    double h, hh, h6, kv, inv_lr, lr_ratio, b0, b1, dt;
    int n_rk4;
    __device__ __forceinline__ int nrk() const { return NRK > 0 ? NRK : n_rk4; }

    __device__ __forceinline__ void init(const KP& P, double b0_, double b1_, double kv_) {
        h = P.h; hh = P.h / 2; h6 = P.h / 6;
        kv = kv_; inv_lr = 1.0 / P.l_r; lr_ratio = P.lr_ratio;
        b0 = b0_; b1 = b1_; dt = P.dt; n_rk4 = P.n_rk4;
    }
    static __device__ __forceinline__ void ssc(double d, double& sd, double& cd) {
        if (HI_ORDER) small_sincos_hi(d, sd, cd); else small_sincos(d, sd, cd);
    }
    // curvature at the break-point-relative argument d + o  (mpc.py:199 pw_const: kv [s >= b0] - kv [s >= b1])
    __device__ __forceinline__ double curv(double d0, double d1, double o) const {
        return ((d0 + o >= 0.0) ? kv : 0.0) - ((d1 + o >= 0.0) ? kv : 0.0);
    }

    // working set of one control step: base pairs (s1,c1) = (sin,cos)(beta+epsi), (s2,c2) = (sin,cos)(psi+beta);
    // v1, ey and the break-point-relative arc lengths d0, d1 advance per sub-step; acc_* are the step's increments
    struct Work {
        double s1, c1, s2, c2, v1, ey, d0, d1, acc_s, acc_ey, acc_ep, acc_x, acc_y, acc_psi;
        double sh, ch;        // (sin,cos)(h/2 w2) of the current sub-step, carried by rotation (see StepConst)
    };
    // The psi offsets of a control step form an arithmetic progression: w = v sin(beta)/l_r is linear in v and v advances
    // by h/2 a per half sub-step, so with delta = h/2 * (h/2 a) * sin(beta)/l_r
    //     h/2 w1 = h/2 w2 - delta,   h w2 = 2 (h/2 w2),   next sub-step: h/2 w2' = h/2 w2 + 2 delta.
    // One polynomial per CONTROL step (h/2 w2 of the first sub-step) plus rotations by +-delta / 2 delta and a double-angle
    // replace three polynomials per SUB-step -- the same numbers in every sub-step variant (mode-independent code).
    struct StepConst {
        double ha, sblr, sd, cd, s2d, c2d;
    };

    // A = g1 + 2 g2 cdA + 2 g3 cdB + g4 cdC,  B = 2 g2 sdA + 2 g3 sdB + g4 sdC   (weights 1,2,2,1)
    static __device__ __forceinline__ void stage_sums(double g1, double g2, double g3, double g4, double sdA, double cdA,
                                                      double sdB, double cdB, double sdC, double cdC, double& A, double& B) {
        const double t2 = 2.0 * g2, t3 = 2.0 * g3;
        A = fma(g4, cdC, fma(t3, cdB, fma(t2, cdA, g1)));
        B = fma(g4, sdC, fma(t3, sdB, t2 * sdA));
    }

    // MODE 0: general (K decided per stage argument); 1: K == 0 at every stage argument of every lane; 2: every stage
    // argument of every lane lies strictly inside the arc, K == kv.  All three give identical bits where they apply.
    // XY = false (search units whose obstacles are out of every candidate's reach): the Cartesian rows are not integrated
    template <int MODE, bool XY = true>
    __device__ __forceinline__ void substep(const StepConst& sc, Work& w) const {
        constexpr bool K0 = MODE == 1, KC = MODE == 2;
        const double ha = sc.ha, sblr = sc.sblr;
        const double v1 = w.v1;
        const double v2 = v1 + ha;            // stages 2,3
        const double v4 = v2 + ha;            // stage 4
        const double s1 = w.s1, c1 = w.c1;
        const double w1 = v1 * sblr, w2 = v2 * sblr, w4 = v4 * sblr;
        // psi offsets h/2 w1 (stage 2), h/2 w2 (stages 3 AND 4, frenet.py:111), h w2 (the sub-step's psi advance)
        const double sd3 = w.sh, cd3 = w.ch;
        double sd2 = sd3, cd2 = cd3;
        rotate(sd2, cd2, -sc.sd, sc.cd);
        const double sdP = 2.0 * sd3 * cd3, cdP = fma(-2.0 * sd3, sd3, 1.0);
        double As, Bs, Ae, Be, sdC, cdC, corr = 0.0;
        if (K0) {
            // 1 - K ey = 1 and depsi = dpsi: the (beta+epsi) stage offsets are the psi offsets, the gains the speeds
            sdC = sdP; cdC = cdP;
            stage_sums(v1, v2, v2, v4, sd2, cd2, sd3, cd3, sdC, cdC, Ae, Be);
            As = Ae; Bs = Be;
        } else {
            const double ey = w.ey, d0 = w.d0, d1 = w.d1;
            double sdA, cdA, sdB, cdB, sa, ca;
            // ---- stage 1                                                            (frenet.py:73-79)
            double K = KC ? kv : curv(d0, d1, 0.0);
            const double g1 = v1 * rcp_nr(fma(-K, ey, 1.0));
            const double ds1 = g1 * c1;
            const double de1 = v1 * s1;
            const double kd1 = ds1 * K;
            // ---- stage 2: arguments base + h/2 k1; its offset h/2 dp1 = h/2 w1 - h/2 kd1: the psi offset turned back by
            //      the curvature part (kd == 0 turns by (0, 1): exactly the psi offset, as the K == 0 variant uses)
            {
                double sk, ck;
                ssc(-(hh * kd1), sk, ck);
                sdA = sd2; cdA = cd2; rotate(sdA, cdA, sk, ck);
            }
            sa = s1; ca = c1; rotate(sa, ca, sdA, cdA);
            if (!KC) K = curv(d0, d1, hh * ds1);
            const double g2 = v2 * rcp_nr(fma(-K, fma(hh, de1, ey), 1.0));
            const double ds2 = g2 * ca;
            const double de2 = v2 * sa;
            const double kd2 = ds2 * K;
            // ---- stage 3: base + h/2 k2
            {
                double sk, ck;
                ssc(-(hh * kd2), sk, ck);
                sdB = sd3; cdB = cd3; rotate(sdB, cdB, sk, ck);
            }
            sa = s1; ca = c1; rotate(sa, ca, sdB, cdB);
            if (!KC) K = curv(d0, d1, hh * ds2);
            const double g3 = v2 * rcp_nr(fma(-K, fma(hh, de2, ey), 1.0));
            const double ds3 = g3 * ca;
            const double de3 = v2 * sa;
            const double kd3 = ds3 * K;
            // ---- stage 4: base + h k3 (only its cosine is needed individually, for depsi)
            {
                double sk, ck;
                ssc(-(h * kd3), sk, ck);
                sdC = sdP; cdC = cdP; rotate(sdC, cdC, sk, ck);
            }
            if (!KC) K = curv(d0, d1, h * ds3);
            const double g4 = v4 * rcp_nr(fma(-K, fma(h, de3, ey), 1.0));
            const double ds4 = g4 * fma(c1, cdC, -(s1 * sdC));
            const double kd4 = ds4 * K;
            stage_sums(g1, g2, g3, g4, sdA, cdA, sdB, cdB, sdC, cdC, As, Bs);
            stage_sums(v1, v2, v2, v4, sdA, cdA, sdB, cdB, sdC, cdC, Ae, Be);
            corr = h6 * (kd1 + 2.0 * kd2 + 2.0 * kd3 + kd4);        // curvature part of the epsi increment
        }
        const double ip = h * w2 - corr;                             // K == 0: h w2 - 0, the same bits
        // ---- Frenet increments (frenet.py:113-115)
        const double is = h6 * fma(c1, As, -(s1 * Bs));
        const double ie = h6 * fma(s1, Ae, c1 * Be);
        // ---- Cartesian rows collapse to one rotation of (A,B) as well (stage 4 shares stage 3's offset)
        if (XY) {
            const double v34 = fma(2.0, v2, v4);
            const double tv2 = 2.0 * v2;
            const double Ac = fma(v34, cd3, fma(tv2, cd2, v1));
            const double Bc = fma(v34, sd3, tv2 * sd2);
            w.acc_x = fma(h6, fma(w.c2, Ac, -(w.s2 * Bc)), w.acc_x);
            w.acc_y = fma(h6, fma(w.s2, Ac, w.c2 * Bc), w.acc_y);
        }
        w.acc_psi = fma(h6, w1 + 4.0 * w2 + w4, w.acc_psi);
        w.acc_s += is; w.acc_ey += ie; w.acc_ep += ip;
        w.d0 += is; w.d1 += is; w.ey += ie; w.v1 = v4;
        // ---- base pairs for the next sub-step: (psi+beta) advances by h w2, (beta+epsi) by h w2 - corr: the psi
        // advance turned back by corr.  K == 0: corr = 0 exactly, the extra turn is by (0, 1) -- bit-identical variants.
        if (XY) rotate(w.s2, w.c2, sdP, cdP);
        rotate(w.s1, w.c1, sdP, cdP);
        if (!K0) {
            double sk, ck;
            ssc(-corr, sk, ck);
            rotate(w.s1, w.c1, sk, ck);
        }
        rotate(w.sh, w.ch, sc.s2d, sc.c2d);                          // h/2 w2 of the next sub-step
    }

    // n_rk4 sub-steps of one control step.  The K == 0 variant (or the K == kv one) is taken when it is provably
    // exact for every active lane of the wave: straight route, or every lane's stage arguments stay outside (inside)
    // the arc [b0, b1] by the travel bound |ds| <= 2 |v| (1/(1 - K ey) < 2 while |K ey| < 1/2: part of the "inside" vote --
    // found by the N = 64 fuzz draws of round 4: an infeasible candidate 6 m off the road travelled 3.5 |v| dt in a step).
    // UNIFORM = false (emit: the lanes of a wave belong to different scenarios): the same choice by votes over the lanes.
    // the n_rk4 sub-steps of one variant; the reference's discretisation (4) is unrolled: no loop-carried register copies
    template <int MODE, bool XY>
    __device__ __forceinline__ void run(const StepConst& sc, Work& w) const {
        if (NRK == 4 || (NRK == 0 && n_rk4 == 4)) {
            substep<MODE, XY>(sc, w); substep<MODE, XY>(sc, w); substep<MODE, XY>(sc, w); substep<MODE, XY>(sc, w);
        } else {
            for (int j = 0; j < nrk(); ++j) substep<MODE, XY>(sc, w);
        }
    }

    // the constants of one control step (and the first sub-step's (sin, cos)(h/2 w2) in w): the same statements for the
    // search / emit / rollout-all roll-outs and for the Cartesian rows rolled on their own (cartesian_rows)
    __device__ __forceinline__ StepConst step_const(double a, double sblr, Work& w) const {
        const double ha = hh * a;
        StepConst sc;
        sc.ha = ha; sc.sblr = sblr;
        {   // delta = h/2 (w2 - w1) = (h/2)^2 a sin(beta)/l_r is tiny (< 4e-3 even at h = 0.1): sin to d^5, cos to d^4
            const double d = hh * ha * sblr, d2 = d * d;
            sc.sd = fma(d * d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), d);
            sc.cd = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
        }
        sc.s2d = 2.0 * sc.sd * sc.cd; sc.c2d = fma(-2.0 * sc.sd, sc.sd, 1.0);
        ssc(hh * ((w.v1 + ha) * sblr), w.sh, w.ch);                  // h/2 w2 of the first sub-step
        return sc;
    }

    template <bool UNIFORM, bool XY = true>
    __device__ __forceinline__ void substeps(double a, double sblr, Work& w) const {
        const double ha = hh * a;
        const StepConst sc = step_const(a, sblr, w);
        // UNIFORM: kv is a scalar (one scenario per wave) and the straight-route test is hoisted; otherwise (emit: one
        // scenario per lane) the whole-step decisions are taken by votes over the active lanes -- the variants are
        // bit-identical where they apply
        if (UNIFORM && kv == 0.0) {
            run<1, XY>(sc, w);
            return;
        }
        {   // the whole control step: |travel| <= 2 dt (|v| + dt |a|)   (a straight route's d0, d1 are -inf: clear)
            const double m = (2.0 * dt) * (fabs(w.v1) + dt * fabs(a));
            const bool clear = (w.d0 + m < 0.0) | (w.d1 - m > 0.0);
            // inside the arc the travel is v / (1 - K ey) per unit time: the factor 2 of the bound holds while |K ey| < 1/2 at
            // every stage (|ey| moves by at most m / 2 within the step) -- a candidate metres off the road votes "general"
            const bool inside = (w.d0 - m > 0.0) & (w.d1 + m < 0.0) & (fabs(kv) * (fabs(w.ey) + 0.5 * m) < 0.5);
            if (__all(clear)) {
                run<1, XY>(sc, w);
                return;
            }
            if (__all(inside)) {
                run<2, XY>(sc, w);
                return;
            }
        }
        if (!UNIFORM) {                        // lanes of different scenarios rarely agree sub-step by sub-step
            run<0, XY>(sc, w);
            return;
        }
        for (int j = 0; j < nrk(); ++j) {
            // travel bound of this sub-step: |o| <= h |ds| <= 2 h (|v| + |h a|)
            const double m = (2.0 * h) * (fabs(w.v1) + 2.0 * fabs(ha));
            const bool clear = (w.d0 + m < 0.0) | (w.d1 - m > 0.0);
            const bool inside = (w.d0 - m > 0.0) & (w.d1 + m < 0.0) & (fabs(kv) * (fabs(w.ey) + 0.5 * m) < 0.5);
            if (__all(clear)) substep<1, XY>(sc, w);
            else if (__all(inside)) substep<2, XY>(sc, w);
            else substep<0, XY>(sc, w);
        }
    }
};

// ---------------------------------------------------------------------------------------
// one pass over ONE candidate per lane
// ---------------------------------------------------------------------------------------
// CAND = CAND_LATTICE / CAND_RAMP_HOLD: generated controls, input box / rate limits hold by construction;
// CAND_TABLE: controls come from the table and are checked.  BOOK = false (emit): cost and verdicts are skipped.
// EARLY_EXIT (search): the unit stops once every candidate of the wave has failed a verdict; then only "failed" is
// reported (vout != 0), the verdict bits are those found so far.
// Steering angle of step k for the generated families whose steering does not depend on the rolled state.
template <int CAND>
__device__ __forceinline__ double steer_next(const KP& P, const Scenario<double>& S, int k, double ddf, double df) {
    if (CAND == CAND_LATTICE) return clampd(df + ddf, -P.df_max, P.df_max);
    double ba, bdf;                                                   // CAND_RAMP_HOLD
    ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
    const double tdf = clampd(bdf + ddf, -P.df_max, P.df_max);
    return clampd(df + clampd(tdf - df, -P.rate_df, P.rate_df), -P.df_max, P.df_max);
}
// ddf of steering column j (lattice increment / ramp-hold target offset) -- as rollout_one derives it from the candidate
template <int CAND>
__device__ __forceinline__ double steer_column(const KP& P, const Scenario<double>& S, int j) {
    if (CAND == CAND_LATTICE) return -P.rate_df + (2 * P.rate_df) * (double)j / (double)(P.G - 1);
    return S.cpar[1] + cand_m(j, P.G, P.refine_it == 0) * S.cpar[3];
}
// (sin, cos) of the slip angle beta = atan(r tan df), r = l_r/(l_f+l_r):  cos = c/n, sin = r s/n, n = sqrt(c^2 + r^2 s^2)
template <int CAND>
__device__ __forceinline__ void slip_trig(const KP& P, double lr_ratio, double df, double& sb, double& cb) {
    double sdf, cdf;
    if (CAND != CAND_TABLE && P.df_small) sincos_kernel(df, sdf, cdf);       // |df| <= df_max < pi/4
    else sincos_reduced(df, sdf, cdf);
    const double n = rsq_nr(fma((lr_ratio * lr_ratio) * sdf, sdf, cdf * cdf));   // argument in [r^2, 1]
    cb = cdf * n;
    sb = lr_ratio * sdf * n;
}
// Steering table of one search unit (lattice, ramp-hold): the lanes of a 64-candidate steering slice share G/W steering
// columns, each rolled by 64 W/G lanes with the same (df_k, sin beta_k, cos beta_k) -- ~50 instructions per lane and
// step that depend on nothing but the column.  The unit's wave lays the columns out once in LDS, [k][column][df, sin,
// cos] (the angles sequentially on one lane per column -- the recursion is the candidates' own -- then the trigonometry
// of all entries in parallel), with the same functions the untabulated roll-out calls: the same bits.
constexpr int STAB_MAX_ENTRIES = STEER_TABLE_MAX_ENTRIES;        // columns x N per unit (C = 256: 4 x 20, N = 40: 160; C = 64: 8 x 40)
template <int CAND>
__device__ __forceinline__ void fill_steer_table(const KP& P, const Scenario<double>& S, int nj, int r_first, int lane,
                                                 double lr_ratio, double* __restrict__ stab) {
    if (lane < nj) {                                                              // nj columns from rank r_first on (< G)
        const int r = r_first + lane;
        const int j = (r & 1) ? P.G / 2 - 1 - (r >> 1) : P.G / 2 + (r >> 1);      // unit_candidate's column order
        const double ddf = steer_column<CAND>(P, S, j);
        double df = S.df_prev;
        for (int k = 0; k < P.N; ++k) {
            df = steer_next<CAND>(P, S, k, ddf, df);
            stab[(k * nj + lane) * 3] = df;
        }
    }
    __syncthreads();
    for (int e = lane; e < nj * P.N; e += 64) {
        double sb, cb;
        slip_trig<CAND>(P, lr_ratio, stab[e * 3], sb, cb);
        stab[e * 3 + 1] = sb;
        stab[e * 3 + 2] = cb;
    }
    __syncthreads();
}

// ---- the incumbent bound (search, progress cost; used by the tracking family, whose candidates hardly ever fail a verdict) ----
// A candidate's cost is  J = (non-negative stage terms, mpc.py:361-364) - (s_N - s_0)  (mpc.py:372), and the progress still to
// come after step k is bounded by the speed profile of the candidate's own acceleration row, which depends on nothing else:
// every RK4 stage derivative is  ds = v cos(.) / (1 - K ey) <= |v| / (1 - |K| |ey|),  the stage speeds of a control step lie
// between v_k and v_k+1, so  s_k+1 - s_k <= lam dt max(|v_k|, |v_k+1|)  with lam = 1 when the scenario cannot meet its arc within
// the horizon (K == 0 at every stage argument) and 1 / (1 - |kv| ey_b) else, ey_b bounding |ey| at any stage of a candidate that
// is feasible at every node (|ey_k| <= ey_lim + tol, mpc.py:296-299; it moves by at most |v| per unit time in between).  Hence
//     LB_k = J_k - (s_k - s_0) - rem_k,   rem_k = lam dt sum_{k' >= k} max(|v_k'|, |v_k'+1|)
// is a lower bound of the final cost of a candidate that ends up feasible, and one with LB_k > J_inc -- the final cost of a
// FEASIBLE candidate of the same scenario and the same pass, left by a unit that has finished (igt_kernels_f64.hip:
// search_unit64 publishes with an atomic min, units run highest acceleration rows first) -- cannot win and cannot tie: it is
// marked lost.  The winner is never among them, so cost / arg-min / trajectory are what they were, bit for bit; only HOW MANY
// steps a unit rolls depends on which incumbents it saw (tools/bound_prune_probe.py: 0.84 -> 0.46 of the wave-steps).
// (VIOL_PRUNED, cost_key / cost_of_key, progress_slack: igt_device.h -- the float path prunes the same way)

// XY = false (search only, decided per unit by obstacles_out_of_reach): x, y are neither integrated nor judged -- no candidate
// that holds the speed box can come within d_min of any forecast position, and one that does not is infeasible already.
// inc (search, CAND_TRACK, progress cost; may be null): the scenario's incumbent key, see above.
// ---- horizon checkpoints (SEGMODE): the winner's trajectory in pieces ----
// emit rolls the winner again (keeping 147 state values per candidate alive in the search would cost more), and one roll-out is
// a serial chain: 51 us at N = 20, all of it exposed when solves do not overlap.  The search pass therefore leaves, at the three
// steps k = i N / 4, what a roll-out needs to RESUME there exactly: (s, ey, epsi), the carried pair (sin, cos)(epsi + beta_k-1)
// and -- tracking family, whose steering is a feedback on the rolled state -- (df_k-1, sin beta_k-1, cos beta_k-1); everything
// else of the state at node k is a function of the candidate alone (a and v: the row's recurrence; df of the lattice / ramp-hold /
// table families: the column's) and is replayed.  SEGMODE 1 (search): every lane writes its values to its LDS slots `ck`
// (field-major, stride 64: igt_kernels_f64.hip search_unit64 copies the unit winner's to HBM).  SEGMODE 2 (emit_seg_f64_kernel):
// rolls steps [seg_k0, seg_k1) from the record `ck` (stride 1) of the checkpoint at seg_k0 -- the same statements on the same
// numbers, so the four pieces are the unsegmented roll-out bit for bit.  x, y, psi feed nothing back and are rolled on their own
// from the controls (cartesian_rows below).
constexpr int CK_FIELDS = 8, CK_PARTS = 4;                      // record: s ey epsi s1 c1 [df sb cb]; pieces of the horizon
__host__ __device__ inline int ckpt_step(int N, int i) { return (i * N) / CK_PARTS; }      // first step of piece i

template <int CAND, bool HI_ORDER, bool BOOK, bool UNIFORM, class Sink, bool EARLY_EXIT = false, bool STAB = false, int NRK = 0,
          bool XY = true, int SEGMODE = 0>
__device__ __forceinline__ void rollout_one(const KP& P, const Scenario<double>& S, int cidx,
                                            const double* __restrict__ table, const double* __restrict__ cinf,
                                            Sink& sink, double& Jout, unsigned& vout, double& sN, double& vN,
                                            const double* __restrict__ stab = nullptr, int stab_stride = 0,
                                            const unsigned long long* inc = nullptr, double* ck = nullptr, int seg_k0 = 0,
                                            int seg_k1 = 0, const double* __restrict__ rem_rows = nullptr) {
    constexpr bool BOUND = CAND == CAND_TRACK && BOOK && UNIFORM && EARLY_EXIT;
    // search on units of live acceleration rows (igt_kernels_f64.hip accel_rows_kernel; launch-time bit 30 of KP::dev): the speed
    // box and the terminal set read the row's (a, v) recurrence alone and were judged there, with these statements -- every lane
    // that holds a candidate holds one of a row that passed: not judged again (74 half-planes per lane at the last step)
    const bool rows_judged = BOOK && EARLY_EXIT && UNIFORM && CAND != CAND_TABLE && (P.dev & (1 << 30)) != 0;
    constexpr int CKF = CAND == CAND_TRACK ? 8 : 5;            // fields this family writes
    constexpr bool KEEP_PSI = Sink::kKeepsStates;
    // search only needs feasible-or-not: |ey|, box v and collision are folded into one running maximum, compared with
    // the tolerance when it is read (x > tol for some x  <=>  max x > tol; a NaN operand is ignored by both forms)
    constexpr bool LEAN = BOOK && EARLY_EXIT;
    typedef Fast64<HI_ORDER, NRK> FP;
    FP fp;
    fp.init(P, S.b0, S.b1, S.kv);
    double x = S.x0[0], y = S.x0[1], s = S.x0[2], ey = S.x0[3], ep = S.x0[4], v = S.x0[5], psi = S.x0[6];
    double a = S.a_prev, df = S.df_prev, da = 0.0, ddf = 0.0, J = 0.0, gmax = -1.0e300;
    // cidx < 0 (search): a lane of the unit that holds no candidate (igt_kernels_f64.hip unit_candidate) -- it rolls candidate 0
    // in step with the wave and is "lost" from the start, so it neither wins nor keeps the unit alive
    unsigned viol = cidx < 0 ? (unsigned)VIOL_EY : 0u;
    if (cidx < 0) cidx = 0;
    bool dead = false;
    if (CAND == CAND_LATTICE) {
        // da_i = -ra + (2 ra) i/(G-1), ddf_j likewise (oracle candidates_lattice)
        const int i = cidx / P.G, j = cidx - i * P.G;
        da = -P.rate_a + (2 * P.rate_a) * (double)i / (double)(P.G - 1);
        ddf = steer_column<CAND>(P, S, j);
    } else if (CAND == CAND_RAMP_HOLD || CAND == CAND_TRACK) {
        // da / ddf hold the OFFSETS of the targets from the base sequence here (igt_device.h cand_m, ramp_base);
        // CAND_TRACK: ddf is the slip-angle offset of the steering feedback (track_steer)
        const int i = cidx / P.G, j = cidx - i * P.G;
        da = S.cpar[0] + cand_m(i, P.G, P.refine_it == 0) * S.cpar[2];
        ddf = S.cpar[1] + cand_m(j, P.G, P.refine_it == 0) * S.cpar[3];
    }
    if (SEGMODE != 2 || seg_k0 == 0) sink.state(0, 0, S.x0);
    typename FP::Work w;
    w.d0 = w.d1 = 0.0;
    if (XY) sincos_reduced(S.x0[6], w.s2, w.c2);       // carried as (sin,cos)(psi + beta_k); beta_{-1} = 0
    else { w.s2 = 0.0; w.c2 = 1.0; }
    // (sin,cos)(epsi + beta_k) is carried the same way: the sub-steps turn the pair by exactly the angle epsi advances by
    // (substep(): h w2 - corr per sub-step), so after control step k-1 it holds (sin,cos)(epsi_k + beta_{k-1}) and step k
    // turns it by beta_k - beta_{k-1} -- no sincos(epsi) per control step (39 of ~460 instructions on a straight route)
    double cb_prev = 1.0, sb_prev = 0.0;
    double trk_sb = 0.0, trk_cb = 1.0;
    bool trk_followed = false;
    int k_first = 0, k_last = P.N;
    if (SEGMODE == 2) { k_first = seg_k0; k_last = seg_k1; }
    if (SEGMODE == 2 && seg_k0 > 0) {
        // resume at node seg_k0: the controls' own recurrences are replayed (the statements of the loop below), the state that
        // depends on the roll-out comes from the checkpoint
        for (int k = 0; k < seg_k0; ++k) {
            if (CAND == CAND_LATTICE) {
                a = clampd(a + da, P.a_min, P.a_max);
                df = steer_next<CAND>(P, S, k, ddf, df);
            } else if (CAND == CAND_RAMP_HOLD || CAND == CAND_TRACK) {
                double ba, bdf;
                ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
                if (CAND == CAND_TRACK) {
                    a = track_accel_next(P, k, ba, da, v, a);
                } else {
                    const double ta = clampd(ba + da, P.a_min, P.a_max);
                    a = clampd(a + clampd(ta - a, -P.rate_a, P.rate_a), P.a_min, P.a_max);
                }
                if (CAND == CAND_RAMP_HOLD) df = steer_next<CAND>(P, S, k, ddf, df);
            } else {
                a = table[((size_t)cidx * 2 + 0) * P.N + k];
                df = table[((size_t)cidx * 2 + 1) * P.N + k];
            }
            v = fma(fp.dt, a, v);
        }
        s = ck[0]; ey = ck[1]; ep = ck[2]; w.s1 = ck[3]; w.c1 = ck[4];
        if (CAND == CAND_TRACK) { df = ck[5]; sb_prev = ck[6]; cb_prev = ck[7]; }
        else slip_trig<CAND>(P, fp.lr_ratio, df, sb_prev, cb_prev);      // of df_{k0-1}: what step k0-1 left in (sb_prev, cb_prev)
    } else {
        sincos_reduced(S.x0[4], w.s1, w.c1);
    }
    int ck_q = 1, ck_k = (SEGMODE == 1 && ck) ? ckpt_step(P.N, 1) : -1;  // next checkpoint and its step (search; none without slots)
    // incumbent bound: rem = bound of the progress still to come (the row's own (a, v) recurrence rolled ahead, the statements of
    // the loop below), jcut = incumbent + a margin far above the rounding of LB_k (1e-9: the comparison is mathematically strict)
    double rem = 0.0, jcut = (double)INFINITY, seg_scale = 0.0;
    if (BOUND && inc) {
        seg_scale = progress_slack(P, S) * P.dt;
        if (seg_scale > 0.0) {
            if (rem_rows) {          // the row's sum as accel_rows_kernel left it (the statements of the loop in the other branch)
                rem = rem_rows[cidx / P.G];
            } else {
                double a2 = S.a_prev, v2 = S.x0[5];
                for (int k = 0; k < P.N; ++k) {
                    double ba, bdf;
                    ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
                    a2 = track_accel_next(P, k, ba, da, v2, a2);
                    const double vn = fma(fp.dt, a2, v2);
                    rem += fmax(fabs(v2), fabs(vn));
                    v2 = vn;
                }
            }
            rem *= seg_scale * (1.0 + 1e-12);
        } else {
            inc = nullptr;
        }
    }

    for (int k = k_first; k < k_last; ++k) {
        if (SEGMODE == 1 && k == ck_k) {          // node k: what a roll-out needs to resume here
            double* c = ck + (size_t)(ck_q - 1) * CKF * 64;
            c[0] = s; c[64] = ey; c[128] = ep; c[192] = w.s1; c[256] = w.c1;
            if (CAND == CAND_TRACK) { c[320] = df; c[384] = sb_prev; c[448] = cb_prev; }
            ++ck_q;
            ck_k = ck_q < CK_PARTS ? ckpt_step(P.N, ck_q) : -1;
        }
        // ---- controls of step k
        if (CAND == CAND_LATTICE) {
            a = clampd(a + da, P.a_min, P.a_max);
            df = STAB ? stab[k * stab_stride] : steer_next<CAND>(P, S, k, ddf, df);
        } else if (CAND == CAND_RAMP_HOLD) {
            double ba, bdf;
            ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
            const double ta = clampd(ba + da, P.a_min, P.a_max);
            a = clampd(a + clampd(ta - a, -P.rate_a, P.rate_a), P.a_min, P.a_max);
            df = STAB ? stab[k * stab_stride] : steer_next<CAND>(P, S, k, ddf, df);
        } else if (CAND == CAND_TRACK) {
            double ba, bdf;
            ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
            a = track_accel_next(P, k, ba, da, v, a);
            df = track_steer(P, df, ey, ep, ddf, &trk_sb, &trk_cb, &trk_followed);
        } else {
            const double an = table[((size_t)cidx * 2 + 0) * P.N + k];
            const double dn = table[((size_t)cidx * 2 + 1) * P.N + k];
            if (BOOK) {   // input-rate (mpc.py:301-312, u_{-1} = u_prev) and input box (mpc.py:318-321)
                if (fmax(fabs(an - a) - P.rate_a, fabs(dn - df) - P.rate_df) > P.tol) viol |= VIOL_RATE;
                if (fmax(fmax(P.a_min - an, an - P.a_max), fmax(-P.df_max - dn, dn - P.df_max)) > P.tol) viol |= VIOL_BOX_U;
            }
            a = an; df = dn;
        }
        sink.ctrl(0, k, a, df);
        double sb, cb;                               // (sin, cos)(beta), beta = atan(r tan df)
        if (STAB && (CAND == CAND_LATTICE || CAND == CAND_RAMP_HOLD)) {
            sb = stab[k * stab_stride + 1];
            cb = stab[k * stab_stride + 2];
        } else if (CAND == CAND_TRACK) {
            // a lane whose steering followed the command has beta = beta_cmd: its (sin, cos) are known already.  The other
            // lanes (rate- or box-limited) go through sincos(df); the wave skips that when no lane needs it -- which lane
            // takes which value does not depend on the vote, so search, emit and rollout-all agree bit for bit
            sb = trk_sb; cb = trk_cb;
            if (!__all(trk_followed)) {
                double sb2, cb2;
                slip_trig<CAND>(P, fp.lr_ratio, df, sb2, cb2);
                sb = trk_followed ? sb : sb2;
                cb = trk_followed ? cb : cb2;
            }
        } else {
            slip_trig<CAND>(P, fp.lr_ratio, df, sb, cb);
        }
        sink.slip(0, k, sb, cb);
        const double sblr = sb * fp.inv_lr;
        // ---- bookkeeping of state k (cost in the oracle's order: control effort, epsi^2, ey^2 -- mpc.py:361-364)
        if (BOOK) {
            J = J + P.w_u * (a * a + df * df);
            J = J + ep * ep;
            J = J + ey * ey;
            if (LEAN) {
                gmax = fmax(gmax, fabs(ey) - P.ey_lim);                              // mpc.py:296-299
                if (!rows_judged) gmax = fmax(gmax, fmax(P.v_min - v, v - P.v_max)); // mpc.py:316-317 (k < N)
            } else {
                if (fabs(ey) - P.ey_lim > P.tol) viol |= VIOL_EY;
                if (fmax(P.v_min - v, v - P.v_max) > P.tol) viol |= VIOL_BOX_V;
            }
            if (k == P.N - 1 && !rows_judged) viol |= terminal_viol(P, v, a, cinf);  // mpc.py:177-180
        }
        w.ey = ey; w.v1 = v;
        // unused on straight routes; with one scenario per lane the votes of substeps() read them on every lane
        // (a straight route's break-points are +inf: d = -inf, "clear")
        if (!UNIFORM || fp.kv != 0.0) { w.d0 = s - fp.b0; w.d1 = s - fp.b1; }
        if (BOUND && inc) {
            // the incumbent is re-read every fourth step (a unit that started before its scenario's first unit finished picks it
            // up on the way); agent scope: the units of a scenario may run on different XCDs, whose L2s are not coherent
            if ((k & 3) == 0) {
                const double ji = cost_of_key(__hip_atomic_load(inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                jcut = ji + 1e-9 * (1.0 + fabs(ji));
            }
            if ((J - (s - S.x0[2])) - rem > jcut) viol |= VIOL_PRUNED;
        }
        if (BOOK && UNIFORM && EARLY_EXIT) {
            // search only: once every candidate of the slice has failed a verdict, nothing rolled further can win
            const bool lost = (viol != 0) | (LEAN && gmax > P.tol);
            if (__all(lost) && !(P.dev & 2)) { dead = true; break; }
        }
        if (BOOK && XY && k >= 1) {                                                  // collision, mpc.py:223-226
            for (int o = 0; o < P.n_obs; ++o) {
                const double dx = x - S.obs[(o * 2 + 0) * (P.N + 1) + k], dy = y - S.obs[(o * 2 + 1) * (P.N + 1) + k];
                const double g = P.dmin2 - (dx * dx + dy * dy);
                if (LEAN) gmax = fmax(gmax, g);
                else if (g > P.tol) viol |= VIOL_COLLISION;
            }
        }
        // (sin,cos)(psi + beta_k) from (psi + beta_{k-1}), and (sin,cos)(epsi + beta_k) from (epsi + beta_{k-1}): rotate by
        // beta_k - beta_{k-1}, re-normalise (first order: the pairs are within rounding of unit length)
        {
            const double sdb = fma(sb, cb_prev, -(cb * sb_prev));
            const double cdb = fma(cb, cb_prev, sb * sb_prev);
            rotate(w.s1, w.c1, sdb, cdb);
            const double r1 = fma(fma(w.s1, w.s1, w.c1 * w.c1), -0.5, 1.5);
            w.s1 *= r1; w.c1 *= r1;
            if (XY) {
                rotate(w.s2, w.c2, sdb, cdb);
                const double r2 = fma(fma(w.s2, w.s2, w.c2 * w.c2), -0.5, 1.5);
                w.s2 *= r2; w.c2 *= r2;
            }
            cb_prev = cb; sb_prev = sb;
        }
        w.acc_s = 0.0; w.acc_ey = 0.0; w.acc_ep = 0.0; w.acc_x = 0.0; w.acc_y = 0.0; w.acc_psi = 0.0;
        fp.template substeps<UNIFORM, XY>(a, sblr, w);
        s += w.acc_s; ey += w.acc_ey; ep += w.acc_ep;
        if (XY) { x += w.acc_x; y += w.acc_y; }
        if (KEEP_PSI) psi += w.acc_psi;        // psi feeds nothing back (search: dead code)
        if (BOUND && inc) rem -= seg_scale * fmax(fabs(v), fabs(fma(fp.dt, a, v)));      // this step's share of the bound is spent
        v = fma(fp.dt, a, v);
        const double nxt[7] = {x, y, s, ey, ep, v, psi};
        sink.state(0, k + 1, nxt);
    }
    if (dead) {            // costs are meaningless; "failed" is what is reported
        Jout = 0.0; vout = viol | ((LEAN && gmax > P.tol) ? VIOL_EY : 0u); sN = 0.0; vN = 0.0;
        return;
    }
    if (BOOK) {
        J = J + ep * ep;
        J = J + ey * ey;
        if (LEAN) gmax = fmax(gmax, fabs(ey) - P.ey_lim);
        else if (fabs(ey) - P.ey_lim > P.tol) viol |= VIOL_EY;
        for (int o = 0; XY && o < P.n_obs; ++o) {
            const double dx = x - S.obs[(o * 2 + 0) * (P.N + 1) + P.N], dy = y - S.obs[(o * 2 + 1) * (P.N + 1) + P.N];
            const double g = P.dmin2 - (dx * dx + dy * dy);
            if (LEAN) gmax = fmax(gmax, g);
            else if (g > P.tol) viol |= VIOL_COLLISION;
        }
        if (LEAN && gmax > P.tol) viol |= VIOL_EY;     // lean form: "some state-side verdict failed"
        if (!(fabs(x) < 1e300 && fabs(y) < 1e300 && fabs(s) < 1e300 && fabs(ey) < 1e300 && fabs(ep) < 1e300 &&
              fabs(psi) < 1e300))
            viol |= VIOL_NONFINITE;
        sN = s; vN = v; Jout = J; vout = viol;
    } else {
        Jout = 0.0; vout = 0; sN = 0.0; vN = 0.0;
    }
}

// The Cartesian rows x, y, psi of a trajectory from its controls alone (emit_seg_f64_kernel): they feed nothing back
// (frenet.py:84-90), and what they read -- the stage speeds, the psi offsets, (sin, cos)(psi + beta) carried by rotation --
// depends on (a_k, sin beta_k, cos beta_k) only.  The statements are rollout_one's: the pair is turned and re-normalised as
// there, the step's constants come from step_const, and the sub-steps are substep<1, true> itself (its Frenet part reads
// nothing of the Cartesian one and is dead code here) -- so x, y, psi come out bit for bit as in the full roll-out, whatever
// sub-step variant that took (the variants differ in the Frenet part only).
// ctl(k, a, sb, cb): the controls of step k -- read back from where the Frenet pieces left them (tracking family: the steering is
// theirs to decide) or generated on the spot (the families whose controls are a function of the candidate alone: StepControls);
// X, Y, PSI: [N+1] outputs with element stride `xs` (node 0 is written by the caller).
template <bool HI_ORDER, int NRK, class Ctl>
__device__ __forceinline__ void cartesian_rows(const KP& P, double x, double y, double psi, double v, Ctl& ctl,
                                               double* X, double* Y, double* PSI, int xs) {
    typedef Fast64<HI_ORDER, NRK> FP;
    FP fp;
    fp.init(P, (double)INFINITY, (double)INFINITY, 0.0);
    typename FP::Work w;
    w.d0 = w.d1 = 0.0; w.s1 = 0.0; w.c1 = 1.0; w.ey = 0.0;
    sincos_reduced(psi, w.s2, w.c2);
    double cb_prev = 1.0, sb_prev = 0.0;
    for (int k = 0; k < P.N; ++k) {
        double a, sb, cb;
        ctl(k, a, sb, cb);
        const double sblr = sb * fp.inv_lr;
        w.v1 = v;
        {
            const double sdb = fma(sb, cb_prev, -(cb * sb_prev));
            const double cdb = fma(cb, cb_prev, sb * sb_prev);
            rotate(w.s2, w.c2, sdb, cdb);
            const double r2 = fma(fma(w.s2, w.s2, w.c2 * w.c2), -0.5, 1.5);
            w.s2 *= r2; w.c2 *= r2;
            cb_prev = cb; sb_prev = sb;
        }
        w.acc_s = 0.0; w.acc_ey = 0.0; w.acc_ep = 0.0; w.acc_x = 0.0; w.acc_y = 0.0; w.acc_psi = 0.0;
        const typename FP::StepConst sc = fp.step_const(a, sblr, w);
        fp.template run<1, true>(sc, w);
        x += w.acc_x; y += w.acc_y; psi += w.acc_psi;
        v = fma(fp.dt, a, v);
        X[(size_t)(k + 1) * xs] = x; Y[(size_t)(k + 1) * xs] = y; PSI[(size_t)(k + 1) * xs] = psi;
    }
}

// The controls of the families whose candidates do not read the rolled state (lattice, ramp-hold, table), step by step, with
// (sin, cos)(beta_k): rollout_one's statements for them (its control block and slip_trig), for the wave that rolls the Cartesian
// rows beside the Frenet pieces.
template <int CAND>
struct StepControls {
    const KP& P;
    const Scenario<double>& S;
    const double* table;
    int cidx;
    double a, df, da, ddf, lr_ratio;
    __device__ __forceinline__ StepControls(const KP& P_, const Scenario<double>& S_, int c, const double* t)
        : P(P_), S(S_), table(t), cidx(c), a(S_.a_prev), df(S_.df_prev), da(0.0), ddf(0.0), lr_ratio(P_.lr_ratio) {
        const int i = cidx / P.G, j = cidx - i * P.G;
        if (CAND == CAND_LATTICE) {
            da = -P.rate_a + (2 * P.rate_a) * (double)i / (double)(P.G - 1);
            ddf = steer_column<CAND>(P, S, j);
        } else if (CAND == CAND_RAMP_HOLD) {
            da = S.cpar[0] + cand_m(i, P.G, P.refine_it == 0) * S.cpar[2];
            ddf = S.cpar[1] + cand_m(j, P.G, P.refine_it == 0) * S.cpar[3];
        }
    }
    __device__ __forceinline__ void operator()(int k, double& a_out, double& sb, double& cb) {
        if (CAND == CAND_LATTICE) {
            a = clampd(a + da, P.a_min, P.a_max);
            df = steer_next<CAND>(P, S, k, ddf, df);
        } else if (CAND == CAND_RAMP_HOLD) {
            double ba, bdf;
            ramp_base<double>(S.ws, P.N, k, S.a_prev, S.df_prev, ba, bdf);
            const double ta = clampd(ba + da, P.a_min, P.a_max);
            a = clampd(a + clampd(ta - a, -P.rate_a, P.rate_a), P.a_min, P.a_max);
            df = steer_next<CAND>(P, S, k, ddf, df);
        } else {
            a = table[((size_t)cidx * 2 + 0) * P.N + k];
            df = table[((size_t)cidx * 2 + 1) * P.N + k];
        }
        slip_trig<CAND>(P, lr_ratio, df, sb, cb);
        a_out = a;
    }
};

}  // namespace f64
}  // namespace igt
