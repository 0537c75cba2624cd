// igt_device.h -- device-side arithmetic of the batched shooting solver (gfx950).
//
// Mapping: a 64-lane wavefront owns one scenario (float path: one 128-candidate
// slice of it); lanes roll different candidates, so every per-scenario input is wave-uniform and lives in
// SGPRs, the recurrence over the horizon runs in VGPRs, and the arg-min over candidates is one 6-step wave
// butterfly.
//
// This header holds what both precisions share (kernel parameters, candidate generation, verdict helpers)
// and the parity path: ExactStepper<T> performs one control step of the reference's RK4 Frenet bicycle
// model (kinematic_bicycle_model_frenet.py:70-127; casadi twin 129-185 used by mpc.py:201-209) operation for
// operation as the float64 oracle does, and rollout_pass evaluates cost (mpc.py:356-373) and constraints
// (mpc.py:177-180, 223-226, 296-321) in double on the step-boundary states.  The float32 production path
// (two candidates per lane as packed pairs) is igt_fast.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "igt_math64.h"

// Developer kernels (A/B timing and cross-checks, selected at run time through IGT_DEV_FLAGS): the 3-waves-per-SIMD builds
// of the search kernels (32), the oracle-order float64 kernels (1024) and the literal north_star mapping (2048).  The
// shipped libigtmpc.so is built without them (IGT_DEV_KERNELS = 0: igt_create refuses those flags); libigtmpc_dev.so, which
// the tests that compare against them load, is the same source with IGT_DEV_KERNELS = 1.
#ifndef IGT_DEV_KERNELS
#define IGT_DEV_KERNELS 0
#endif
#define IGT_DEV_KERNEL_FLAGS (32 | 1024 | 2048)

namespace igt {

struct KP {  // kernel parameters (by value -> SGPRs)
    int N, n_rk4, C, n_obs, cand_mode, cost_mode, F, G, hi_order, refine_it;
    int df_small;   // df_max < pi/4: generated steering angles need no range reduction
    // developer switches (env IGT_DEV_FLAGS; A/B measurements only, results do not depend on them):
    //   1 slices along the acceleration axis, 2 no early exit, 4 no steering table (f64 search), 8 / 32 force 2 / 3 search waves per SIMD,
    //   16 no longest-first queue order, 256 unit trace (with IGT_DEV_TRACE=<file>), 512 no stealing between queues,
    //   bits 12-15: items per wave at the end of a queue whose index is fetched late (default 4)
    int dev;
    double dt, h, l_r, lr_ratio, v_min, v_max, a_min, a_max, df_max;
    double rate_a, rate_df, ey_lim, dmin2, w_u, tol;
    double trk_ke, trk_span, trk_blim;      // IGT_CAND_TRACK: lateral gain [1/m], span of the slip-angle offsets, |beta| limit
    double trk_env;                         // IGT_CAND_TRACK: slope of the acceleration envelope (+inf = none), track_accel_target_uncapped
    double trk_vmax, inv_dt, inv_rate_a;    // IGT_CAND_TRACK: the speed the cap of the targets looks ahead to (+inf = no cap); 1 / dt, 1 / rate_a
};

enum { CAND_LATTICE = 0, CAND_TABLE = 1, CAND_RAMP_HOLD = 2, CAND_TRACK = 3 };
enum { VIOL_BOX_V = 1, VIOL_BOX_U = 2, VIOL_RATE = 4, VIOL_EY = 8, VIOL_TERMINAL = 16,
       VIOL_COLLISION = 32, VIOL_NONFINITE = 64 };

// ---------------------------------------------------------------------------------------
// candidate control sequences (always generated in double so that u_out is the oracle's U)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
    return fmin(fmax(x, lo), hi);
}

struct Ctl {
    double a, df;        // last generated control
    double da, ddf;      // lattice increments  /  ramp-hold: offsets of the targets from the base sequence
};
// Tracking family (IGT_CAND_TRACK): the acceleration of candidate (i, j) is the ramp-hold one (offset i from the base
// sequence); its STEERING is a state feedback evaluated inside the roll-out, so that every acceleration profile gets
// the steering that belongs to where it actually is (an open-loop steering sequence is timed, not placed: a candidate
// that brakes reaches the arc later but would steer at the same step).  With the slip angle beta = atan(r tan df):
//     e_y' = v sin(beta + epsi)   ->   beta_cmd,k = clamp(-epsi_k - k_e e_y,k + off_j, +-beta_lim),
//     df_cmd = atan(tan(beta_cmd) / r),   df_k = df_{k-1} + clamp(df_cmd - df_{k-1}, +-rate)   (mpc.py:301-312)
// off_j = c_b + m(u_j) span_b are the G offsets (dense around 0; re-centred by refinement passes like the ramp-hold
// ones).  The realised (a_k, df_k) are an ordinary control sequence: u_out, cost and verdicts are those of the roll-out.
// sbt, cbt (optional outputs): (sin, cos) of the commanded slip angle.  When the step is limited neither by the steering rate
// nor by the box -- *followed = true -- the steering angle IS the command and the slip angle it produces, atan(r tan df),
// IS beta_cmd: the roll-out takes (sbt, cbt) as its (sin, cos)(beta) and skips sincos(df) and the normalisation
// (igt_fast64.h; a wave whose lanes all follow skips them altogether).
__device__ __forceinline__ double track_steer(const KP& P, double df_prev, double ey, double ep, double off,
                                              double* sbt_out = nullptr, double* cbt_out = nullptr, bool* followed = nullptr) {
    const double beta = clampd(-ep - P.trk_ke * ey + off, -P.trk_blim, P.trk_blim);
    // cmd = atan(tan(beta) / r) = atan2(sin beta, r cos beta); cos beta > 0 (beta_lim < pi/2, igt_api.hip)
    double sbt, cbt;
    if (P.trk_blim < m64::QUADRANT0) m64::sincos_kernel(beta, sbt, cbt);
    else m64::sincos_reduced(beta, sbt, cbt);
    const double cmd = m64::atan2_xpos(sbt, P.lr_ratio * cbt);
    const double df = clampd(df_prev + clampd(cmd - df_prev, -P.rate_df, P.rate_df), -P.df_max, P.df_max);
    if (sbt_out) { *sbt_out = sbt; *cbt_out = cbt; *followed = df == cmd; }
    return df;
}

// Acceleration target of the tracking family at step k.  One more unit of a_k buys dt (T - t_k - dt/2) of progress
// (the -(s_N - s_0) term of the cost, mpc.py:372) and costs 2 w_u a_k (mpc.py:362): no unconstrained optimum asks for
// more than E_k = dt^2 (N - k - 1/2) / (2 w_u), so the candidates' targets stay under that envelope -- a candidate
// with a large offset ramps up at the jerk limit until it meets E_k and follows it down, which is the shape the
// NLP's optimum has (tools/nlp_gap.py).  trk_env = track_env dt^2 / (2 w_u) (host, igt_api.hip), +inf when off.
// Speed cap of the targets (oracle: np_oracle.track_speed_cap).  The speed is v_{k+1} = v_k + dt a_k and the acceleration
// comes down by at most r = dt jerk per step (mpc.py:301-304): from a_k the speed still gains dt S(a_k),
// S(a) = (n + 1) a - r n (n + 1) / 2 with n = floor(a / r), before a reaches 0.  The largest a_k with v_k + dt S(a_k) <= v_max
// (mpc.py:316-317) is, with D = (v_max - v_k) / dt:  n = floor((sqrt(1 + 8 D / r) - 1) / 2),  a = D / (n + 1) + r n / 2
// (continuous in D, so a last-bit difference in the square root moves nothing; D < 0: a = D, one step back under the limit).
// A candidate with a large offset therefore accelerates as hard as the limits allow and eases off so as to arrive at v_max
// with a = 0 -- the NLP optimum's longitudinal shape (profiles/r04_nlp_gap.txt) -- where it used to fail the speed
// box and leave the winner to a lower row.  D is clipped to 1000 (far from the cap, or no cap: trk_vmax = +inf).
// n comes from a float square root and is then set right by the two float64 comparisons that define it (r n (n + 1) / 2 <= D <
// r (n + 1) (n + 2) / 2); the quotient is rcp_nr's (1 ulp).
__device__ __forceinline__ double track_speed_cap(const KP& P, double v) {
    const double D = fmin((P.trk_vmax - v) * P.inv_dt, 1000.0);
    const double Dp = fmax(D, 0.0);
    const double hr = 0.5 * P.rate_a;
    double n = (double)floorf((__builtin_sqrtf(1.0f + (float)(Dp * (8.0 / P.rate_a))) - 1.0f) * 0.5f);
    n = fmax(n, 0.0);
    if (hr * n * (n + 1.0) > Dp) n -= 1.0;
    else if (hr * (n + 1.0) * (n + 2.0) <= Dp) n += 1.0;
    return D < 0.0 ? D : fma(Dp, m64::rcp_nr(n + 1.0), hr * n);
}
__device__ __forceinline__ double track_accel_target_uncapped(const KP& P, int k, double base, double off) {
    return clampd(fmin(base + off, P.trk_env * ((double)(P.N - k) - 0.5)), P.a_min, P.a_max);
}
// a_k of the tracking family: one jerk-limited step (mpc.py:301-304) from a_{k-1} towards min(envelope target, speed cap).  The step towards
// the UNCAPPED target is tried first: if it leaves the speed inside the cap -- S(a_try) <= D, the cap's own inequality, a
// product and a floor -- the cap, being no smaller than a_try, would have changed nothing (a target above the cap is then out
// of the step's reach anyway), and the wave skips the square root and the quotient unless one of its lanes needs them.
// Which lanes take which branch decides nothing: both give the same a_k where both apply.
__device__ __forceinline__ double track_accel_next(const KP& P, int k, double base, double off, double v, double a_prev) {
    const double tu = track_accel_target_uncapped(P, k, base, off);
    const double a_try = clampd(a_prev + clampd(tu - a_prev, -P.rate_a, P.rate_a), P.a_min, P.a_max);
    const double D = fmin((P.trk_vmax - v) * P.inv_dt, 1000.0);
    const double n = fmax(floor(a_try * P.inv_rate_a), 0.0);
    const double S = fma(n + 1.0, a_try, -(0.5 * P.rate_a) * n * (n + 1.0));
    if (__all(S <= D)) return a_try;
    const double ta = clampd(fmin(tu, track_speed_cap(P, v)), P.a_min, P.a_max);
    const double a_cap = clampd(a_prev + clampd(ta - a_prev, -P.rate_a, P.rate_a), P.a_min, P.a_max);
    return S <= D ? a_try : a_cap;
}

// base sequence of the ramp-hold targets at step k
template <typename T>
__device__ __forceinline__ void ramp_base(const T* __restrict__ ws, int N, int k, double a_prev, double df_prev, double& ba,
                                          double& bdf) {
    ba = ws ? (double)ws[k] : a_prev;
    bdf = ws ? (double)ws[N + k] : df_prev;
}

// Ramp-and-hold family (IGT_CAND_RAMP_HOLD): candidate (i, j) tracks, at the rate limits (mpc.py:301-312),
//   a_tgt,k = clamp(base_a,k + off_a),  off_a = c_a + m(u_i) span_a,   u_i = (i - G/2)/(G/2)      (df likewise with j)
// where the base sequence is u_prev held over the horizon or -- warm start -- the previous solution shifted by one
// step (utils.py:354-363).  First pass: c = 0, m(u) = u |u| sqrt|u| (dense around the base; span = N * rate limit);
// refinement passes: m(u) = u around the previous winner's offset (refine_targets_kernel).  The candidate with
// offset 0 (i = j = G/2) IS the base sequence wherever that respects the limits.
__device__ __forceinline__ double cand_m(int i, int G, bool first) {
    const double u = (double)(i - G / 2) / (double)(G / 2);
    return first ? u * fabs(u) * sqrt(fabs(u)) : u;
}


__device__ __forceinline__ void ctl_init(Ctl& c, const KP& P, int idx, double a_prev, double df_prev,
                                         const double (&cpar)[4]) {
    c.a = a_prev;
    c.df = df_prev;
    const int i = idx / P.G, j = idx - i * P.G;
    if (P.cand_mode == CAND_RAMP_HOLD || P.cand_mode == CAND_TRACK) {     // da / ddf hold the OFFSETS (see above)
        c.da = cpar[0] + cand_m(i, P.G, P.refine_it == 0) * cpar[2];
        c.ddf = cpar[1] + cand_m(j, P.G, P.refine_it == 0) * cpar[3];
        return;
    }
    // da_i = -ra + (2 ra) i / (G-1)   (oracle candidates_lattice)
    c.da = -P.rate_a + (2 * P.rate_a) * (double)i / (double)(P.G - 1);
    c.ddf = -P.rate_df + (2 * P.rate_df) * (double)j / (double)(P.G - 1);
}

// advances to step k; returns violation bits for the input box / rate constraints
template <typename T>
__device__ __forceinline__ unsigned ctl_step(Ctl& c, const KP& P, int idx, int k, const double* __restrict__ table,
                                             const T* __restrict__ ws, double a_prev, double df_prev, double ey, double ep,
                                             double speed) {
    unsigned v = 0;
    if (P.cand_mode == CAND_TABLE) {
        const double a = table[((size_t)idx * 2 + 0) * P.N + k];
        const double d = table[((size_t)idx * 2 + 1) * P.N + k];
        // input-rate constraints, mpc.py:301-312 (u_{-1} = u_prev)
        if (fmax(fabs(a - c.a) - P.rate_a, fabs(d - c.df) - P.rate_df) > P.tol) v |= VIOL_RATE;
        c.a = a;
        c.df = d;
    } else if (P.cand_mode == CAND_RAMP_HOLD) {
        double ba, bdf;
        ramp_base<T>(ws, P.N, k, a_prev, df_prev, ba, bdf);
        const double ta = clampd(ba + c.da, P.a_min, P.a_max), tdf = clampd(bdf + c.ddf, -P.df_max, P.df_max);
        c.a = clampd(c.a + clampd(ta - c.a, -P.rate_a, P.rate_a), P.a_min, P.a_max);
        c.df = clampd(c.df + clampd(tdf - c.df, -P.rate_df, P.rate_df), -P.df_max, P.df_max);
    } else if (P.cand_mode == CAND_TRACK) {
        double ba, bdf;
        ramp_base<T>(ws, P.N, k, a_prev, df_prev, ba, bdf);
        c.a = track_accel_next(P, k, ba, c.da, speed, c.a);
        c.df = track_steer(P, c.df, ey, ep, c.ddf);
    } else {
        c.a = clampd(c.a + c.da, P.a_min, P.a_max);
        c.df = clampd(c.df + c.ddf, -P.df_max, P.df_max);
    }
    // input box, mpc.py:318-321
    if (fmax(fmax(P.a_min - c.a, c.a - P.a_max), fmax(-P.df_max - c.df, c.df - P.df_max)) > P.tol)
        v |= VIOL_BOX_U;
    return v;
}

// ---------------------------------------------------------------------------------------
// ExactStepper<T>: the oracle's arithmetic in type T
// ---------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void sincos_t(T x, T* s, T* c);
template <> __device__ __forceinline__ void sincos_t<double>(double x, double* s, double* c) { sincos(x, s, c); }
template <> __device__ __forceinline__ void sincos_t<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <typename T> __device__ __forceinline__ T tan_t(T x);
template <> __device__ __forceinline__ double tan_t<double>(double x) { return tan(x); }
template <> __device__ __forceinline__ float tan_t<float>(float x) { return tanf(x); }
template <typename T> __device__ __forceinline__ T atan_t(T x);
template <> __device__ __forceinline__ double atan_t<double>(double x) { return atan(x); }
template <> __device__ __forceinline__ float atan_t<float>(float x) { return atanf(x); }

template <typename T>
struct ExactStepper {
    struct State { T x, y, s, ey, ep, v, psi; };
    struct Beta { T beta, sinb; };
    T h, hh, h6, l_r, lr_ratio, b0, b1, kv;
    int n_rk4;

    __device__ __forceinline__ void init(const KP& P, double b0_, double b1_, double kv_) {
        h = (T)P.h; hh = h / 2; h6 = h / 6;
        l_r = (T)P.l_r; lr_ratio = (T)P.lr_ratio;
        b0 = (T)b0_; b1 = (T)b1_; kv = (T)kv_;
        n_rk4 = P.n_rk4;
    }
    __device__ __forceinline__ void set(State& st, const double (&x0)[7]) const {
        st.x = (T)x0[0]; st.y = (T)x0[1]; st.s = (T)x0[2]; st.ey = (T)x0[3];
        st.ep = (T)x0[4]; st.v = (T)x0[5]; st.psi = (T)x0[6];
    }
    __device__ __forceinline__ void get(const State& st, double (&o)[7]) const {
        o[0] = st.x; o[1] = st.y; o[2] = st.s; o[3] = st.ey; o[4] = st.ep; o[5] = st.v; o[6] = st.psi;
    }
    __device__ __forceinline__ Beta prep(double df) const {
        Beta b;
        b.beta = atan_t<T>(lr_ratio * tan_t<T>((T)df));          // frenet.py:72
        T c;
        sincos_t<T>(b.beta, &b.sinb, &c);
        return b;
    }
    // derivative in the reference's order [s, ey, epsi, v, x, y, psi] (frenet.py:108)
    __device__ __forceinline__ void deriv(T s, T ey, T ep, T v, T psi, T a, const Beta& B, T (&k)[7]) const {
        const T K = (s >= b0 ? kv : (T)0) - (s >= b1 ? kv : (T)0);  // mpc.py:199 pw_const
        T sn, cs;
        sincos_t<T>(B.beta + ep, &sn, &cs);
        k[0] = v * cs / ((T)1 - K * ey);                            // :73
        k[1] = v * sn;                                              // :76
        const T w = v * B.sinb / l_r;                               // :90
        k[2] = w - k[0] * K;                                        // :79
        k[3] = a;                                                   // :81
        T s2, c2;
        sincos_t<T>(psi + B.beta, &s2, &c2);
        k[4] = v * c2;                                              // :84
        k[5] = v * s2;                                              // :87
        k[6] = w;
    }
    __device__ __forceinline__ void step(State& st, double a_, const Beta& B) const {
        const T a = (T)a_;
        T s = st.s, ey = st.ey, ep = st.ep, v = st.v, x = st.x, y = st.y, psi = st.psi;
        for (int j = 0; j < n_rk4; ++j) {                           // :107
            T k1[7], k2[7], k3[7], k4[7];
            deriv(s, ey, ep, v, psi, a, B, k1);
            deriv(s + hh * k1[0], ey + hh * k1[1], ep + hh * k1[2], v + hh * k1[3], psi + hh * k1[6], a, B, k2);
            deriv(s + hh * k2[0], ey + hh * k2[1], ep + hh * k2[2], v + hh * k2[3], psi + hh * k2[6], a, B, k3);
            // reference quirk kept: k4's x,y rows see psi + h/2*k3[6]      (:111)
            deriv(s + h * k3[0], ey + h * k3[1], ep + h * k3[2], v + h * k3[3], psi + hh * k3[6], a, B, k4);
            s = s + h6 * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0]);     // :113
            ey = ey + h6 * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1]);
            ep = ep + h6 * (k1[2] + 2 * k2[2] + 2 * k3[2] + k4[2]);
            v = v + h6 * (k1[3] + 2 * k2[3] + 2 * k3[3] + k4[3]);
            x = x + h6 * (k1[4] + 2 * k2[4] + 2 * k3[4] + k4[4]);
            y = y + h6 * (k1[5] + 2 * k2[5] + 2 * k3[5] + k4[5]);
            psi = psi + h6 * (k1[6] + 2 * k2[6] + 2 * k3[6] + k4[6]);
        }
        st.s = s; st.ey = ey; st.ep = ep; st.v = v; st.x = x; st.y = y; st.psi = psi;
    }
};

// ---------------------------------------------------------------------------------------
// Bookkeeper: cost (mpc.py:356-373) and constraint verdicts on the step-boundary states
// ---------------------------------------------------------------------------------------
struct Book {
    double J;
    double s0;
    unsigned viol;
};

__device__ __forceinline__ void book_state(Book& bk, const KP& P, int k, const double (&st)[7],
                                           const double* __restrict__ obs /* scenario's [n_obs,2,N+1] */,
                                           bool obs_is_float) {
    // tracking terms, mpc.py:363-364 (epsi first, then ey)
    bk.J = bk.J + st[4] * st[4];
    bk.J = bk.J + st[3] * st[3];
    // |ey_k| <= ey_lim, k = 0..N (mpc.py:296-299)
    if (fabs(st[3]) - P.ey_lim > P.tol) bk.viol |= VIOL_EY;
    // 0 <= v_k <= v_max, k = 0..N-1 (mpc.py:315-317)
    if (k < P.N && fmax(P.v_min - st[5], st[5] - P.v_max) > P.tol) bk.viol |= VIOL_BOX_V;
    // non-finite guard
    if (!(fabs(st[0]) < 1e300 && fabs(st[1]) < 1e300 && fabs(st[2]) < 1e300 && fabs(st[3]) < 1e300 &&
          fabs(st[4]) < 1e300 && fabs(st[6]) < 1e300))
        bk.viol |= VIOL_NONFINITE;
    (void)obs; (void)obs_is_float;
}

template <typename T>
__device__ __forceinline__ unsigned collision_viol(const KP& P, int k, double x, double y,
                                                   const T* __restrict__ obs) {
    // d_min^2 - |p_k - p_obs,k|^2 <= 0 for k = 1..N (mpc.py:223-226)
    unsigned v = 0;
    for (int o = 0; o < P.n_obs; ++o) {
        const double ox = (double)obs[(o * 2 + 0) * (P.N + 1) + k];
        const double oy = (double)obs[(o * 2 + 1) * (P.N + 1) + k];
        const double dx = x - ox, dy = y - oy;
        if (P.dmin2 - (dx * dx + dy * dy) > P.tol) v |= VIOL_COLLISION;
    }
    return v;
}

__device__ __forceinline__ unsigned terminal_viol(const KP& P, double v, double a,
                                                  const double* __restrict__ cinf /* [F,3] = A0,A1,b */) {
    // C_inf.A [v_{N-1}; a_{N-1}] <= C_inf.b   (mpc.py:177-180)
    double worst = -1e300;
    for (int m = 0; m < P.F; ++m) {
        const double t = cinf[m * 3 + 0] * v + cinf[m * 3 + 1] * a - cinf[m * 3 + 2];
        worst = fmax(worst, t);
    }
    return (P.F > 0 && worst > P.tol) ? (unsigned)VIOL_TERMINAL : 0u;
}

// the same test with the facets fetched eight at a time -- 24 consecutive doubles, three 64-byte scalar loads in flight, where
// the loop above waits for one facet's three numbers per trip (74 facets: 74 scalar-load latencies, 3 us of accel_rows_kernel's
// 8 us).  The same expressions in the same order: the same bits.
__device__ __forceinline__ unsigned terminal_viol_x8(const KP& P, double v, double a, const double* __restrict__ cinf) {
    double worst = -1e300;
    int m = 0;
    for (; m + 8 <= P.F; m += 8) {
        double f[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) f[i] = cinf[m * 3 + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double t = f[i * 3 + 0] * v + f[i * 3 + 1] * a - f[i * 3 + 2];
            worst = fmax(worst, t);
        }
    }
    for (; m < P.F; ++m) {
        const double t = cinf[m * 3 + 0] * v + cinf[m * 3 + 1] * a - cinf[m * 3 + 2];
        worst = fmax(worst, t);
    }
    return (P.F > 0 && worst > P.tol) ? (unsigned)VIOL_TERMINAL : 0u;
}

// the same test for two candidates at once, four facets per trip: the facets arrive by scalar loads whose latency
// (not the 6 flops per facet) is what one wave pays, so they are issued in batches and shared by both candidates
__device__ __forceinline__ void terminal_viol2(const KP& P, const double (&v)[2], const double (&a)[2],
                                               const double* __restrict__ cinf, unsigned (&viol)[2]) {
    double w0 = -1e300, w1 = -1e300;
    int m = 0;
    for (; m + 4 <= P.F; m += 4) {
        double A0[4], A1[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { A0[i] = cinf[(m + i) * 3 + 0]; A1[i] = cinf[(m + i) * 3 + 1]; bb[i] = cinf[(m + i) * 3 + 2]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w0 = fmax(w0, A0[i] * v[0] + A1[i] * a[0] - bb[i]);
            w1 = fmax(w1, A0[i] * v[1] + A1[i] * a[1] - bb[i]);
        }
    }
    for (; m < P.F; ++m) {
        w0 = fmax(w0, cinf[m * 3 + 0] * v[0] + cinf[m * 3 + 1] * a[0] - cinf[m * 3 + 2]);
        w1 = fmax(w1, cinf[m * 3 + 0] * v[1] + cinf[m * 3 + 1] * a[1] - cinf[m * 3 + 2]);
    }
    if (P.F > 0 && w0 > P.tol) viol[0] |= VIOL_TERMINAL;
    if (P.F > 0 && w1 > P.tol) viol[1] |= VIOL_TERMINAL;
}

// ---------------------------------------------------------------------------------------
// one rollout pass over NC candidates of one scenario
// ---------------------------------------------------------------------------------------
constexpr int STEER_TABLE_MAX_ENTRIES = 360;     // steering columns x N of one float64 search unit's LDS table (igt_fast64.h):
                                                 // 4 x 20 in the plain layout, up to 17 x 20 with 4 live rows (igt_kernels_common.h)

template <typename T>
struct Scenario {           // wave-uniform inputs of one scenario
    double x0[7];
    double a_prev, df_prev;
    double b0, b1, kv;
    double cpar[4];         // ramp-hold candidates: centre offset (a, df) and span (a, df)
    const T* obs;           // [n_obs, 2, N+1]
    const T* ws;            // [2, N] warm start of this scenario (base sequence of the ramp-hold targets), or null
};

// Whether no candidate of the scenario that holds the speed box can come within d_min of any forecast position (wave-uniform;
// the lanes share the horizon).  A control step moves the vehicle by at most dt max(|v_k|, |v_k+1|) (the RK4 stage speeds lie
// between the two), speed-feasible candidates have |v_k| <= max(|v_min|, |v_max|) + tol for k < N, and |v_N| exceeds that by at
// most max|a| dt: the reach over the horizon is N dt (v_abs + a_abs dt), taken with a metre to spare.  A candidate outside
// the speed box is infeasible whatever its distance to the obstacle, so the search's answer is the same either way.
template <typename T>
__device__ __forceinline__ bool obstacles_out_of_reach(const KP& P, const Scenario<T>& S, int lane) {
    const double reach = (fmax(fabs(P.v_min), fabs(P.v_max)) + fmax(fabs(P.a_min), fabs(P.a_max)) * P.dt) * (P.N * P.dt) + 1.0;
    const double lim = sqrt(P.dmin2) + reach, lim2 = lim * lim;
    bool near = false;
    for (int e = lane; e < P.n_obs * P.N; e += 64) {
        const int o = e / P.N, k = e - o * P.N + 1;
        const double dx = S.x0[0] - (double)S.obs[(o * 2 + 0) * (P.N + 1) + k], dy = S.x0[1] - (double)S.obs[(o * 2 + 1) * (P.N + 1) + k];
        near |= !(dx * dx + dy * dy > lim2);                      // NaN counts as near
    }
    return !__any(near);
}



// ---- the incumbent bound of the tracking family's search (igt_fast64.h rollout_one has the derivation) ----
constexpr unsigned VIOL_PRUNED = 128u;            // internal to the search pass: never reported
__device__ __forceinline__ unsigned long long cost_key(double J) {           // order-preserving map double -> u64
    const unsigned long long b = (unsigned long long)__double_as_longlong(J);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double cost_of_key(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
// lam of the scenario, or 0: no usable bound (|kv| ey_b too close to 1)
template <typename T>
__device__ __forceinline__ double progress_slack(const KP& P, const Scenario<T>& S) {
    const double vabs = fmax(fabs(P.v_min), fabs(P.v_max)) + P.tol + fmax(fabs(P.a_min), fabs(P.a_max)) * P.dt;
    const double eyb = P.ey_lim + P.tol + 1.5 * P.dt * vabs;
    const double q = fabs(S.kv) * eyb;
    if (!(q < 0.5) || !(P.w_u >= 0.0)) return 0.0;                           // (a negative effort weight: stage terms of either sign)
    const double lam = 1.0 / (1.0 - q);
    const double reach = (P.N + 1) * P.dt * vabs * lam;                      // every stage argument stays within s_0 +- reach
    const bool clear = S.kv == 0.0 || S.x0[2] + reach < S.b0 || S.x0[2] - reach >= S.b1;
    return clear ? 1.0 : lam;
}


// what the candidate generators are centred on: refinement parameters [B,4] (null: first pass) and the batch's warm
// starts [B,2,N] (null: none)
template <typename T>
struct Centre {
    const double* cpar;
    const T* ws;
};

// Horizon checkpoints (small batches).  The search pass leaves every still-feasible candidate's state at `parts - 1`
// evenly spaced control steps in HBM, exactly (doubles and floats as they are); emit then rolls the winner's `parts`
// pieces of the horizon on as many lanes at once, each resuming from a checkpoint -- bit for bit the unsegmented
// roll-out.  Used up to B = 2048 (the records cost the search pass bandwidth; emit latency dominates only there).
// One unit's record: SEG_FIELDS x 128 slots x 8 B, slot = 64 q + lane of the search wave.
//   fields 0..9: x y s ey epsi v psi J a df (double); 10: (sin,cos)(psi+beta) floats; 11: (cos,sin)(beta_{k-1}) floats
constexpr int SEG_FIELDS = 12, SEG_SLOTS = 128, SEG_MAX_PARTS = 4;
constexpr size_t SEG_UNIT_DOUBLES = (size_t)SEG_FIELDS * SEG_SLOTS;
struct Seg {
    int k0, k1;              // this call rolls control steps [k0, k1)
    const double* load;      // record with the state at k0 (null: start from x0)
    int slot0;               // slot of candidate q of this lane = slot0 + 64 q
};
struct Ckpt {
    double* base;            // [parts-1][n_units] records (null: no checkpoints)
    size_t n_units;
    int unit, slot0;
    int every;               // control steps between checkpoints = N / parts
};

struct NullSink {
    static constexpr bool kKeepsStates = false;
    __device__ __forceinline__ void ctrl(int, int, double, double) {}
    __device__ __forceinline__ void slip(int, int, double, double) {}      // (sin, cos)(beta_k) of step k (igt_fast64.h SegSink)
    __device__ __forceinline__ void state(int, int, const double (&)[7]) {}
};

template <class Stepper, int NC, bool SHARED_DF, typename T, class Sink>
__device__ __forceinline__ void rollout_pass(const KP& P, const Scenario<T>& S, const int (&cidx)[NC],
                                             const double* __restrict__ table,
                                             const double* __restrict__ cinf, Sink& sink,
                                             double (&Jout)[NC], unsigned (&vout)[NC],
                                             double (&sN)[NC], double (&vN)[NC]) {
    Stepper stp;
    stp.init(P, S.b0, S.b1, S.kv);
    typename Stepper::State st[NC];
    Ctl ctl[NC];
    Book bk[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        stp.set(st[q], S.x0);
        ctl_init(ctl[q], P, cidx[q], S.a_prev, S.df_prev, S.cpar);
        bk[q].J = 0.0;
        bk[q].s0 = S.x0[2];
        bk[q].viol = 0;
        sink.state(q, 0, S.x0);
    }
    for (int k = 0; k < P.N; ++k) {
        typename Stepper::Beta B[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            double cur[7];
            stp.get(st[q], cur);
            bk[q].viol |= ctl_step<T>(ctl[q], P, cidx[q], k, table, S.ws, S.a_prev, S.df_prev, cur[3], cur[4], cur[5]);
            sink.ctrl(q, k, ctl[q].a, ctl[q].df);
            // control effort first, then the tracking terms of state k (mpc.py:361-364)
            bk[q].J = bk[q].J + P.w_u * (ctl[q].a * ctl[q].a + ctl[q].df * ctl[q].df);
            book_state(bk[q], P, k, cur, nullptr, false);
            if (k >= 1) bk[q].viol |= collision_viol<T>(P, k, cur[0], cur[1], S.obs);
            if (k == P.N - 1) bk[q].viol |= terminal_viol(P, cur[5], ctl[q].a, cinf);
            if (!SHARED_DF || q == 0) B[q] = stp.prep(ctl[q].df);
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            stp.step(st[q], ctl[q].a, B[SHARED_DF ? 0 : q]);
            double nxt[7];
            stp.get(st[q], nxt);
            sink.state(q, k + 1, nxt);
        }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        double cur[7];
        stp.get(st[q], cur);
        book_state(bk[q], P, P.N, cur, nullptr, false);
        bk[q].viol |= collision_viol<T>(P, P.N, cur[0], cur[1], S.obs);
        sN[q] = cur[2];
        vN[q] = cur[5];
        // terminal progress term, mpc.py:372 (the value-net variant subtracts V instead, :369)
        Jout[q] = bk[q].J;
        vout[q] = bk[q].viol;
    }
}

}  // namespace igt
