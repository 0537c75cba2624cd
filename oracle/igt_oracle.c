/*
 * igt_oracle.c -- TEST INFRASTRUCTURE ONLY.  Plain-C float64 restatement of the
 * reference hot path, used (a) as a second opinion next to oracle/np_oracle.py and
 * (b) as bench.py's "cpu_baseline" (kind "port"), timed on the GPU box's host cores.
 * Nothing in the product path links or loads this file.
 *
 * Each function cites the reference file:line (relative to /root/reference) it follows;
 * operation order is the reference's (no FMA contraction: build with -ffp-contract=off).
 * Pinning: validated against np_oracle.py (itself bit-exact against reference-generated
 * golden vectors) to 1e-12 -- libm's sin/cos/tan/atan differ from numpy's by <= 1 ulp.
 * Cost / constraints / arg-min are restated from source only: PARITY UNPINNED (the
 * reference has no tests and its IPOPT path cannot run here).
 *
 * State order: [x, y, s, ey, epsi, v, psi]  (mpc.py:163)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t N, n_rk4, C, n_obs, G, F;
    double dt, l_r, l_f;
    double v_min, v_max, a_min, a_max, df_max;
    double jerk, steer_rate, ey_lim, d_min, w_u, feas_tol;
} orc_params;

/* mpc.py:199: ca.pw_const(s,[b0,b1],[0,Kv,0]) */
static inline double curvature(double s, const double* kp) {
    return (s >= kp[0] ? kp[2] : 0.0) - (s >= kp[1] ? kp[2] : 0.0);
}

/* kinematic_bicycle_model_frenet.py:71-90, derivative order [s,ey,epsi,v,x,y,psi] (:108) */
static inline void deriv(double s, double ey, double ep, double v, double psi, double a, double beta,
                         double sinb, double l_r, const double* kp, double* k) {
    const double K = curvature(s, kp);
    k[0] = v * cos(beta + ep) / (1 - K * ey); /* :73 */
    k[1] = v * sin(beta + ep);                /* :76 */
    k[2] = v * sinb / l_r - k[0] * K;         /* :79 */
    k[3] = a;                                 /* :81 */
    k[4] = v * cos(psi + beta);               /* :84 */
    k[5] = v * sin(psi + beta);               /* :87 */
    k[6] = v * sinb / l_r;                    /* :90 */
}

/* one control step, kinematic_bicycle_model_frenet.py:93-127; x in planner order */
void orc_frenet_rk4_step(const orc_params* P, const double* x, double a, double df, const double* kp,
                         double* out) {
    const double h = P->dt / P->n_rk4; /* :93 */
    const double beta = atan((P->l_r / (P->l_f + P->l_r)) * tan(df)); /* :72 */
    const double sinb = sin(beta);
    double X = x[0], Y = x[1], s = x[2], ey = x[3], ep = x[4], v = x[5], psi = x[6];
    for (int j = 0; j < P->n_rk4; ++j) { /* :107 */
        double k1[7], k2[7], k3[7], k4[7];
        deriv(s, ey, ep, v, psi, a, beta, sinb, P->l_r, kp, k1);
        deriv(s + h / 2 * k1[0], ey + h / 2 * k1[1], ep + h / 2 * k1[2], v + h / 2 * k1[3], psi + h / 2 * k1[6], a,
              beta, sinb, P->l_r, kp, k2);
        deriv(s + h / 2 * k2[0], ey + h / 2 * k2[1], ep + h / 2 * k2[2], v + h / 2 * k2[3], psi + h / 2 * k2[6], a,
              beta, sinb, P->l_r, kp, k3);
        /* quirk kept (:111): the x,y rows of k4 see psi + h/2*k3[6] */
        deriv(s + h * k3[0], ey + h * k3[1], ep + h * k3[2], v + h * k3[3], psi + h / 2 * k3[6], a, beta, sinb,
              P->l_r, kp, k4);
        s = s + h / 6 * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0]); /* :113 */
        ey = ey + h / 6 * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1]);
        ep = ep + h / 6 * (k1[2] + 2 * k2[2] + 2 * k3[2] + k4[2]);
        v = v + h / 6 * (k1[3] + 2 * k2[3] + 2 * k3[3] + k4[3]);
        X = X + h / 6 * (k1[4] + 2 * k2[4] + 2 * k3[4] + k4[4]);
        Y = Y + h / 6 * (k1[5] + 2 * k2[5] + 2 * k3[5] + k4[5]);
        psi = psi + h / 6 * (k1[6] + 2 * k2[6] + 2 * k3[6] + k4[6]);
    }
    out[0] = X; out[1] = Y; out[2] = s; out[3] = ey; out[4] = ep; out[5] = v; out[6] = psi;
}

/* kinematic_bicycle_model.py:27-31; z = (x, y, psi, v) */
void orc_cartesian_euler_step(const orc_params* P, const double* z, double a, double df, double* out) {
    const double beta = atan(((P->l_r / (P->l_f + P->l_r)) * tan(df)));
    out[0] = z[0] + P->dt * z[3] * cos(z[2] + beta);
    out[1] = z[1] + P->dt * z[3] * sin(z[2] + beta);
    out[2] = z[2] + P->dt * (z[3] * cos(beta) / (P->l_r + P->l_f) * tan(df));
    out[3] = z[3] + P->dt * a;
}

static inline double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

/* One candidate: controls (lattice, SURVEY 8d, or table), rollout (mpc.py:201-209), cost
 * (mpc.py:356-373), verdict bits (mpc.py:177-180, 223-226, 296-321).
 * X[7*(N+1)] / U[2*N] may be NULL.  Returns the violation mask; *J gets the cost. */
static uint32_t one_candidate(const orc_params* P, int c, const double* x0, const double* u_prev,
                              const double* kp, const double* obs, const double* cinfA, const double* cinfb,
                              const double* table, double* X, double* U, double* J) {
    const int N = P->N;
    const double ra = P->dt * P->jerk, rd = P->dt * P->steer_rate, tol = P->feas_tol;
    double a = u_prev[0], d = u_prev[1], da = 0, dd = 0;
    if (!table) {
        const int i = c / P->G, j = c % P->G;
        da = -ra + (2 * ra) * i / (P->G - 1);
        dd = -rd + (2 * rd) * j / (P->G - 1);
    }
    double st[7], nx[7], cost = 0.0;
    uint32_t viol = 0;
    memcpy(st, x0, sizeof st);
    for (int k = 0; k <= N; ++k) {
        if (X) for (int i = 0; i < 7; ++i) X[i * (N + 1) + k] = st[i];
        if (k < N) {
            if (table) {
                const double an = table[((size_t)c * 2 + 0) * N + k], dn = table[((size_t)c * 2 + 1) * N + k];
                if (fmax(fabs(an - a) - ra, fabs(dn - d) - rd) > tol) viol |= 4; /* mpc.py:301-312 */
                a = an; d = dn;
            } else {
                a = clampd(a + da, P->a_min, P->a_max);
                d = clampd(d + dd, -P->df_max, P->df_max);
            }
            if (U) { U[k] = a; U[N + k] = d; }
            if (fmax(fmax(P->a_min - a, a - P->a_max), fmax(-P->df_max - d, d - P->df_max)) > tol) viol |= 2;
            cost = cost + P->w_u * (a * a + d * d);                         /* mpc.py:362 */
            if (fmax(P->v_min - st[5], st[5] - P->v_max) > tol) viol |= 1;   /* mpc.py:316-317 */
        }
        cost = cost + st[4] * st[4]; /* mpc.py:363 */
        cost = cost + st[3] * st[3]; /* mpc.py:364 */
        if (fabs(st[3]) - P->ey_lim > tol) viol |= 8; /* mpc.py:297-299 */
        if (k >= 1)
            for (int o = 0; o < P->n_obs; ++o) { /* mpc.py:223-226 */
                const double dx = st[0] - obs[(o * 2 + 0) * (N + 1) + k], dy = st[1] - obs[(o * 2 + 1) * (N + 1) + k];
                if (P->d_min * P->d_min - (dx * dx + dy * dy) > tol) viol |= 32;
            }
        if (k == N - 1 && P->F > 0) { /* mpc.py:177-180 */
            double worst = -INFINITY;
            for (int m = 0; m < P->F; ++m)
                worst = fmax(worst, cinfA[m * 2] * st[5] + cinfA[m * 2 + 1] * a - cinfb[m]);
            if (worst > tol) viol |= 16;
        }
        for (int i = 0; i < 7; ++i) if (!isfinite(st[i])) viol |= 64;
        if (k < N) {
            orc_frenet_rk4_step(P, st, a, d, kp, nx);
            memcpy(st, nx, sizeof st);
        }
    }
    cost = cost - (st[2] - x0[2]); /* mpc.py:372 */
    *J = cost;
    return viol;
}

/* The shooting solve over a batch (contract of mpc.py:383-406 per scenario).  All arrays
 * float64, layouts as include/igtmpc.h.  Returns 0. */
int orc_solve_batch(const orc_params* P, int B, const double* x0, const double* u_prev, const double* kparams,
                    const uint32_t* flags, const double* obs_xy, const double* cinfA, const double* cinfb,
                    const double* table, double* x_out, double* u_out, double* cost_out, int32_t* argmin_out,
                    int32_t* status_out, int nthreads) {
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; ++b) {
        double xs[7];
        memcpy(xs, x0 + (size_t)b * 7, sizeof xs);
        if (flags[b] & 1u) xs[6] = fabs(xs[6]); /* mpc.py:231-234, 282-285 */
        const double* obs = obs_xy + (size_t)b * P->n_obs * 2 * (N + 1);
        double best = INFINITY;
        int arg = -1;
        for (int c = 0; c < P->C; ++c) {
            double J;
            const uint32_t v = one_candidate(P, c, xs, u_prev + (size_t)b * 2, kparams + (size_t)b * 3, obs, cinfA,
                                             cinfb, table, NULL, NULL, &J);
            if (v == 0 && isfinite(J) && (arg < 0 || J < best)) { best = J; arg = c; }
        }
        double* xo = x_out + (size_t)b * 7 * (N + 1);
        double* uo = u_out + (size_t)b * 2 * N;
        if (arg >= 0) {
            double J;
            one_candidate(P, arg, xs, u_prev + (size_t)b * 2, kparams + (size_t)b * 3, obs, cinfA, cinfb, table, xo,
                          uo, &J);
            cost_out[b] = J;
        } else {
            for (int i = 0; i < 7 * (N + 1); ++i) xo[i] = NAN;
            for (int i = 0; i < 2 * N; ++i) uo[i] = NAN;
            cost_out[b] = INFINITY;
        }
        argmin_out[b] = arg;
        status_out[b] = arg >= 0 ? 0 : 1;
    }
    return 0;
}

/* every candidate (debug): X_all[B,C,7,N+1] (may be NULL), cost_all[B,C], viol_all[B,C] */
int orc_rollout_all(const orc_params* P, int B, const double* x0, const double* u_prev, const double* kparams,
                    const uint32_t* flags, const double* obs_xy, const double* cinfA, const double* cinfb,
                    const double* table, double* X_all, double* U_all, double* cost_all, uint32_t* viol_all,
                    int nthreads) {
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; ++b) {
        double xs[7];
        memcpy(xs, x0 + (size_t)b * 7, sizeof xs);
        if (flags[b] & 1u) xs[6] = fabs(xs[6]);
        const double* obs = obs_xy + (size_t)b * P->n_obs * 2 * (N + 1);
        for (int c = 0; c < P->C; ++c) {
            const size_t bc = (size_t)b * P->C + c;
            double J;
            viol_all[bc] = one_candidate(P, c, xs, u_prev + (size_t)b * 2, kparams + (size_t)b * 3, obs, cinfA, cinfb,
                                         table, X_all ? X_all + bc * 7 * (N + 1) : NULL,
                                         U_all ? U_all + bc * 2 * N : NULL, &J);
            cost_all[bc] = J;
        }
    }
    return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
