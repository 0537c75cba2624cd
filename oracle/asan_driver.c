/*
 * asan_driver.c -- TEST INFRASTRUCTURE ONLY.  Runs oracle/igt_oracle.c under AddressSanitizer +
 * UndefinedBehaviorSanitizer on the host (GPU sanitizers are not available on the pool):
 *
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off \
 *       oracle/igt_oracle.c oracle/asan_driver.c -lm -o asan_driver      (make -C oracle asan)
 *   ./asan_driver in.bin out.bin
 *
 * in.bin  : int32 header {B, N, n_rk4, C, n_obs, F, table(0/1), all(0/1)} then float64 arrays in the order
 *           x0[B,7] u_prev[B,2] kparams[B,3] obs_xy[B,n_obs,2,N+1] cinfA[F,2] cinfb[F] table[C,2,N] (if table)
 *           and uint32 flags[B]; exactly-sized heap blocks, so any out-of-bounds read of the kernels is reported.
 * out.bin : solve (x, u, cost, argmin, status) or, with all = 1, every candidate's (X, U, cost, viol).
 * tests/test_host_logic.py compares out.bin bit for bit with what the ordinary liboracle.so build returns.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct {
    int32_t N, n_rk4, C, n_obs, G, F;
    double dt, l_r, l_f;
    double v_min, v_max, a_min, a_max, df_max;
    double jerk, steer_rate, ey_lim, d_min, w_u, feas_tol;
} orc_params;

int orc_solve_batch(const orc_params*, int, const double*, const double*, const double*, const uint32_t*, const double*,
                    const double*, const double*, const double*, double*, double*, double*, int32_t*, int32_t*, int);
int orc_rollout_all(const orc_params*, int, const double*, const double*, const double*, const uint32_t*, const double*,
                    const double*, const double*, const double*, double*, double*, double*, uint32_t*, int);

static void* rd(FILE* f, size_t n, size_t sz) {
    const size_t bytes = n * sz;
    void* p = malloc(bytes > 0 ? bytes : 1);
    if (!p || fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[8];
    if (fread(h, 4, 8, f) != 8) return 2;
    const int B = h[0], N = h[1], C = h[3], n_obs = h[4], F = h[5], table = h[6], all = h[7];
    int G = 1;
    while (!table && G * G < C) ++G;
    orc_params P = {N, h[2], C, n_obs, G, F, 0.1, 4.47 / 2, 4.47 / 2, 0, 5, -4, 3, 1, 0.9, 0.7, 0.2, 5.6, 0.05, 1e-6};
    double* x0 = rd(f, (size_t)B * 7, 8);
    double* up = rd(f, (size_t)B * 2, 8);
    double* kp = rd(f, (size_t)B * 3, 8);
    double* obs = rd(f, (size_t)B * n_obs * 2 * (N + 1), 8);
    double* A = rd(f, (size_t)F * 2, 8);
    double* b = rd(f, (size_t)F, 8);
    double* tab = table ? rd(f, (size_t)C * 2 * N, 8) : NULL;
    uint32_t* fl = rd(f, (size_t)B, 4);
    fclose(f);
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    if (all) {
        const size_t bc = (size_t)B * C;
        double* X = malloc(bc * 7 * (N + 1) * 8 + 1);
        double* U = malloc(bc * 2 * N * 8 + 1);
        double* J = malloc(bc * 8 + 1);
        uint32_t* v = malloc(bc * 4 + 1);
        orc_rollout_all(&P, B, x0, up, kp, fl, obs, F ? A : NULL, F ? b : NULL, tab, X, U, J, v, 2);
        fwrite(X, 8, bc * 7 * (N + 1), o); fwrite(U, 8, bc * 2 * N, o); fwrite(J, 8, bc, o); fwrite(v, 4, bc, o);
        free(X); free(U); free(J); free(v);
    } else {
        double* x = malloc((size_t)B * 7 * (N + 1) * 8 + 1);
        double* u = malloc((size_t)B * 2 * N * 8 + 1);
        double* J = malloc((size_t)B * 8 + 1);
        int32_t* am = malloc((size_t)B * 4 + 1);
        int32_t* st = malloc((size_t)B * 4 + 1);
        orc_solve_batch(&P, B, x0, up, kp, fl, obs, F ? A : NULL, F ? b : NULL, tab, x, u, J, am, st, 2);
        fwrite(x, 8, (size_t)B * 7 * (N + 1), o); fwrite(u, 8, (size_t)B * 2 * N, o); fwrite(J, 8, B, o);
        fwrite(am, 4, B, o); fwrite(st, 4, B, o);
        free(x); free(u); free(J); free(am); free(st);
    }
    fclose(o);
    free(x0); free(up); free(kp); free(obs); free(A); free(b); free(tab); free(fl);
    return 0;
}
