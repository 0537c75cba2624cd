/* c_caller.c -- the C ABI of libigtmpc.so bound from plain C, no Python, no torch.
 *
 * What a C/C++ maintainer of a driver like the reference's evaluate.py would write instead of
 *   planner.update_initial_condition(...); planner.update_predictions(...); planner.solve(...)   (evaluate.py:470-482)
 * for a whole batch of (scenario, ego) problems: host buffers in, host buffers out (IGT_MEM_HOST).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_caller.c -o c_caller -Ligt-mpc-int_amd/igtmpc -ligtmpc \
 *       -Wl,-rpath,$PWD/igt-mpc-int_amd/igtmpc -L/opt/rocm/lib -lamdhip64 -lm
 *   ./c_caller [B]        -> prints the number of solved scenarios and the first winner's cost / first control
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "igtmpc.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != IGT_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, igt_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    igt_params p;
    CHECK(igt_params_default(&p));                 /* mpc.py:45-62 constants, N = 20, C = 256, lattice candidates */
    p.n_obs = 1;
    igt_handle* h = NULL;
    CHECK(igt_create(&p, 0, &h));

    const int N = p.N;
    double* x0 = calloc((size_t)B * 7, sizeof(double));
    double* u_prev = calloc((size_t)B * 2, sizeof(double));
    double* kparams = calloc((size_t)B * 3, sizeof(double));
    uint32_t* flags = calloc((size_t)B, sizeof(uint32_t));
    double* obs = calloc((size_t)B * 2 * (N + 1), sizeof(double));
    double* x_out = malloc((size_t)B * 7 * (N + 1) * sizeof(double));
    double* u_out = malloc((size_t)B * 2 * N * sizeof(double));
    double* cost = malloc((size_t)B * sizeof(double));
    int32_t* argmin = malloc((size_t)B * sizeof(int32_t));
    int32_t* status = malloc((size_t)B * sizeof(int32_t));
    if (!x0 || !u_prev || !kparams || !flags || !obs || !x_out || !u_out || !cost || !argmin || !status) return 2;

    /* vehicles on route '12' (left turn: K = 1/8.6 on [19.3, 32.81), mpc.py:183-200) at different arc lengths and speeds;
     * the other vehicle far away (what filter_preds writes for an opponent behind the ego, utils.py:381-386) */
    for (int b = 0; b < B; ++b) {
        const double s = 5.0 + 30.0 * b / B, v = 1.0 + 3.0 * ((b * 7) % B) / B;
        double* x = x0 + (size_t)b * 7;
        x[0] = s; x[1] = 2.8; x[2] = s; x[3] = 0.01; x[4] = -0.005; x[5] = v; x[6] = 0.0;   /* x y s ey epsi v psi */
        u_prev[b * 2 + 0] = 0.1; u_prev[b * 2 + 1] = 0.0;
        kparams[b * 3 + 0] = 19.3; kparams[b * 3 + 1] = 19.3 + 8.6 * acos(-1.0) / 2; kparams[b * 3 + 2] = 1.0 / 8.6;
        for (int k = 0; k <= N; ++k) { obs[(size_t)b * 2 * (N + 1) + k] = -20.0; obs[(size_t)b * 2 * (N + 1) + (N + 1) + k] = -20.0; }
    }
    CHECK(igt_solve_batch_f64(h, B, x0, u_prev, kparams, flags, obs, NULL, NULL, x_out, u_out, cost, argmin, status,
                              IGT_MEM_HOST, NULL));
    int solved = 0, first = -1;
    for (int b = 0; b < B; ++b) {
        if (status[b] == 0) { ++solved; if (first < 0) first = b; }
        else if (argmin[b] != -1 || !isinf(cost[b]) || !isnan(x_out[(size_t)b * 7 * (N + 1)])) { fprintf(stderr, "bad failure contract at %d\n", b); return 3; }
    }
    printf("solved %d of %d", solved, B);
    if (first >= 0)
        printf("; scenario %d: candidate %d, cost %.6f, applies a = %.4f df = %.4f, s_N = %.4f", first, argmin[first], cost[first],
               u_out[(size_t)first * 2 * N], u_out[(size_t)first * 2 * N + N], x_out[(size_t)first * 7 * (N + 1) + 2 * (N + 1) + N]);
    printf("\n");
    CHECK(igt_destroy(h));
    free(x0); free(u_prev); free(kparams); free(flags); free(obs); free(x_out); free(u_out); free(cost); free(argmin); free(status);
    return solved > 0 ? 0 : 4;
}
