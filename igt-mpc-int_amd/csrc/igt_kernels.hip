// igt_kernels.hip -- gfx950 kernels of the batched shooting solver and their launchers.
//
//   search_kernel   one wavefront per scenario: rolls all C candidates (NC per lane per pass),
//                   cost + verdicts, lane-local best, 6-step wave butterfly arg-min,
//                   writes (cost, argmin, status) = 12 B per solve.
//   emit_kernel     one lane per scenario: re-rolls the winner with the same arithmetic and
//                   writes x*[7,N+1], u*[2,N] (1/C of the search work).
//   rollout_all_kernel   debug/parity: every candidate's trajectory, cost and verdict bits.
//   cartesian_euler_kernel   kinematic_bicycle_model.py:15-50, one lane per trajectory.
#define IGT_KERNELS_TU 1
#include "igt_device.h"
#include "igt_fast.h"
#include "igt_fast64.h"
#include "igt_launch.h"

namespace igt {

template <typename T>
__device__ __forceinline__ void load_scenario(Scenario<T>& S, const KP& P, int b, const T* __restrict__ x0,
                                              const T* __restrict__ u_prev, const T* __restrict__ kparams,
                                              const uint32_t* __restrict__ flags, const T* __restrict__ obs,
                                              Centre<T> cpar = Centre<T>{nullptr, nullptr}) {
#pragma unroll
    for (int i = 0; i < 7; ++i) S.x0[i] = (double)x0[(size_t)b * 7 + i];
    // ego routes '32','41' use |heading| (mpc.py:231-234, 282-285)
    if (flags[b] & 1u) S.x0[6] = fabs(S.x0[6]);
    S.a_prev = (double)u_prev[(size_t)b * 2 + 0];
    S.df_prev = (double)u_prev[(size_t)b * 2 + 1];
    S.b0 = (double)kparams[(size_t)b * 3 + 0];
    S.b1 = (double)kparams[(size_t)b * 3 + 1];
    S.kv = (double)kparams[(size_t)b * 3 + 2];
    S.obs = obs + (size_t)b * P.n_obs * 2 * (P.N + 1);
    // ramp-hold targets = base sequence + offset.  Base: the warm start u_ws[b] (the previous solution shifted by one
    // step, utils.py:354-363 augment_prev_sol) when the scenario carries one (IGT_FLAG_WARM), else u_prev held.
    S.ws = (cpar.ws && (flags[b] & 2u)) ? cpar.ws + (size_t)b * 2 * P.N : nullptr;
    if (cpar.cpar) {     // refinement pass: centre offset / span chosen by refine_targets_kernel
#pragma unroll
        for (int i = 0; i < 4; ++i) S.cpar[i] = cpar.cpar[(size_t)b * 4 + i];
    } else {             // first pass: offsets centred on 0, span = what the rate limits reach over the horizon
        S.cpar[0] = 0.0; S.cpar[1] = 0.0;
        S.cpar[2] = P.N * P.rate_a; S.cpar[3] = P.cand_mode == CAND_TRACK ? P.trk_span : P.N * P.rate_df;
    }
}

__device__ __forceinline__ bool finite_d(double x) { return fabs(x) < 1.79e308; }

template <class Stepper, typename T, int NC, bool SHARED_DF, bool VALUE>
__global__ __launch_bounds__(256) void search_kernel(KP P, int B, const T* __restrict__ x0,
                                                     const T* __restrict__ u_prev,
                                                     const T* __restrict__ kparams,
                                                     const uint32_t* __restrict__ flags,
                                                     const T* __restrict__ obs,
                                                     const double* __restrict__ table,
                                                     const double* __restrict__ cinf, Centre<T> cpar,
                                                     T* __restrict__ cost_out, int32_t* __restrict__ argmin_out,
                                                     int32_t* __restrict__ status_out, T* __restrict__ rec_sN,
                                                     T* __restrict__ rec_vN, double* __restrict__ rec_J,
                                                     uint32_t* __restrict__ rec_viol) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    Scenario<T> S;
    load_scenario<T>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);

    double bestJ = 0.0;
    int bestC = -1;
    NullSink sink;
    const int passes = P.C / (64 * NC);
    for (int p = 0; p < passes; ++p) {
        int cidx[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) cidx[q] = (p * NC + q) * 64 + lane;
        double J[NC], sN[NC], vN[NC];
        unsigned viol[NC];
        rollout_pass<Stepper, NC, SHARED_DF, T>(P, S, cidx, table, cinf, sink, J, viol, sN, vN);
        if (VALUE) {   // terminal value network: leave the terminal term to value_kernel (mpc.py:369)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const size_t idx = (size_t)b * P.C + cidx[q];
                rec_sN[idx] = (T)sN[q]; rec_vN[idx] = (T)vN[q]; rec_J[idx] = J[q]; rec_viol[idx] = viol[q];
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const double Jq = J[q] - (sN[q] - S.x0[2]);  // mpc.py:372
            const bool ok = (viol[q] == 0) && finite_d(Jq);
            // candidates arrive in increasing index per lane: strict '<' keeps the lowest index
            if (ok && (bestC < 0 || Jq < bestJ)) { bestJ = Jq; bestC = cidx[q]; }
        }
    }
    // wave butterfly arg-min, ties -> lowest candidate index
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if (VALUE) return;
    if (lane == 0) {
        cost_out[b] = bestC >= 0 ? (T)bestJ : (T)INFINITY;
        argmin_out[b] = bestC;
        status_out[b] = bestC >= 0 ? 0 : 1;
    }
}

// partials [B, W] -> cost / argmin / status.  (J, c) is compared lexicographically: steering-ordered slices hold their
// columns from the centre outwards, so a later slice can hold the LOWER candidate index of an exact tie
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(int B, int W, const double* __restrict__ part_J,
                                                              const int32_t* __restrict__ part_c,
                                                              T* __restrict__ cost_out, int32_t* __restrict__ argmin_out,
                                                              int32_t* __restrict__ status_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double bestJ = 0.0;
    int c = -1;
    for (int w = 0; w < W; ++w) {
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; }
    }
    cost_out[b] = c >= 0 ? (T)bestJ : (T)INFINITY;
    argmin_out[b] = c;
    status_out[b] = c >= 0 ? 0 : 1;
}

// Ramp-hold refinement: new centre OFFSET (relative to the base sequence: u_prev held, or the warm start) = the winner's
// offset, new span = half the distance between the winner's neighbours on the previous grid.  No feasible candidate:
// parameters are left as they are.
__global__ __launch_bounds__(256) void refine_targets_kernel(KP P, int B, int W, const double* __restrict__ part_J,
                                                             const int32_t* __restrict__ part_c,
                                                             double* __restrict__ cpar, int first) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double ca, cd, sa, sd;
    if (first) {   // parameters of the first pass (load_scenario's defaults)
        ca = 0.0; cd = 0.0;
        sa = P.N * P.rate_a; sd = P.cand_mode == CAND_TRACK ? P.trk_span : P.N * P.rate_df;
    } else {
        ca = cpar[(size_t)b * 4 + 0]; cd = cpar[(size_t)b * 4 + 1]; sa = cpar[(size_t)b * 4 + 2]; sd = cpar[(size_t)b * 4 + 3];
    }
    double bestJ = 0.0;
    int c = -1;
    for (int w = 0; w < W; ++w) {
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; }
    }
    if (c >= 0) {
        const int G = P.G, i = c / G, j = c - i * G;
        const int ilo = i > 0 ? i - 1 : i, ihi = i < G - 1 ? i + 1 : i;
        const int jlo = j > 0 ? j - 1 : j, jhi = j < G - 1 ? j + 1 : j;
        const bool f = first != 0;
        const double na = ca + cand_m(i, G, f) * sa;
        const double nd = cd + cand_m(j, G, f) * sd;
        sa = sa * (cand_m(ihi, G, f) - cand_m(ilo, G, f)) / (double)(ihi - ilo > 1 ? 2 : 1);
        sd = sd * (cand_m(jhi, G, f) - cand_m(jlo, G, f)) / (double)(jhi - jlo > 1 ? 2 : 1);
        ca = na; cd = nd;
    }
    cpar[(size_t)b * 4 + 0] = ca; cpar[(size_t)b * 4 + 1] = cd; cpar[(size_t)b * 4 + 2] = sa; cpar[(size_t)b * 4 + 3] = sd;
}

template <typename T>
struct StoreSink {
    static constexpr bool kKeepsStates = true;
    T* x;   // [7, N+1] of this scenario/candidate (may be null)
    T* u;   // [2, N]
    int N;
    __device__ __forceinline__ void ctrl(int, int k, double a, double df) {
        if (u) { u[k] = (T)a; u[N + k] = (T)df; }
    }
    __device__ __forceinline__ void state(int, int k, const double (&st)[7]) {
        if (x) {
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i * (N + 1) + k] = (T)st[i];
        }
    }
};

template <class Stepper, typename T>
__global__ __launch_bounds__(64) void emit_kernel(KP P, int B, const T* __restrict__ x0,
                                                  const T* __restrict__ u_prev, const T* __restrict__ kparams,
                                                  const uint32_t* __restrict__ flags, const T* __restrict__ obs,
                                                  const double* __restrict__ table,
                                                  const double* __restrict__ cinf, Centre<T> cpar,
                                                  const int32_t* __restrict__ argmin, T* __restrict__ x_out,
                                                  T* __restrict__ u_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T* xo = x_out + (size_t)b * 7 * (P.N + 1);
    T* uo = u_out + (size_t)b * 2 * P.N;
    const int c = argmin[b];
    if (c < 0) {  // is_opt False (mpc.py:402-406): no trajectory
        const T nan = (T)NAN;
        for (int i = 0; i < 7 * (P.N + 1); ++i) xo[i] = nan;
        for (int i = 0; i < 2 * P.N; ++i) uo[i] = nan;
        return;
    }
    Scenario<T> S;
    load_scenario<T>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    StoreSink<T> sink{xo, uo, P.N};
    const int cidx[1] = {c};
    double J[1], sN[1], vN[1];
    unsigned viol[1];
    rollout_pass<Stepper, 1, false, T>(P, S, cidx, table, cinf, sink, J, viol, sN, vN);
}

template <class Stepper, typename T>
__global__ __launch_bounds__(256) void rollout_all_kernel(KP P, int B, const T* __restrict__ x0,
                                                          const T* __restrict__ u_prev,
                                                          const T* __restrict__ kparams,
                                                          const uint32_t* __restrict__ flags,
                                                          const T* __restrict__ obs,
                                                          const double* __restrict__ table,
                                                          const double* __restrict__ cinf, Centre<T> cpar, T* __restrict__ X_all,
                                                          T* __restrict__ U_all, T* __restrict__ cost_all,
                                                          uint32_t* __restrict__ viol_all, T* __restrict__ rec_sN,
                                                          T* __restrict__ rec_vN, double* __restrict__ rec_J,
                                                          uint32_t* __restrict__ rec_viol) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    Scenario<T> S;
    load_scenario<T>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    for (int c = lane; c < P.C; c += 64) {
        const size_t bc = (size_t)b * P.C + c;
        StoreSink<T> sink{X_all ? X_all + bc * 7 * (P.N + 1) : nullptr, U_all ? U_all + bc * 2 * P.N : nullptr, P.N};
        const int cidx[1] = {c};
        double J[1], sN[1], vN[1];
        unsigned viol[1];
        rollout_pass<Stepper, 1, false, T>(P, S, cidx, table, cinf, sink, J, viol, sN, vN);
        if (rec_J) {   // value-net cost: value_kernel adds the terminal term and fills cost_all / viol_all
            rec_sN[bc] = (T)sN[0]; rec_vN[bc] = (T)vN[0]; rec_J[bc] = J[0]; rec_viol[bc] = viol[0];
            continue;
        }
        const double Jq = J[0] - (sN[0] - S.x0[2]);
        if (!finite_d(Jq)) viol[0] |= VIOL_NONFINITE;
        cost_all[bc] = (T)Jq;
        viol_all[bc] = viol[0];
    }
}

// ---------------------------------------------------------------------------------------
// float32 path: two candidates per lane rolled as a packed pair (igt_fast.h)
// ---------------------------------------------------------------------------------------
template <typename T>
struct PairSink {
    static constexpr bool kKeepsStates = true;
    T* x[2];
    T* u[2];
    int N;
    __device__ __forceinline__ void ctrl(int q, int k, double a, double df) {
        if (u[q]) { u[q][k] = (T)a; u[q][N + k] = (T)df; }
    }
    __device__ __forceinline__ void state(int q, int k, const double (&st)[7]) {
        if (x[q]) {
#pragma unroll
            for (int i = 0; i < 7; ++i) x[q][i * (N + 1) + k] = (T)st[i];
        }
    }
};

// Which candidates a unit (slice p of W) rolls on which lane, and the inverse.  Generated G x G families with
// W * 128 == C and W | G: a slice takes G/W STEERING values (all G accelerations), the values handed out from the centre
// of the range outwards.  The |e_y| verdict (87 % of all failures) depends mostly on the steering sequence, so the
// slices holding the extreme steering values fail as a whole within a few steps and leave through the early exit
// (tools/death_steps.py: 23 % fewer wave-steps than slices cut along the acceleration axis).  Pairs still share their
// steering column: (i, j) and (i + G/2, j).  Otherwise: chunks of 64 in index order, two per slice.
template <int CAND>
__device__ __forceinline__ bool steering_slices(const KP& P, int W) {
    return CAND != CAND_TABLE && P.G * P.G == P.C && W * 128 == P.C && P.G % W == 0 && !(P.dev & 1);
}
template <int CAND>
__device__ __forceinline__ void slice_candidates(const KP& P, int W, int p, int lane, int (&cidx)[2]) {
    const int chunks = P.C / 64;
    // an odd number of 64-candidate chunks: the last slice rolls its chunk twice (harmless duplicate)
    cidx[0] = (2 * p) * 64 + lane;
    cidx[1] = (2 * p + 1 < chunks ? 2 * p + 1 : 2 * p) * 64 + lane;
    if (steering_slices<CAND>(P, W)) {
        const int nj = P.G / W, jl = lane % nj, il = lane / nj, r = p * nj + jl;
        const int j = (r & 1) ? P.G / 2 - 1 - (r >> 1) : P.G / 2 + (r >> 1);
        cidx[0] = il * P.G + j;
        cidx[1] = (il + P.G / 2) * P.G + j;
    }
}
template <int CAND>
__device__ __forceinline__ void candidate_slot(const KP& P, int W, int c, int& p, int& slot) {   // slot = 64 q + lane
    if (steering_slices<CAND>(P, W)) {
        const int i = c / P.G, j = c - i * P.G, nj = P.G / W;
        const int r = j >= P.G / 2 ? 2 * (j - P.G / 2) : 2 * (P.G / 2 - 1 - j) + 1;
        const int q = i >= P.G / 2 ? 1 : 0, il = i - q * (P.G / 2);
        p = r / nj;
        slot = 64 * q + il * nj + (r - p * nj);
    } else {
        const int chunk = c / 64;
        p = chunk / 2;
        slot = 64 * (chunk & 1) + (c & 63);
    }
}

// One work unit = one (scenario, 128-candidate slice): W = ceil(C/128) units per scenario.  A unit is rolled by
// one 64-lane wave and leaves its slice's best (J, c); emit_fast_kernel reduces the W partials (ties -> lowest
// candidate index).
template <int CAND, bool HI, bool VALUE, bool CKPT>
__device__ __forceinline__ void search_unit(const KP& P, int W, int b, int p, double* __restrict__ ckpt, int ck_parts,
                                            int n_units,
                                            const float* __restrict__ x0,
                                            const float* __restrict__ u_prev, const float* __restrict__ kparams,
                                            const uint32_t* __restrict__ flags, const float* __restrict__ obs,
                                            const double* __restrict__ table, const double* __restrict__ cinf,
                                            Centre<float> cpar, double* __restrict__ part_J,
                                            int32_t* __restrict__ part_c, float* __restrict__ rec_sN,
                                            float* __restrict__ rec_vN, double* __restrict__ rec_J,
                                            uint32_t* __restrict__ rec_viol, unsigned* __restrict__ rec_count,
                                            int32_t* __restrict__ rec_b) {
    const int gw = b * W + p;
    const int lane = threadIdx.x & 63;
    Scenario<float> S;
    load_scenario<float>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    NullSink sink;
    int cidx[2];
    slice_candidates<CAND>(P, W, p, lane, cidx);
    double J[2], sN[2], vN[2];
    unsigned viol[2];
    const Ckpt ck{CKPT && ck_parts > 1 ? ckpt : nullptr, (size_t)n_units, gw, lane, ck_parts > 1 ? P.N / ck_parts : 0};
    rollout_pair<CAND, HI, true, true, float, NullSink, true, CKPT>(P, S, cidx, table, cinf, sink, J, viol, sN, vN, ck);
    if (VALUE) {   // terminal value network (mpc.py:369): append the feasible candidates for the value kernels
        const bool dup = cidx[1] == cidx[0];                       // odd chunk count: second half is a duplicate
        const bool ok0 = viol[0] == 0 && finite_d(J[0]), ok1 = viol[1] == 0 && finite_d(J[1]) && !dup;
        const unsigned long long m0 = __ballot(ok0), m1 = __ballot(ok1);
        const unsigned n0 = __popcll(m0), n1 = __popcll(m1);
        unsigned base = 0;
        if (lane == 0 && n0 + n1) base = atomicAdd(rec_count, n0 + n1);
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int32_t* rec_c = reinterpret_cast<int32_t*>(rec_viol);
        if (ok0) {
            const unsigned e = base + __popcll(m0 & lt);
            rec_b[e] = b; rec_c[e] = cidx[0]; rec_sN[e] = (float)sN[0]; rec_vN[e] = (float)vN[0]; rec_J[e] = J[0];
        }
        if (ok1) {
            const unsigned e = base + n0 + __popcll(m1 & lt);
            rec_b[e] = b; rec_c[e] = cidx[1]; rec_sN[e] = (float)sN[1]; rec_vN[e] = (float)vN[1]; rec_J[e] = J[1];
        }
        return;
    }
    double bestJ = 0.0;
    int bestC = -1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double Jq = J[q] - (sN[q] - S.x0[2]);  // mpc.py:372
        const bool ok = (viol[q] == 0) && finite_d(Jq);
        if (ok && (bestC < 0 || Jq < bestJ)) { bestJ = Jq; bestC = cidx[q]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if (lane == 0) { part_J[gw] = bestJ; part_c[gw] = bestC; }
}

// Scenario j of queue q.  Blocks of 8 consecutive scenarios are dealt to the 8 queues rotated by the block index, so
// that a batch whose make-up repeats with a period of 8 (the benchmark's does: route pair and ego index are functions
// of b mod 64) does not give one XCD all the turning routes: measured 30 % spread between the queues' finishing times
// with b mod 8, a few % with the rotation.  b >= B marks a hole in the last block.
__device__ __forceinline__ int queue_scenario(int q, int j) { return 8 * j + ((q - j) & 7); }

// Longest units first (small batches).  A unit's wall time is what the tail of the search kernel is made of, and units
// differ 8x (16 .. 130 us at 2 waves per SIMD).  What the traces (tools/trace_units.py) and the oracle-side analysis
// (tools/death_steps.py) show to matter:
//   * the centre-steering slice runs (nearly) the whole horizon; the others leave through the early exit after a
//     number of steps that falls with the speed (|e_y| grows with v: 18 steps at v0 < 1 m/s, 9 at v0 > 4);
//   * a scenario that meets its arc within the horizon rolls the long sub-step variants (about 1.9x per step).
// One workgroup per queue sorts its units into QC cost classes, most expensive first, keeping the scenario order
// inside a class (a stable counting sort, so the order is a function of the inputs alone).
// order[q][k] = (scenario ordinal in the queue) * 256 + slice.
constexpr int QC = 8, QB_THREADS = 1024, QB_TRIPS = 2;     // up to 2048 units per queue (B <= 8192 at W = 2)
template <typename T>
__global__ __launch_bounds__(QB_THREADS) void build_queues_kernel(KP P, int B, int W, const T* __restrict__ x0,
                                                                  const T* __restrict__ kparams,
                                                                  unsigned* __restrict__ order, int stride,
                                                                  unsigned* __restrict__ work_counter) {
    __shared__ int cnt[QB_TRIPS][QB_THREADS / 64][QC];   // [trip][wave][class] counts, then exclusive offsets
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) work_counter[q * 64] = 0u;             // this queue's unit counter (saves the memset node)
    const int n_scen = (B + 7) / 8, n = n_scen * W;
    int cls[QB_TRIPS], rank[QB_TRIPS];
#pragma unroll
    for (int t = 0; t < QB_TRIPS; ++t) {
        const int i = t * QB_THREADS + tid;
        int c = -1;
        if (i < n) {
            const int j = i / W, p = i - j * W, b = queue_scenario(q, j);
            c = QC - 1;                                  // a hole of the last block of 8: sorts last, skipped by the search
            if (b < B) {
                const float s0 = (float)x0[(size_t)b * 7 + 2], v0 = (float)x0[(size_t)b * 7 + 5];
                const float b0 = (float)kparams[(size_t)b * 3 + 0], b1 = (float)kparams[(size_t)b * 3 + 1], kv = (float)kparams[(size_t)b * 3 + 2];
                const float reach = s0 + 1.5f * fmaxf(v0, 1.0f) * (float)(P.N * P.dt);
                const bool arc = kv != 0.0f && reach >= b0 && s0 <= b1;
                const float frac = p == 0 ? 0.95f : fminf(fmaxf(1.05f - 0.15f * v0, 0.4f), 0.95f);   // share of the horizon rolled
                const float cost = frac * (arc ? 1.9f : 1.0f);                                        // 0.4 .. 1.8
                c = (int)((1.85f - cost) * ((float)QC / 1.5f));
                c = c < 0 ? 0 : (c > QC - 1 ? QC - 1 : c);
            }
        }
        cls[t] = c; rank[t] = 0;
#pragma unroll
        for (int cc = 0; cc < QC; ++cc) {
            const unsigned long long m = __ballot(c == cc);
            if (lane == 0) cnt[t][wv][cc] = __popcll(m);
            if (c == cc) rank[t] = __popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    __shared__ int total[QC];
    if (tid < QC) {          // per class: exclusive offsets in index order (trip, wave), and the class total
        int run = 0;
        for (int t = 0; t < QB_TRIPS; ++t)
            for (int w2 = 0; w2 < QB_THREADS / 64; ++w2) { const int v = cnt[t][w2][tid]; cnt[t][w2][tid] = run; run += v; }
        total[tid] = run;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < QB_TRIPS; ++t) {
        if (cls[t] < 0) continue;
        int off = cnt[t][wv][cls[t]] + rank[t];
        for (int cc = 0; cc < cls[t]; ++cc) off += total[cc];      // the more expensive classes come first
        const int i = t * QB_THREADS + tid, j = i / W, p = i - j * W;
        order[(size_t)q * stride + off] = (unsigned)j * 256u + (unsigned)p;
    }
}

// Persistent waves, one per workgroup.  Replacing a retired single-unit workgroup costs tens of microseconds of idle
// wave slot on this part (measured: 2 of 3 slots occupied on average) and unit durations differ 3x (early exit,
// straight vs arc), so the waves loop, taking units from counters until none is left.  A returning device-scope
// atomic on ONE address retires every ~11.4 ns on MI355X (tools/atomic_probe.hip; the XCDs' L2s are not coherent,
// so it executes memory-side): a single counter would cap the kernel at 44 M solves/s and queue the waves of a
// small batch behind each other.  Hence one counter per XCD (workgroup n runs on XCD n mod 8), 256 B apart; queue q
// owns one scenario of every block of 8 (queue_scenario) and deals them out scenario-major, or longest first when
// the batch is small (build_queues_kernel); a wave whose queue is dry takes from the other queues in turn.
// `unit(b, p)` rolls slice p of scenario b (float path: search_unit, 128 candidates; double path: search_unit64, 64).
template <class Unit>
__device__ __forceinline__ void search_waves(const KP& P, int B, int W, int queues, unsigned* __restrict__ work_counter,
                                             const unsigned* __restrict__ order, int order_stride, const Unit& unit) {
    const unsigned q = blockIdx.x % (unsigned)queues, uW = (unsigned)W;
    const unsigned n_scen = ((unsigned)B + 7u) / 8u;          // blocks of 8 scenarios; queue q takes one of each
    const unsigned K = n_scen * uW;
    const bool lane0 = (threadIdx.x & 63) == 0;
    const unsigned hold = ((P.dev >> 12) & 15u ? (P.dev >> 12) & 15u : 4u) * (gridDim.x / (unsigned)queues + 1u);
    const unsigned late_from = K > hold ? K - hold : 0u;
    // own queue first, then the other XCDs' queues in turn (the XCDs are not equally fast: one of the eight took 10 %
    // longer over the same work in every trace).  Item k of a queue is scenario ordinal j and slice p, through the
    // longest-first order when one was built.
    // The next index is fetched while the current unit is rolled -- but an index taken is an item reserved: towards the
    // end of a queue a wave in a long unit would sit on an item that idle waves could run (measured at B = 4096: waves
    // started leaving at 60 % of the kernel's span with items still held; units last 16 .. 130 us).  So over the last
    // four items per wave of the queue, and when stealing, the index is fetched only when the wave is ready for it.
    for (unsigned d = 0; d < (unsigned)queues; ++d) {
        const unsigned qq = (q + d) % (unsigned)queues;
        if (d > 0 && (P.dev & 512)) break;    // developer switch: no stealing
        unsigned* counter = work_counter + qq * 64u;
        const unsigned* ord = order ? order + (size_t)qq * order_stride : nullptr;
        unsigned k = 0, item = 0;
        if (lane0) {
            k = atomicAdd(counter, 1u);
            item = (ord && k < K) ? ord[k] : 0u;
        }
        k = __builtin_amdgcn_readfirstlane(k);
        item = __builtin_amdgcn_readfirstlane(item);
        while (k < K) {                       // every wave gets there: the counters only grow
            unsigned nxt = 0, nxt_item = 0;
            const bool early = d == 0 && k < late_from;
            if (early && lane0) {
                nxt = atomicAdd(counter, 1u);
                nxt_item = (ord && nxt < K) ? ord[nxt] : 0u;
            }
            const unsigned j = ord ? item >> 8 : k / uW, p = ord ? item & 255u : k - (k / uW) * uW;
            const unsigned long long t0 = (P.dev & 256) ? wall_clock64() : 0ull;
            const int b = queue_scenario((int)qq, (int)j);
            if (b < B) unit(b, (int)p);
            if ((P.dev & 256) && lane0) {      // developer trace (IGT_DEV_TRACE): when each unit ran, and where
                unsigned long long* tr =
                    reinterpret_cast<unsigned long long*>(work_counter + 1024) + ((size_t)qq * order_stride + k) * 4;
                tr[0] = t0; tr[1] = wall_clock64(); tr[2] = blockIdx.x; tr[3] = ((unsigned long long)j << 8) | p;
            }
            if (!early && lane0) {
                nxt = atomicAdd(counter, 1u);
                nxt_item = (ord && nxt < K) ? ord[nxt] : 0u;
            }
            k = __builtin_amdgcn_readfirstlane(nxt);
            item = __builtin_amdgcn_readfirstlane(nxt_item);
        }
    }
}

// The same loop built twice: 3 waves per SIMD (168 VGPRs, a 128 B/lane spill around each unit) keeps the VALU ~90 %
// busy on big batches; 2 per SIMD (no spill) rolls a unit in less wall time, which is what bounds a small batch.
#define IGT_SEARCH_ARGS                                                                                              \
    KP P, int B, int W, int queues, unsigned* __restrict__ work_counter, const unsigned* __restrict__ order,          \
        int order_stride, double* __restrict__ ckpt, int ck_parts, const float* __restrict__ x0,                        \
        const float* __restrict__ u_prev, const float* __restrict__ kparams, const uint32_t* __restrict__ flags,     \
        const float* __restrict__ obs, const double* __restrict__ table, const double* __restrict__ cinf,            \
        Centre<float> cpar, double* __restrict__ part_J, int32_t* __restrict__ part_c,                  \
        float* __restrict__ rec_sN, float* __restrict__ rec_vN, double* __restrict__ rec_J,                          \
        uint32_t* __restrict__ rec_viol, unsigned* __restrict__ rec_count, int32_t* __restrict__ rec_b
#define IGT_SEARCH_PASS                                                                                              \
    P, B, W, queues, work_counter, order, order_stride, ckpt, ck_parts, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J,    \
        part_c, rec_sN, rec_vN, rec_J, rec_viol, rec_count, rec_b
template <int CAND, bool HI, bool VALUE, bool CKPT>
__device__ __forceinline__ void search_waves_f32(IGT_SEARCH_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit<CAND, HI, VALUE, CKPT>(P, W, b, p, ckpt, ck_parts, B * W, x0, u_prev, kparams, flags, obs, table, cinf, cpar,
                                           part_J, part_c, rec_sN, rec_vN, rec_J, rec_viol, rec_count, rec_b);
    });
}
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void search_fast_kernel_o3(IGT_SEARCH_ARGS) {
    search_waves_f32<CAND, HI, VALUE, false>(IGT_SEARCH_PASS);
}
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) void search_fast_kernel_o2(IGT_SEARCH_ARGS) {
    search_waves_f32<CAND, HI, VALUE, false>(IGT_SEARCH_PASS);
}
// the same kernel held to 256 registers: the tracking family's steering feedback (sincos + atan2 per candidate and step)
// takes the plain build to 280 and with that to ONE wave per SIMD; a few dwords of scratch buy the second wave back
// (3.69 vs 5.94 ms at B = 65 536 together with igt_math64.h's atan2).  Not for the other families: the attribute alone
// costs the lattice kernels 5-7 % (different allocation, same register count).
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void search_fast_kernel_o2w(IGT_SEARCH_ARGS) {
    search_waves_f32<CAND, HI, VALUE, false>(IGT_SEARCH_PASS);
}
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) void search_fast_kernel_o2c(IGT_SEARCH_ARGS) {   // leaves horizon checkpoints (carries psi)
    search_waves_f32<CAND, HI, VALUE, true>(IGT_SEARCH_PASS);
}

template <int CAND, bool HI>
__global__ __launch_bounds__(64) void emit_fast_kernel(KP P, int B, int W, const float* __restrict__ x0,
                                                       const float* __restrict__ u_prev,
                                                       const float* __restrict__ kparams,
                                                       const uint32_t* __restrict__ flags,
                                                       const float* __restrict__ obs,
                                                       const double* __restrict__ table,
                                                       const double* __restrict__ cinf, Centre<float> cpar,
                                                       const double* __restrict__ part_J,
                                                       const int32_t* __restrict__ part_c,
                                                       float* __restrict__ cost_out, int32_t* __restrict__ argmin_out,
                                                       int32_t* __restrict__ status_out, float* __restrict__ x_out,
                                                       float* __restrict__ u_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // final arg-min over the W slices, (J, c) lexicographic: ties -> lowest candidate index whatever the slice order
    double bestJ = 0.0;
    int c = -1;
    for (int w = 0; w < W; ++w) {
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; }
    }
    cost_out[b] = c >= 0 ? (float)bestJ : INFINITY;
    argmin_out[b] = c;
    status_out[b] = c >= 0 ? 0 : 1;
    float* xo = x_out + (size_t)b * 7 * (P.N + 1);
    float* uo = u_out + (size_t)b * 2 * P.N;
    if (c < 0) {  // is_opt False (mpc.py:402-406): no trajectory
        for (int i = 0; i < 7 * (P.N + 1); ++i) xo[i] = NAN;
        for (int i = 0; i < 2 * P.N; ++i) uo[i] = NAN;
        return;
    }
    Scenario<float> S;
    load_scenario<float>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    // one candidate per lane (the unpacked build of the same arithmetic: a lone roll-out is latency, and half of
    // every packed instruction would carry a duplicate)
    PairSink<float> sink{{xo, nullptr}, {uo, nullptr}, P.N};
    const int cidx[1] = {c};
    double J[1], sN[1], vN[1];
    unsigned viol[1];
    single::rollout_pair<CAND, HI, false, false, float>(P, S, cidx, table, cinf, sink, J, viol, sN, vN);
}

// The same, from the search pass's checkpoints (small batches): `parts` lanes per scenario roll the pieces of the
// winner's horizon at once -- a lone roll-out is latency, and this is 1/parts of it.
template <int CAND, bool HI>
__global__ __launch_bounds__(64) void emit_seg_kernel(KP P, int B, int W, int Wk, int parts,
                                                      const float* __restrict__ x0,
                                                      const float* __restrict__ u_prev,
                                                      const float* __restrict__ kparams,
                                                      const uint32_t* __restrict__ flags,
                                                      const float* __restrict__ obs,
                                                      const double* __restrict__ table,
                                                      const double* __restrict__ cinf, Centre<float> cpar,
                                                      const double* __restrict__ part_J,
                                                      const int32_t* __restrict__ part_c,
                                                      const double* __restrict__ ckpt,
                                                      float* __restrict__ cost_out, int32_t* __restrict__ argmin_out,
                                                      int32_t* __restrict__ status_out, float* __restrict__ x_out,
                                                      float* __restrict__ u_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / parts, g = t - b * parts;
    if (b >= B) return;
    double bestJ = 0.0;
    int c = -1;
    for (int w = 0; w < W; ++w) {
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; }
    }
    float* xo = x_out + (size_t)b * 7 * (P.N + 1);
    float* uo = u_out + (size_t)b * 2 * P.N;
    if (g == 0) {
        cost_out[b] = c >= 0 ? (float)bestJ : INFINITY;
        argmin_out[b] = c;
        status_out[b] = c >= 0 ? 0 : 1;
        if (c < 0) {  // is_opt False (mpc.py:402-406): no trajectory
            for (int i = 0; i < 7 * (P.N + 1); ++i) xo[i] = NAN;
            for (int i = 0; i < 2 * P.N; ++i) uo[i] = NAN;
        }
    }
    if (c < 0) return;
    Scenario<float> S;
    load_scenario<float>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    int p, slot;
    candidate_slot<CAND>(P, Wk, c, p, slot);
    const int ns = P.N / parts;
    const size_t n_units = (size_t)B * Wk, unit = (size_t)b * Wk + p;
    const Seg seg{g * ns, (g + 1) * ns, g > 0 ? ckpt + ((size_t)(g - 1) * n_units + unit) * SEG_UNIT_DOUBLES : nullptr, slot};
    PairSink<float> sink{{xo, nullptr}, {uo, nullptr}, P.N};
    const int cidx[1] = {c};
    double J[1], sN[1], vN[1];
    unsigned viol[1];
    single::rollout_pair<CAND, HI, false, false, float, PairSink<float>, false, false, true>(
        P, S, cidx, table, cinf, sink, J, viol, sN, vN, Ckpt{nullptr, 0, 0, 0, 0}, seg);
}

template <int CAND, bool HI>
__global__ __launch_bounds__(256) void rollout_all_fast_kernel(KP P, int B, const float* __restrict__ x0,
                                                               const float* __restrict__ u_prev,
                                                               const float* __restrict__ kparams,
                                                               const uint32_t* __restrict__ flags,
                                                               const float* __restrict__ obs,
                                                               const double* __restrict__ table,
                                                               const double* __restrict__ cinf, Centre<float> cpar,
                                                               float* __restrict__ X_all, float* __restrict__ U_all,
                                                               float* __restrict__ cost_all,
                                                               uint32_t* __restrict__ viol_all,
                                                               float* __restrict__ rec_sN, float* __restrict__ rec_vN,
                                                               double* __restrict__ rec_J,
                                                               uint32_t* __restrict__ rec_viol) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    Scenario<float> S;
    load_scenario<float>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    for (int c = lane; c < P.C; c += 128) {
        const int cidx[2] = {c, c + 64 < P.C ? c + 64 : c};
        const size_t bc0 = (size_t)b * P.C + c, bc1 = (size_t)b * P.C + cidx[1];
        PairSink<float> sink{{X_all ? X_all + bc0 * 7 * (P.N + 1) : nullptr, X_all ? X_all + bc1 * 7 * (P.N + 1) : nullptr},
                             {U_all ? U_all + bc0 * 2 * P.N : nullptr, U_all ? U_all + bc1 * 2 * P.N : nullptr}, P.N};
        double J[2], sN[2], vN[2];
        unsigned viol[2];
        rollout_pair<CAND, HI, true, true, float>(P, S, cidx, table, cinf, sink, J, viol, sN, vN);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t bc = q ? bc1 : bc0;
            if (rec_J) {   // value-net cost: value_kernel adds the terminal term and fills cost_all / viol_all
                rec_sN[bc] = (float)sN[q]; rec_vN[bc] = (float)vN[q]; rec_J[bc] = J[q]; rec_viol[bc] = viol[q];
                continue;
            }
            const double Jq = J[q] - (sN[q] - S.x0[2]);
            if (!finite_d(Jq)) viol[q] |= VIOL_NONFINITE;
            cost_all[bc] = (float)Jq;
            viol_all[bc] = viol[q];
        }
    }
}

// ---------------------------------------------------------------------------------------
// float64 path (igt_fast64.h): one candidate per lane, W = C/64 units per scenario
// ---------------------------------------------------------------------------------------
// Generated G x G families with W * 64 == C and W | G: unit p takes G/W STEERING values (all G accelerations), handed out
// from the centre of the range outwards, as the float path does with its 128-candidate slices -- the extreme-steering
// units fail as a whole within a few steps and leave through the early exit.  Otherwise: chunks of 64 in index order.
template <int CAND>
__device__ __forceinline__ bool steering_slices64(const KP& P, int W) {
    return CAND != CAND_TABLE && P.G * P.G == P.C && W * 64 == P.C && P.G % W == 0 && !(P.dev & 1);
}
template <int CAND>
__device__ __forceinline__ int slice_candidate64(const KP& P, int W, int p, int lane) {
    if (steering_slices64<CAND>(P, W)) {
        const int nj = P.G / W, jl = lane % nj, il = lane / nj, r = p * nj + jl;      // il < 64 W / G = G
        const int j = (r & 1) ? P.G / 2 - 1 - (r >> 1) : P.G / 2 + (r >> 1);
        return il * P.G + j;
    }
    return p * 64 + lane;
}

#define IGT_SEARCH64_ARGS                                                                                            \
    KP P, int B, int W, int queues, unsigned* __restrict__ work_counter, const unsigned* __restrict__ order,          \
        int order_stride, const double* __restrict__ x0, const double* __restrict__ u_prev,                          \
        const double* __restrict__ kparams, const uint32_t* __restrict__ flags, const double* __restrict__ obs,      \
        const double* __restrict__ table, const double* __restrict__ cinf, Centre<double> cpar,          \
        double* __restrict__ part_J, int32_t* __restrict__ part_c, double* __restrict__ rec_sN,                      \
        double* __restrict__ rec_vN, double* __restrict__ rec_J, uint32_t* __restrict__ rec_viol,                    \
        unsigned* __restrict__ rec_count, int32_t* __restrict__ rec_b, int2* __restrict__ unit_seg

// One work unit = one (scenario, 64-candidate slice), rolled by one wave; leaves the slice's best (J, c), or -- value-net
// cost -- every candidate's record for value_kernel<double> (mpc.py:369).
template <int CAND, bool HI, bool VALUE>
__device__ __forceinline__ void search_unit64(const KP& P, int W, int b, int p, const double* __restrict__ x0,
                                              const double* __restrict__ u_prev, const double* __restrict__ kparams,
                                              const uint32_t* __restrict__ flags, const double* __restrict__ obs,
                                              const double* __restrict__ table, const double* __restrict__ cinf,
                                              Centre<double> cpar, double* __restrict__ part_J,
                                              int32_t* __restrict__ part_c, double* __restrict__ rec_sN,
                                              double* __restrict__ rec_vN, double* __restrict__ rec_J,
                                              uint32_t* __restrict__ rec_viol, unsigned* __restrict__ rec_count,
                                              int32_t* __restrict__ rec_b, int2* __restrict__ unit_seg) {
    const int lane = threadIdx.x & 63;
    Scenario<double> S;
    load_scenario<double>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    NullSink sink;
    const int c = slice_candidate64<CAND>(P, W, p, lane);
    double J, sN, vN;
    unsigned viol;
    // steering slices of the families with state-independent steering: the slice's G/W steering columns are laid out in
    // LDS once per unit instead of being recomputed by each of their 64 W/G lanes at every step (igt_fast64.h)
    constexpr bool TABULATED = CAND == CAND_LATTICE || CAND == CAND_RAMP_HOLD;
    const int nj = P.G / W;
    if (TABULATED && steering_slices64<CAND>(P, W) && nj * P.N <= f64::STAB_MAX_ENTRIES && !(P.dev & 4)) {
        __shared__ double stab[f64::STAB_MAX_ENTRIES * 3];
        f64::fill_steer_table<CAND>(P, S, nj, p, lane, P.lr_ratio, stab);
        f64::rollout_one<CAND, HI, true, true, NullSink, true, true>(P, S, c, table, cinf, sink, J, viol, sN, vN,
                                                                     stab + (lane % nj) * 3, nj * 3);
        __syncthreads();                                  // the next unit of this wave rewrites the table
    } else {
        f64::rollout_one<CAND, HI, true, true, NullSink, true>(P, S, c, table, cinf, sink, J, viol, sN, vN);
    }
    if (VALUE) {   // terminal value network (mpc.py:369): append the feasible candidates for value_mfma_f64_kernel; the
                   // unit's entries are contiguous, unit_seg remembers where (unit_reduce_kernel picks the unit's best)
        const bool ok = viol == 0 && finite_d(J);
        const unsigned long long m = __ballot(ok);
        const unsigned n = __popcll(m);
        unsigned base = 0;
        if (lane == 0 && n) base = atomicAdd(rec_count, n);
        base = __builtin_amdgcn_readfirstlane(base);
        if (ok) {
            const unsigned e = base + __popcll(m & ((1ull << lane) - 1ull));
            rec_b[e] = b; reinterpret_cast<int32_t*>(rec_viol)[e] = c; rec_sN[e] = sN; rec_vN[e] = vN; rec_J[e] = J;
        }
        if (lane == 0) unit_seg[b * W + p] = make_int2((int)base, (int)n);
        return;
    }
    const double Jq = J - (sN - S.x0[2]);                       // mpc.py:372
    double bestJ = Jq;
    int bestC = ((viol == 0) && finite_d(Jq)) ? c : -1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {                   // wave butterfly arg-min, ties -> lowest candidate index
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if (lane == 0) { part_J[b * W + p] = bestJ; part_c[b * W + p] = bestC; }
}

// persistent waves on the per-XCD queues (search_waves), 2 or 3 per SIMD like the float kernels
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) void search_f64_kernel_o2(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg);
    });
}
template <int CAND, bool HI, bool VALUE>      // held to 256 registers for the tracking family (see search_fast_kernel_o2w)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void search_f64_kernel_o2w(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg);
    });
}
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void search_f64_kernel_o3(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg);
    });
}

// one lane per scenario: final arg-min over the W partials, then the winner re-rolled with the same arithmetic
// (general sub-step variant: the lanes of the wave belong to different scenarios) -> x*[7,N+1], u*[2,N]
template <int CAND, bool HI>
__global__ __launch_bounds__(64) void emit_f64_kernel(KP P, int B, int W, const double* __restrict__ x0,
                                                      const double* __restrict__ u_prev,
                                                      const double* __restrict__ kparams,
                                                      const uint32_t* __restrict__ flags, const double* __restrict__ obs,
                                                      const double* __restrict__ table, const double* __restrict__ cinf,
                                                      Centre<double> cpar, const double* __restrict__ part_J,
                                                      const int32_t* __restrict__ part_c, double* __restrict__ cost_out,
                                                      int32_t* __restrict__ argmin_out, int32_t* __restrict__ status_out,
                                                      double* __restrict__ x_out, double* __restrict__ u_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double bestJ = 0.0;
    int c = -1;
    for (int w = 0; w < W; ++w) {      // (J, c) lexicographic: ties -> lowest candidate index whatever the slice order
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; }
    }
    cost_out[b] = c >= 0 ? bestJ : (double)INFINITY;
    argmin_out[b] = c;
    status_out[b] = c >= 0 ? 0 : 1;
    double* xo = x_out + (size_t)b * 7 * (P.N + 1);
    double* uo = u_out + (size_t)b * 2 * P.N;
    if (c < 0) {  // is_opt False (mpc.py:402-406): no trajectory
        for (int i = 0; i < 7 * (P.N + 1); ++i) xo[i] = (double)NAN;
        for (int i = 0; i < 2 * P.N; ++i) uo[i] = (double)NAN;
        return;
    }
    Scenario<double> S;
    load_scenario<double>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    StoreSink<double> sink{xo, uo, P.N};
    double J, sN, vN;
    unsigned viol;
    f64::rollout_one<CAND, HI, false, false, StoreSink<double>>(P, S, c, table, cinf, sink, J, viol, sN, vN);
}

template <int CAND, bool HI>
__global__ __launch_bounds__(256) void rollout_all_f64_kernel(KP P, int B, const double* __restrict__ x0,
                                                              const double* __restrict__ u_prev,
                                                              const double* __restrict__ kparams,
                                                              const uint32_t* __restrict__ flags,
                                                              const double* __restrict__ obs,
                                                              const double* __restrict__ table,
                                                              const double* __restrict__ cinf, Centre<double> cpar,
                                                              double* __restrict__ X_all, double* __restrict__ U_all,
                                                              double* __restrict__ cost_all, uint32_t* __restrict__ viol_all,
                                                              double* __restrict__ rec_sN, double* __restrict__ rec_vN,
                                                              double* __restrict__ rec_J, uint32_t* __restrict__ rec_viol) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    Scenario<double> S;
    load_scenario<double>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    for (int c = lane; c < P.C; c += 64) {          // C is a multiple of 64: the wave stays whole (votes inside)
        const size_t bc = (size_t)b * P.C + c;
        StoreSink<double> sink{X_all ? X_all + bc * 7 * (P.N + 1) : nullptr, U_all ? U_all + bc * 2 * P.N : nullptr, P.N};
        double J, sN, vN;
        unsigned viol;
        f64::rollout_one<CAND, HI, true, true, StoreSink<double>>(P, S, c, table, cinf, sink, J, viol, sN, vN);
        if (rec_J) {   // value-net cost: value_kernel adds the terminal term and fills cost_all / viol_all
            rec_sN[bc] = sN; rec_vN[bc] = vN; rec_J[bc] = J; rec_viol[bc] = viol;
            continue;
        }
        const double Jq = J - (sN - S.x0[2]);
        if (!finite_d(Jq)) viol |= VIOL_NONFINITE;
        cost_all[bc] = Jq;
        viol_all[bc] = viol;
    }
}

// ---------------------------------------------------------------------------------------
// The LITERAL mapping of BASELINE.json's north_star, kept as a measurement variant (IGT_DEV_FLAGS = 2048):
// one wavefront per (scenario, candidate) trajectory -- lane 0 rolls the horizon (the recurrence is sequential) and
// stages the states in LDS; then the wave evaluates the stage costs and verdicts stage-parallel (lane k = stage k,
// terminal-set facets spread over all 64 lanes), butterfly-reduces them, and the 16 waves of the scenario's workgroup
// take the arg-min over the candidates.  DESIGN.md section 3 has the numbers: the roll-out is 63/64 idle, so this is
// ~40x slower than one lane per candidate; it is NOT a production path (sum order differs from the oracle's).
// ---------------------------------------------------------------------------------------
struct LdsSink {
    static constexpr bool kKeepsStates = true;
    double* x;   // [7, N+1]
    double* u;   // [2, N]
    int N;
    __device__ __forceinline__ void ctrl(int, int k, double a, double df) { u[k] = a; u[N + k] = df; }
    __device__ __forceinline__ void state(int, int k, const double (&st)[7]) {
#pragma unroll
        for (int i = 0; i < 7; ++i) x[i * (N + 1) + k] = st[i];
    }
};
constexpr int LIT_WAVES = 16, LIT_MAX_N = 40;
template <int CAND, bool HI>
__global__ __launch_bounds__(64 * LIT_WAVES) void search_literal_f64_kernel(
    KP P, int B, int W, const double* __restrict__ x0, const double* __restrict__ u_prev, const double* __restrict__ kparams,
    const uint32_t* __restrict__ flags, const double* __restrict__ obs, const double* __restrict__ table,
    const double* __restrict__ cinf, Centre<double> cpar, double* __restrict__ part_J, int32_t* __restrict__ part_c) {
    __shared__ double lds[LIT_WAVES][9 * (LIT_MAX_N + 1)];
    __shared__ double wJ[LIT_WAVES];
    __shared__ int wC[LIT_WAVES];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Scenario<double> S;
    load_scenario<double>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    double* X = lds[wave];
    double* U = X + 7 * (P.N + 1);
    double bestJ = 0.0;
    int bestC = -1;
    for (int c = wave; c < P.C; c += LIT_WAVES) {
        if (lane == 0) {                       // the trajectory: one lane, the other 63 wait
            LdsSink sink{X, U, P.N};
            double J, sN, vN;
            unsigned viol;
            f64::rollout_one<CAND, HI, false, true, LdsSink>(P, S, c, table, cinf, sink, J, viol, sN, vN);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // stage-parallel cost (mpc.py:361-364) and verdicts (mpc.py:296-299, 316-317, 223-226): lane k = stage k
        double part = 0.0, g = -1.0e300;
        for (int k = lane; k <= P.N; k += 64) {
            const double ey = X[3 * (P.N + 1) + k], ep = X[4 * (P.N + 1) + k], v = X[5 * (P.N + 1) + k];
            part += ep * ep + ey * ey;
            g = fmax(g, fabs(ey) - P.ey_lim);
            if (k < P.N) {
                const double a = U[k], df = U[P.N + k];
                part += P.w_u * (a * a + df * df);
                g = fmax(g, fmax(P.v_min - v, v - P.v_max));
            }
            if (k >= 1)
                for (int o = 0; o < P.n_obs; ++o) {
                    const double dx = X[k] - S.obs[(o * 2 + 0) * (P.N + 1) + k], dy = X[(P.N + 1) + k] - S.obs[(o * 2 + 1) * (P.N + 1) + k];
                    g = fmax(g, P.dmin2 - (dx * dx + dy * dy));
                }
        }
        {   // terminal set (mpc.py:177-180): the facets over the lanes
            const double vt = X[5 * (P.N + 1) + P.N - 1], at = U[P.N - 1];
            for (int m = lane; m < P.F; m += 64) g = fmax(g, cinf[m * 3 + 0] * vt + cinf[m * 3 + 1] * at - cinf[m * 3 + 2]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            part += __shfl_xor(part, off, 64);
            g = fmax(g, __shfl_xor(g, off, 64));
        }
        const double Jq = part - (X[2 * (P.N + 1) + P.N] - S.x0[2]);          // mpc.py:372
        if (g <= P.tol && finite_d(Jq) && (bestC < 0 || Jq < bestJ)) { bestJ = Jq; bestC = c; }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { wJ[wave] = bestJ; wC[wave] = bestC; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double J = 0.0;
        int c = -1;
        for (int w = 0; w < LIT_WAVES; ++w)
            if (wC[w] >= 0 && (c < 0 || wJ[w] < J || (wJ[w] == J && wC[w] < c))) { J = wJ[w]; c = wC[w]; }
        part_J[(size_t)b * W] = J; part_c[(size_t)b * W] = c;
        for (int w = 1; w < W; ++w) part_c[(size_t)b * W + w] = -1;
    }
}

// one control step for n independent states (kinematic_bicycle_model_frenet.py:70-127)
template <class Stepper, typename T>
__global__ __launch_bounds__(256) void frenet_step_kernel(KP P, int n, const T* __restrict__ x,
                                                          const T* __restrict__ u, const T* __restrict__ kparams,
                                                          T* __restrict__ x_next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x0[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x0[k] = (double)x[(size_t)i * 7 + k];
    Stepper stp;
    stp.init(P, (double)kparams[(size_t)i * 3 + 0], (double)kparams[(size_t)i * 3 + 1],
             (double)kparams[(size_t)i * 3 + 2]);
    typename Stepper::State st;
    stp.set(st, x0);
    const double a = (double)u[(size_t)i * 2 + 0], df = (double)u[(size_t)i * 2 + 1];
    stp.step(st, a, stp.prep(df));
    double o[7];
    stp.get(st, o);
#pragma unroll
    for (int k = 0; k < 7; ++k) x_next[(size_t)i * 7 + k] = (T)o[k];
}

// ---------------------------------------------------------------------------------------
// opponent forecast (constant_acceleration_model.py:18-82, utils.py:339-352, 365-388, 532-586)
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void frenet2global_dev(const double* __restrict__ r, T s, T& x, T& y) {
    // r = (p0x, p0y, tx, ty, cx, cy, b0, b1, R, endx, endy, straight)
    const T p0x = (T)r[0], p0y = (T)r[1], tx = (T)r[2], ty = (T)r[3], cx = (T)r[4], cy = (T)r[5];
    const T b0 = (T)r[6], b1 = (T)r[7], R = (T)r[8];
    const bool straight = r[11] != 0.0;
    T along = s, cross = 0;
    bool post = false;
    if (!straight && !(s < b0)) {
        post = s > b1;
        const T ph = (s - b0) / R;
        T sn, cs;
        sincos_t<T>(ph, &sn, &cs);
        along = b0 + R * sn;
        cross = post ? (s - b1 + R) : R * ((T)1 - cs);
    }
    if (post) {   // the reference freezes the along-coordinate at the end value of its reference path
        const T ea = (T)r[9] * (tx < 0 ? -tx : tx) + (T)r[10] * (ty < 0 ? -ty : ty);
        const T acx = cx < 0 ? -cx : cx, acy = cy < 0 ? -cy : cy;
        x = ea * (tx < 0 ? -tx : tx) + p0x * acx + cross * cx;
        y = ea * (ty < 0 ? -ty : ty) + p0y * acy + cross * cy;
    } else {
        x = p0x + along * tx + cross * cx;
        y = p0y + along * ty + cross * cy;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void forecast_kernel(KP P, int B, const double* __restrict__ routes, int n_routes,
                                                       const T* __restrict__ ego_xyh, const T* __restrict__ opp,
                                                       const T* __restrict__ opp_a,
                                                       const int32_t* __restrict__ opp_route,
                                                       const T* __restrict__ plan_x, const T* __restrict__ plan_u,
                                                       const int32_t* __restrict__ has_plan, T* __restrict__ obs_xy,
                                                       T* __restrict__ tv_sv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = P.N;
    const T dt = (T)P.dt;
    int rid = opp_route[b];
    rid = rid < 0 ? 0 : (rid >= n_routes ? n_routes - 1 : rid);
    const double* __restrict__ r = routes + (size_t)rid * 12;
    T* ox = obs_xy + (size_t)b * 2 * (N + 1);
    T* oy = ox + (N + 1);
    const bool shared = plan_x && has_plan && has_plan[b] != 0;
    T x0, y0, sl, vl;
    if (shared) {
        // utils.py:339-352: plan states k = 1..N, then one predicted step from the plan's last state
        const T* px = plan_x + (size_t)b * 7 * (N + 1);
        for (int k = 0; k < N; ++k) { ox[k] = px[0 * (N + 1) + k + 1]; oy[k] = px[1 * (N + 1) + k + 1]; }
        T s = px[2 * (N + 1) + N], v = px[5 * (N + 1) + N];
        T a = plan_u[(size_t)b * 2 * N + (N - 1)];
        T vn = fmin(fmax(v + a * dt, (T)-2), (T)20);                       // cam:71
        if (vn > (T)5) { a = 0; vn = fmin(fmax(v, (T)-2), (T)20); }         // utils.py:348-349
        s = s + (v * dt + (T)0.5 * a * dt * dt);                           // cam:70
        T xe, ye;
        frenet2global_dev<T>(r, s, xe, ye);
        ox[N] = xe; oy[N] = ye;
        sl = s; vl = vn;
        x0 = ox[0]; y0 = oy[0];
    } else {
        T s = opp[(size_t)b * 4 + 2], v = opp[(size_t)b * 4 + 3];
        const T a = opp_a[b];
        x0 = opp[(size_t)b * 4 + 0]; y0 = opp[(size_t)b * 4 + 1];
        ox[0] = x0; oy[0] = y0;                                            // cam:40: k = 0 is the true state
        for (int k = 1; k <= N; ++k) {
            s = s + (v * dt + (T)0.5 * a * dt * dt);                       // cam:70
            v = fmin(fmax(v + a * dt, (T)-2), (T)20);                      // cam:71 (fourwayint.yaml:23-24)
            T xk, yk;
            frenet2global_dev<T>(r, s, xk, yk);                            // cam:75
            ox[k] = xk; oy[k] = yk;
        }
        sl = s; vl = v;
    }
    tv_sv[(size_t)b * 2 + 0] = sl;
    tv_sv[(size_t)b * 2 + 1] = vl;
    // filter_preds (utils.py:365-388)
    T sh, ch;
    sincos_t<T>(ego_xyh[(size_t)b * 3 + 2], &sh, &ch);
    const T dot = (x0 - ego_xyh[(size_t)b * 3 + 0]) * ch + (y0 - ego_xyh[(size_t)b * 3 + 1]) * sh;
    if (dot < 0)
        for (int k = 0; k <= N; ++k) { ox[k] = (T)-20; oy[k] = (T)-20; }
}

// float path of igt_frenet_step_f32: one control step through the same pair arithmetic the solver uses
template <bool HI>
__global__ __launch_bounds__(256) void frenet_step_fast_kernel(KP P, int n, const float* __restrict__ x,
                                                               const float* __restrict__ u,
                                                               const float* __restrict__ kparams,
                                                               float* __restrict__ x_next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KP P1 = P;
    P1.N = 1;
    P1.cand_mode = CAND_TABLE;
    Scenario<float> S;
#pragma unroll
    for (int k = 0; k < 7; ++k) S.x0[k] = (double)x[(size_t)i * 7 + k];
    S.a_prev = 0.0; S.df_prev = 0.0;
    S.b0 = (double)kparams[(size_t)i * 3 + 0]; S.b1 = (double)kparams[(size_t)i * 3 + 1]; S.kv = (double)kparams[(size_t)i * 3 + 2];
    S.cpar[0] = S.cpar[1] = S.cpar[2] = S.cpar[3] = 0.0;
    S.ws = nullptr;
    S.obs = nullptr;
    const double tab[2] = {(double)u[(size_t)i * 2 + 0], (double)u[(size_t)i * 2 + 1]};   // table [1 candidate, 2, N = 1]
    float out[7 * 2];
    PairSink<float> sink{{out, nullptr}, {nullptr, nullptr}, 1};
    const int cidx[2] = {0, 0};
    double J[2], sN[2], vN[2];
    unsigned viol[2];
    rollout_pair<CAND_TABLE, HI, false, false, float>(P1, S, cidx, tab, nullptr, sink, J, viol, sN, vN);
#pragma unroll
    for (int k = 0; k < 7; ++k) x_next[(size_t)i * 7 + k] = out[k * 2 + 1];
}

// kinematic_bicycle_model.py:27-31, T steps per trajectory
template <typename T>
__global__ __launch_bounds__(256) void cartesian_euler_kernel(int n, int steps, T dt, T l_r, T l_f,
                                                              const T* __restrict__ z0, const T* __restrict__ u,
                                                              T* __restrict__ z_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T x = z0[(size_t)i * 4 + 0], y = z0[(size_t)i * 4 + 1], psi = z0[(size_t)i * 4 + 2], v = z0[(size_t)i * 4 + 3];
    T* zo = z_out + (size_t)i * 4 * (steps + 1);
    zo[0] = x; zo[steps + 1] = y; zo[2 * (steps + 1)] = psi; zo[3 * (steps + 1)] = v;
    const T ratio = l_r / (l_f + l_r);
    for (int k = 0; k < steps; ++k) {
        const T a = u[((size_t)i * 2 + 0) * steps + k];
        const T df = u[((size_t)i * 2 + 1) * steps + k];
        const T tdf = tan_t<T>(df);
        const T beta = atan_t<T>(ratio * tdf);                          // :27
        T sb, cb, s2, c2;
        sincos_t<T>(beta, &sb, &cb);
        sincos_t<T>(psi + beta, &s2, &c2);
        const T xn = x + dt * v * c2;                                    // :28
        const T yn = y + dt * v * s2;                                    // :29
        const T pn = psi + dt * (v * cb / (l_r + l_f) * tdf);            // :30
        const T vn = v + dt * a;                                         // :31
        x = xn; y = yn; psi = pn; v = vn;
        zo[k + 1] = x; zo[steps + 1 + k + 1] = y; zo[2 * (steps + 1) + k + 1] = psi; zo[3 * (steps + 1) + k + 1] = v;
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <bool VALUE>
static hipError_t launch_search_exact(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    typedef ExactStepper<double> St;
    const dim3 grid((B + 3) / 4), block(256);
    if (P.cand_mode != CAND_TABLE)
        hipLaunchKernelGGL((search_kernel<St, double, 1, true, VALUE>), grid, block, 0, st, P, B, A.x0, A.u_prev,
                           A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.cost_out, A.argmin_out, A.status_out,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol);
    else
        hipLaunchKernelGGL((search_kernel<St, double, 1, false, VALUE>), grid, block, 0, st, P, B, A.x0, A.u_prev,
                           A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.cost_out, A.argmin_out, A.status_out,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol);
    return hipGetLastError();
}

// whether launch_search_fast will run build_queues_kernel (which also zeroes the unit counters)
bool search_builds_queues(const KP& P, int B, const SolveArgs<float>& A) {
    const int W = (P.C + 127) / 128;
    return A.queue_order && W <= 256 && ((B + 7) / 8) * W <= QB_THREADS * QB_TRIPS && !(P.dev & 16);
}
bool search_builds_queues(const KP& P, int B, const SolveArgs<double>& A) {
    const int W = P.C / 64;
    return A.queue_order && W <= 256 && ((B + 7) / 8) * W <= QB_THREADS * QB_TRIPS && !(P.dev & 16) && !(P.dev & (1024 | 2048));
}

template <int CAND, bool HI, bool VALUE>
static hipError_t launch_search_fast(const KP& P, int B, const SolveArgs<float>& A, hipStream_t st) {
    const int W = (P.C + 127) / 128;
    const size_t total = (size_t)B * W;
    // persistent waves on per-XCD queues, 2 per SIMD.  The 3-per-SIMD build (168 VGPRs, a spill around each unit) was
    // ahead on big batches in round 1; with the sub-step variants unrolled it spills 228 B/lane and is behind at every
    // size but one (B = 32 768: +0.7 %) -- it stays selectable for A/B runs (IGT_DEV_FLAGS = 32)
    const bool o3 = (P.dev & 32) != 0;
    const size_t slots = (size_t)A.n_cu * 4 * (o3 ? 3 : 2);
    const size_t grid = total < slots ? total : slots;
    const unsigned* order = nullptr;
    const int order_stride = ((B + 7) / 8) * W;
    if (search_builds_queues(P, B, A)) {                     // small batches: longest units first
        hipLaunchKernelGGL(build_queues_kernel<float>, dim3(8), dim3(QB_THREADS), 0, st, P, B, W, A.x0, A.kparams, A.queue_order,
                           order_stride, A.work_counter);
        order = A.queue_order;
    }
    if (o3)
        hipLaunchKernelGGL((search_fast_kernel_o3<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.rec_sN,
                           A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else if (A.ckpt && A.ck_parts > 1)
        hipLaunchKernelGGL((search_fast_kernel_o2c<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf,
                           A.centre(), A.part_J, A.part_c, A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else if constexpr (CAND == CAND_TRACK)
        hipLaunchKernelGGL((search_fast_kernel_o2w<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.rec_sN,
                           A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else
        hipLaunchKernelGGL((search_fast_kernel_o2<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.rec_sN,
                           A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    return hipGetLastError();
}
template <bool VALUE>
static hipError_t dispatch_search_fast(const KP& P, int B, const SolveArgs<float>& A, hipStream_t st) {
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_search_fast<CAND_LATTICE, true, VALUE>(P, B, A, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_search_fast<CAND_RAMP_HOLD, true, VALUE>(P, B, A, st);
        if (P.cand_mode == CAND_TRACK) return launch_search_fast<CAND_TRACK, true, VALUE>(P, B, A, st);
        return launch_search_fast<CAND_TABLE, true, VALUE>(P, B, A, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_search_fast<CAND_LATTICE, false, VALUE>(P, B, A, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_search_fast<CAND_RAMP_HOLD, false, VALUE>(P, B, A, st);
    if (P.cand_mode == CAND_TRACK) return launch_search_fast<CAND_TRACK, false, VALUE>(P, B, A, st);
    return launch_search_fast<CAND_TABLE, false, VALUE>(P, B, A, st);
}

template <>
hipError_t launch_search<float>(const KP& P, int B, const SolveArgs<float>& A, int, hipStream_t st) {
    return dispatch_search_fast<false>(P, B, A, st);
}
// float64 search: persistent waves on the per-XCD queues, one 64-candidate unit at a time
template <int CAND, bool HI, bool VALUE>
static hipError_t launch_search64(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    const int W = P.C / 64;
    if ((P.dev & 2048) && !VALUE && P.N <= LIT_MAX_N) {       // measurement variant: the literal wave-per-trajectory mapping
        hipLaunchKernelGGL((search_literal_f64_kernel<CAND, HI>), dim3(B), dim3(64 * LIT_WAVES), 0, st, P, B, W, A.x0, A.u_prev,
                           A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * W;
    // 2 waves per SIMD (232 VGPRs, no spill); the 3-per-SIMD build spills 244 B/lane and is 3-8 % behind at every batch
    // size (IGT_DEV_FLAGS = 32 selects it for A/B runs)
    const bool o3 = (P.dev & 32) != 0;
    const size_t slots = (size_t)A.n_cu * 4 * (o3 ? 3 : 2);
    const size_t grid = total < slots ? total : slots;
    const unsigned* order = nullptr;
    const int order_stride = ((B + 7) / 8) * W;
    if (search_builds_queues(P, B, A)) {                     // small batches: longest units first
        hipLaunchKernelGGL(build_queues_kernel<double>, dim3(8), dim3(QB_THREADS), 0, st, P, B, W, A.x0, A.kparams,
                           A.queue_order, order_stride, A.work_counter);
        order = A.queue_order;
    }
    if (o3)
        hipLaunchKernelGGL((search_f64_kernel_o3<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg);
    else if constexpr (CAND == CAND_TRACK)
        hipLaunchKernelGGL((search_f64_kernel_o2w<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg);
    else
        hipLaunchKernelGGL((search_f64_kernel_o2<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg);
    return hipGetLastError();
}
template <bool VALUE>
static hipError_t dispatch_search64(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    if (P.dev & 1024) return launch_search_exact<VALUE>(P, B, A, st);     // developer switch: oracle-order kernels
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_search64<CAND_LATTICE, true, VALUE>(P, B, A, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_search64<CAND_RAMP_HOLD, true, VALUE>(P, B, A, st);
        if (P.cand_mode == CAND_TRACK) return launch_search64<CAND_TRACK, true, VALUE>(P, B, A, st);
        return launch_search64<CAND_TABLE, true, VALUE>(P, B, A, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_search64<CAND_LATTICE, false, VALUE>(P, B, A, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_search64<CAND_RAMP_HOLD, false, VALUE>(P, B, A, st);
    if (P.cand_mode == CAND_TRACK) return launch_search64<CAND_TRACK, false, VALUE>(P, B, A, st);
    return launch_search64<CAND_TABLE, false, VALUE>(P, B, A, st);
}
template <>
hipError_t launch_search<double>(const KP& P, int B, const SolveArgs<double>& A, int, hipStream_t st) {
    return dispatch_search64<false>(P, B, A, st);
}
template <>
hipError_t launch_search_records<float>(const KP& P, int B, const SolveArgs<float>& A, hipStream_t st) {
    return dispatch_search_fast<true>(P, B, A, st);
}
template <>
hipError_t launch_search_records<double>(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    return dispatch_search64<true>(P, B, A, st);
}

// dynamic-LDS limits of the value kernels, raised once when a net is loaded (igt_set_value_net) -- not on the launch
// path, which must stay a pure sequence of stream operations (stream capture)
hipError_t prepare_value_kernels(int n_hidden_mats) {
    const size_t lds_d = (size_t)VN_H * 64 * sizeof(double) * (n_hidden_mats > 1 ? 2 : 1);
    hipError_t e = hipSuccess;
    if (lds_d > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&value_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&value_mfma_f64_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FRAGD_LDS * sizeof(double)));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&value_mfma_f64_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FRAGD_LDS * sizeof(double)));
    if (e != hipSuccess) return e;
    const size_t lds_f = (size_t)frag_floats(n_hidden_mats) * sizeof(float);
    if (n_hidden_mats > 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&value_mfma_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&value_mfma_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
}

template <typename T>
hipError_t launch_value(const KP& P, int B, const DevNet<T>& net, const SolveArgs<T>& A, T* cost_all,
                        uint32_t* viol_all, hipStream_t st);
template <>
hipError_t launch_value<double>(const KP& P, int B, const DevNet<double>& net, const SolveArgs<double>& A,
                                double* cost_all, uint32_t* viol_all, hipStream_t st) {
    if (!cost_all && !(P.dev & 1024)) {   // solve path: the compact list of feasible candidates on the f64 matrix cores
        const size_t lds = (size_t)FRAGD_LDS * sizeof(double);
        if (net.n_hidden_mats > 1)
            hipLaunchKernelGGL(value_mfma_f64_kernel<2>, dim3(A.n_cu), dim3(512), lds, st, net, A.rec_count, A.rec_b, A.rec_sN,
                               A.rec_vN, A.rec_J, A.tv_sv, A.enc);
        else
            hipLaunchKernelGGL(value_mfma_f64_kernel<1>, dim3(A.n_cu), dim3(512), lds, st, net, A.rec_count, A.rec_b, A.rec_sN,
                               A.rec_vN, A.rec_J, A.tv_sv, A.enc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int n_units = B * (P.C / 64);
        hipLaunchKernelGGL(unit_reduce_kernel, dim3((n_units + 255) / 256), dim3(256), 0, st, n_units, A.unit_seg, A.rec_J,
                           reinterpret_cast<const int32_t*>(A.rec_viol), A.part_J, A.part_c);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((value_prep_kernel<double>), dim3(B), dim3(VN_H), 0, st, B, net, A.tv_sv, A.enc, A.p_vec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)VN_H * 64 * sizeof(double) * (net.n_hidden_mats > 1 ? 2 : 1);
    hipLaunchKernelGGL((value_kernel<double>), dim3((size_t)B * (P.C / 64)), dim3(64), lds, st, B, P.C, net, A.p_vec,
                       A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.part_J, A.part_c, cost_all, viol_all);
    return hipGetLastError();
}
template <>
hipError_t launch_value<float>(const KP& P, int B, const DevNet<float>& net, const SolveArgs<float>& A, float* cost_all,
                               uint32_t* viol_all, hipStream_t st) {
    if (!cost_all) {   // solve path: walk the compact list of feasible candidates
        CompactRecs R{A.rec_count, A.rec_b, reinterpret_cast<const int32_t*>(A.rec_viol), A.rec_sN, A.rec_vN, A.rec_J};
        // one 8-wave workgroup per CU (the weight fragments take 68 / 134 KB of its LDS), grid-stride over the list
        const size_t lds = (size_t)frag_floats(net.n_hidden_mats) * sizeof(float);
        if (net.n_hidden_mats > 1)
            hipLaunchKernelGGL(value_mfma_kernel<2>, dim3(A.n_cu), dim3(512), lds, st, net, R, A.tv_sv, A.enc, A.best_key);
        else
            hipLaunchKernelGGL(value_mfma_kernel<1>, dim3(A.n_cu), dim3(512), lds, st, net, R, A.tv_sv, A.enc, A.best_key);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(keys_to_partials_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, A.best_key, A.part_J, A.part_c);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((value_prep_kernel<float>), dim3(B), dim3(VN_H), 0, st, B, net, A.tv_sv, A.enc, A.p_vec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 grid((size_t)B * (P.C / 64)), block(64);
    if (net.n_hidden_mats > 1)
        hipLaunchKernelGGL(value_kernel_f32_h3, grid, block, 0, st, B, P.C, net, A.p_vec, A.rec_sN, A.rec_vN, A.rec_J,
                           A.rec_viol, A.part_J, A.part_c, cost_all, viol_all);
    else
        hipLaunchKernelGGL(value_kernel_f32_h2, grid, block, 0, st, B, P.C, net, A.p_vec, A.rec_sN, A.rec_vN, A.rec_J,
                           A.rec_viol, A.part_J, A.part_c, cost_all, viol_all);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_refine(const KP& P, int B, int W, const SolveArgs<T>& A, double* cpar, int first, hipStream_t st) {
    hipLaunchKernelGGL(refine_targets_kernel, dim3((B + 255) / 256), dim3(256), 0, st, P, B, W, A.part_J, A.part_c, cpar, first);
    return hipGetLastError();
}
template hipError_t launch_refine<float>(const KP&, int, int, const SolveArgs<float>&, double*, int, hipStream_t);
template hipError_t launch_refine<double>(const KP&, int, int, const SolveArgs<double>&, double*, int, hipStream_t);

template <typename T>
hipError_t launch_reduce(int B, int W, const SolveArgs<T>& A, hipStream_t st) {
    hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, st, B, W, A.part_J, A.part_c,
                       A.cost_out, A.argmin_out, A.status_out);
    return hipGetLastError();
}
template hipError_t launch_reduce<float>(int, int, const SolveArgs<float>&, hipStream_t);
template hipError_t launch_reduce<double>(int, int, const SolveArgs<double>&, hipStream_t);

template <int CAND, bool HI>
static hipError_t launch_emit_fast(const KP& P, int B, int W, const SolveArgs<float>& A, hipStream_t st) {
    if (A.ckpt) {
        const int Wk = (P.C + 127) / 128;
        hipLaunchKernelGGL((emit_seg_kernel<CAND, HI>), dim3(((size_t)B * A.ck_parts + 63) / 64), dim3(64), 0, st, P, B, W, Wk,
                           A.ck_parts,
                           A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.ckpt,
                           A.cost_out, A.argmin_out, A.status_out, A.x_out, A.u_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((emit_fast_kernel<CAND, HI>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, W, A.x0, A.u_prev,
                       A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.cost_out, A.argmin_out,
                       A.status_out, A.x_out, A.u_out);
    return hipGetLastError();
}
// float path: W per-slice partials per scenario are reduced here; double path: argmin_out is already final
template <>
hipError_t launch_emit<float>(const KP& P, int B, int W, const SolveArgs<float>& A, hipStream_t st) {
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_emit_fast<CAND_LATTICE, true>(P, B, W, A, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_emit_fast<CAND_RAMP_HOLD, true>(P, B, W, A, st);
        if (P.cand_mode == CAND_TRACK) return launch_emit_fast<CAND_TRACK, true>(P, B, W, A, st);
        return launch_emit_fast<CAND_TABLE, true>(P, B, W, A, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_emit_fast<CAND_LATTICE, false>(P, B, W, A, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_emit_fast<CAND_RAMP_HOLD, false>(P, B, W, A, st);
    if (P.cand_mode == CAND_TRACK) return launch_emit_fast<CAND_TRACK, false>(P, B, W, A, st);
    return launch_emit_fast<CAND_TABLE, false>(P, B, W, A, st);
}
template <int CAND, bool HI>
static hipError_t launch_emit64(const KP& P, int B, int W, const SolveArgs<double>& A, hipStream_t st) {
    hipLaunchKernelGGL((emit_f64_kernel<CAND, HI>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, W, A.x0, A.u_prev, A.kparams,
                       A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.cost_out, A.argmin_out, A.status_out,
                       A.x_out, A.u_out);
    return hipGetLastError();
}
template <>
hipError_t launch_emit<double>(const KP& P, int B, int W, const SolveArgs<double>& A, hipStream_t st) {
    if (P.dev & 1024) {     // developer switch: oracle-order kernels (argmin_out is already final there)
        hipLaunchKernelGGL((emit_kernel<ExactStepper<double>, double>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, A.x0,
                           A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.argmin_out, A.x_out, A.u_out);
        return hipGetLastError();
    }
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_emit64<CAND_LATTICE, true>(P, B, W, A, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_emit64<CAND_RAMP_HOLD, true>(P, B, W, A, st);
        if (P.cand_mode == CAND_TRACK) return launch_emit64<CAND_TRACK, true>(P, B, W, A, st);
        return launch_emit64<CAND_TABLE, true>(P, B, W, A, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_emit64<CAND_LATTICE, false>(P, B, W, A, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_emit64<CAND_RAMP_HOLD, false>(P, B, W, A, st);
    if (P.cand_mode == CAND_TRACK) return launch_emit64<CAND_TRACK, false>(P, B, W, A, st);
    return launch_emit64<CAND_TABLE, false>(P, B, W, A, st);
}

template <int CAND, bool HI>
static hipError_t launch_rollout_all_fast(const KP& P, int B, const SolveArgs<float>& A, float* X_all, float* U_all,
                                          float* cost_all, uint32_t* viol_all, hipStream_t st) {
    hipLaunchKernelGGL((rollout_all_fast_kernel<CAND, HI>), dim3((B + 3) / 4), dim3(256), 0, st, P, B, A.x0, A.u_prev,
                       A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), X_all, U_all, cost_all, viol_all, A.rec_sN, A.rec_vN,
                       A.rec_J, A.rec_viol);
    return hipGetLastError();
}
template <>
hipError_t launch_rollout_all<float>(const KP& P, int B, const SolveArgs<float>& A, float* X_all, float* U_all,
                                     float* cost_all, uint32_t* viol_all, hipStream_t st) {
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_rollout_all_fast<CAND_LATTICE, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_rollout_all_fast<CAND_RAMP_HOLD, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        if (P.cand_mode == CAND_TRACK) return launch_rollout_all_fast<CAND_TRACK, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        return launch_rollout_all_fast<CAND_TABLE, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_rollout_all_fast<CAND_LATTICE, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_rollout_all_fast<CAND_RAMP_HOLD, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    if (P.cand_mode == CAND_TRACK) return launch_rollout_all_fast<CAND_TRACK, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    return launch_rollout_all_fast<CAND_TABLE, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
}
template <int CAND, bool HI>
static hipError_t launch_rollout_all64(const KP& P, int B, const SolveArgs<double>& A, double* X_all, double* U_all,
                                       double* cost_all, uint32_t* viol_all, hipStream_t st) {
    hipLaunchKernelGGL((rollout_all_f64_kernel<CAND, HI>), dim3((B + 3) / 4), dim3(256), 0, st, P, B, A.x0, A.u_prev,
                       A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), X_all, U_all, cost_all, viol_all, A.rec_sN, A.rec_vN,
                       A.rec_J, A.rec_viol);
    return hipGetLastError();
}
template <>
hipError_t launch_rollout_all<double>(const KP& P, int B, const SolveArgs<double>& A, double* X_all, double* U_all,
                                      double* cost_all, uint32_t* viol_all, hipStream_t st) {
    if (P.dev & 1024) {     // developer switch: oracle-order kernels
        hipLaunchKernelGGL((rollout_all_kernel<ExactStepper<double>, double>), dim3((B + 3) / 4), dim3(256), 0, st, P, B,
                           A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), X_all, U_all, cost_all, viol_all,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol);
        return hipGetLastError();
    }
    if (P.hi_order) {
        if (P.cand_mode == CAND_LATTICE) return launch_rollout_all64<CAND_LATTICE, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        if (P.cand_mode == CAND_RAMP_HOLD) return launch_rollout_all64<CAND_RAMP_HOLD, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        if (P.cand_mode == CAND_TRACK) return launch_rollout_all64<CAND_TRACK, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
        return launch_rollout_all64<CAND_TABLE, true>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    }
    if (P.cand_mode == CAND_LATTICE) return launch_rollout_all64<CAND_LATTICE, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    if (P.cand_mode == CAND_RAMP_HOLD) return launch_rollout_all64<CAND_RAMP_HOLD, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    if (P.cand_mode == CAND_TRACK) return launch_rollout_all64<CAND_TRACK, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
    return launch_rollout_all64<CAND_TABLE, false>(P, B, A, X_all, U_all, cost_all, viol_all, st);
}

template <>
hipError_t launch_frenet_step<float>(const KP& P, int n, const float* x, const float* u, const float* kparams,
                                     float* x_next, hipStream_t st) {
    // always the long stage-offset polynomials: this entry has no verdicts, so a caller may step states far outside the
    // planner's speed box (the predictor allows v up to 20 m/s), where the short forms would leave the 1e-5 envelope
    (void)P.hi_order;
    hipLaunchKernelGGL((frenet_step_fast_kernel<true>), dim3((n + 255) / 256), dim3(256), 0, st, P, n, x, u, kparams, x_next);
    return hipGetLastError();
}
template <>
hipError_t launch_frenet_step<double>(const KP& P, int n, const double* x, const double* u, const double* kparams,
                                      double* x_next, hipStream_t st) {
    hipLaunchKernelGGL((frenet_step_kernel<ExactStepper<double>, double>), dim3((n + 255) / 256), dim3(256), 0, st, P,
                       n, x, u, kparams, x_next);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_forecast(const KP& P, int B, const double* routes, int n_routes, const T* ego_xyh, const T* opp,
                           const T* opp_a, const int32_t* opp_route, const T* plan_x, const T* plan_u,
                           const int32_t* has_plan, T* obs_xy, T* tv_sv, hipStream_t st) {
    hipLaunchKernelGGL((forecast_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, st, P, B, routes, n_routes, ego_xyh, opp,
                       opp_a, opp_route, plan_x, plan_u, has_plan, obs_xy, tv_sv);
    return hipGetLastError();
}
template hipError_t launch_forecast<float>(const KP&, int, const double*, int, const float*, const float*, const float*,
                                           const int32_t*, const float*, const float*, const int32_t*, float*, float*,
                                           hipStream_t);
template hipError_t launch_forecast<double>(const KP&, int, const double*, int, const double*, const double*, const double*,
                                            const int32_t*, const double*, const double*, const int32_t*, double*, double*,
                                            hipStream_t);

// u[B,2,N] -> u0[B,2] = u[:, :, 0]: the (a, df) each agent applies (evaluate.py:492), contiguous for the all-gather
template <typename T>
__global__ __launch_bounds__(256) void first_controls_kernel(int B, int N, const T* __restrict__ u, T* __restrict__ u0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * B) u0[i] = u[(size_t)i * N];
}
template <typename T>
hipError_t launch_first_controls(int B, int N, const T* u, T* u0, hipStream_t st) {
    hipLaunchKernelGGL((first_controls_kernel<T>), dim3((2 * B + 255) / 256), dim3(256), 0, st, B, N, u, u0);
    return hipGetLastError();
}
template hipError_t launch_first_controls<float>(int, int, const float*, float*, hipStream_t);
template hipError_t launch_first_controls<double>(int, int, const double*, double*, hipStream_t);

template <typename T>
hipError_t launch_cartesian(int n, int steps, double dt, double l_r, double l_f, const T* z0, const T* u, T* z_out,
                            hipStream_t st) {
    hipLaunchKernelGGL((cartesian_euler_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, n, steps, (T)dt, (T)l_r,
                       (T)l_f, z0, u, z_out);
    return hipGetLastError();
}
template hipError_t launch_cartesian<float>(int, int, double, double, double, const float*, const float*, float*, hipStream_t);
template hipError_t launch_cartesian<double>(int, int, double, double, double, const double*, const double*, double*, hipStream_t);

}  // namespace igt
