// igt_kernels.hip -- gfx950 kernels of the batched shooting solver and their launchers (float path and everything that is not
// the float64 search: igt_kernels_f64.hip holds that).
//
//   search_fast_kernel_*   persistent waves, one 128-candidate steering slice of one scenario per unit, two candidates per
//                          lane as packed float pairs (igt_fast.h)
//   emit_fast_kernel / emit_seg_kernel   the winner re-rolled (from checkpoints of the search pass for small batches)
//   rollout_all_fast_kernel   debug / parity: every candidate's trajectory, cost and verdict bits
//   reduce / refine kernels, the value-network launchers (igt_value_net.h), forecast_kernel, frenet_step kernels,
//   cartesian_euler_kernel (kinematic_bicycle_model.py:15-50), first_controls_kernel (all-gather staging)
#define IGT_KERNELS_TU 1
#include "igt_device.h"
#include "igt_fast.h"
#include "igt_launch.h"
#include "igt_kernels_common.h"

namespace igt {

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
// Tracking family (its steering is a feedback on the rolled state: its candidates fail on what the ACCELERATION row decides,
// igt_kernels_f64.hip): unit p takes the G / W acceleration rows G-1 - p G/W downwards with all G steering offsets -- the rows
// that make most progress in unit 0, whose best cost is the incumbent the later units are pruned against (igt_device.h,
// igt_fast64.h BOUND).  A lane holds (row, j) and (row - 64 / G, j).  G in {16, 32, 64}; IGT_DEV_FLAGS = 262144: steering slices.
constexpr int DEV_INCUMBENTS = 1 << 30;      // launch-time bit of KP::dev (float kernels): [B] incumbent keys behind the partials
template <int CAND>
__device__ __forceinline__ bool accel_units(const KP& P, int W) {
    return CAND == CAND_TRACK && P.G * P.G == P.C && W * 128 == P.C && P.G >= 16 && P.G <= 64 && !(P.dev & (1 | 262144));
}
template <int CAND>
__device__ __forceinline__ void slice_candidates(const KP& P, int W, int p, int lane, int (&cidx)[2]) {
    if (accel_units<CAND>(P, W)) {
        const int nr = P.G / W, j = lane % P.G, rl = lane / P.G, top = P.G - 1 - p * nr;
        cidx[0] = (top - rl) * P.G + j;
        cidx[1] = (top - rl - nr / 2) * P.G + j;
        return;
    }
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
    if (accel_units<CAND>(P, W)) {
        const int i = c / P.G, j = c - i * P.G, nr = P.G / W, down = P.G - 1 - i;
        p = down / nr;
        const int d = down - p * nr, q = d >= nr / 2 ? 1 : 0;
        slot = 64 * q + (d - q * (nr / 2)) * P.G + j;
        return;
    }
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
    // the scenario's incumbent (tracking family, progress cost): [B] keys behind the [B W] partials
    unsigned long long* inc = (CAND == CAND_TRACK && !VALUE && (P.dev & DEV_INCUMBENTS))
                                  ? reinterpret_cast<unsigned long long*>(part_J + n_units) + b : nullptr;
    // units whose obstacles are out of every speed-feasible candidate's reach roll without the Cartesian rows (igt_device.h
    // obstacles_out_of_reach); the builds that leave checkpoints for emit need x, y
    if (!CKPT && !(P.dev & 65536) && obstacles_out_of_reach<float>(P, S, lane))
        rollout_pair<CAND, HI, true, true, float, NullSink, true, false, false, false>(P, S, cidx, table, cinf, sink, J, viol, sN, vN, ck,
                                                                                       Seg{0, 0, nullptr, 0}, inc);
    else
        rollout_pair<CAND, HI, true, true, float, NullSink, true, CKPT>(P, S, cidx, table, cinf, sink, J, viol, sN, vN, ck,
                                                                        Seg{0, 0, nullptr, 0}, inc);
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
        // (J, c) lexicographic: the tracking family's units hold the higher row first, and rows clipped to the same envelope
        // tie exactly -- the lowest index wins whatever the order
        if (ok && (bestC < 0 || Jq < bestJ || (Jq == bestJ && cidx[q] < bestC))) { bestJ = Jq; bestC = cidx[q]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oJ = __shfl_xor(bestJ, off, 64);
        const int oC = __shfl_xor(bestC, off, 64);
        const bool take = (oC >= 0) && (bestC < 0 || oJ < bestJ || (oJ == bestJ && oC < bestC));
        if (take) { bestJ = oJ; bestC = oC; }
    }
    if (lane == 0) {
        part_J[gw] = bestJ; part_c[gw] = bestC;
        if (inc && bestC >= 0) atomicMin(inc, cost_key(bestJ));      // the scenario's incumbent from now on (device scope)
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
    }, CAND == CAND_TRACK && !VALUE && (P.dev & DEV_INCUMBENTS) != 0);      // unit-rank-major items: every unit 0 before any unit 1
}
#if IGT_DEV_KERNELS
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void search_fast_kernel_o3(IGT_SEARCH_ARGS) {
    search_waves_f32<CAND, HI, VALUE, false>(IGT_SEARCH_PASS);
}
#endif
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
    // one lane per (problem, other vehicle): pair = b n_obs + o -- the M - 1 obstacles of a scene (mpc.py:82-83 is written for any
    // M) are forecast, shared and filtered independently of each other; `b` below indexes the per-pair arrays, `e` the ego's
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int e = P.n_obs > 1 ? b / P.n_obs : b;
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
    tv_sv[(size_t)b * 2 + 0] = sl;           // last raw prediction of every other vehicle (mpc.py:263-276 raw_preds; the value
    tv_sv[(size_t)b * 2 + 1] = vl;           // network's features read the one of a two-vehicle scene, mpc.py:330)
    // filter_preds (utils.py:365-388)
    T sh, ch;
    sincos_t<T>(ego_xyh[(size_t)e * 3 + 2], &sh, &ch);
    const T dot = (x0 - ego_xyh[(size_t)e * 3 + 0]) * ch + (y0 - ego_xyh[(size_t)e * 3 + 1]) * sh;
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
// whether launch_search_fast will run build_queues_kernel (which also zeroes the unit counters)
bool search_builds_queues(const KP& P, int B, const SolveArgs<float>& A) {
    const int W = (P.C + 127) / 128;
    return A.queue_order && W <= 256 && ((B + 7) / 8) * W <= QB_THREADS * QB_TRIPS && !(P.dev & 16);
}
template <int CAND, bool HI, bool VALUE>
static hipError_t launch_search_fast(const KP& P, int B, const SolveArgs<float>& A, hipStream_t st) {
    const int W = (P.C + 127) / 128;
    const size_t total = (size_t)B * W;
    // persistent waves on per-XCD queues, 2 per SIMD.  The 3-per-SIMD build (168 VGPRs, a spill around each unit) was
    // ahead on big batches in round 1; with the sub-step variants unrolled it spills 228 B/lane and is behind at every
    // size but one (B = 32 768: +0.7 %) -- it stays selectable for A/B runs (IGT_DEV_FLAGS = 32)
    const bool o3 = IGT_DEV_KERNELS && (P.dev & 32) != 0;
    const size_t slots = (size_t)A.n_cu * 4 * (o3 ? 3 : (A.waves_per_simd == 1 ? 1 : 2));
    const size_t grid = total < slots ? total : slots;
    const unsigned* order = nullptr;
    const int order_stride = ((B + 7) / 8) * W;
    KP Pr = P;
    if constexpr (CAND == CAND_TRACK && !VALUE) {
        // tracking family, progress cost: units of acceleration rows, pruned against the scenario's incumbent (igt_fast64.h
        // BOUND).  The [B] keys behind the partials start every pass at "none" (all ones: above every cost's key).
        if (P.G * P.G == P.C && W * 128 == P.C && P.G >= 16 && P.G <= 64 && !(P.dev & (1 | 262144 | 8388608)) && W > 1) {
            hipError_t e = hipMemsetAsync(A.part_J + total, 0xff, (size_t)B * 8, st);
            if (e != hipSuccess) return e;
            Pr.dev |= DEV_INCUMBENTS;
        }
    }
    if (search_builds_queues(P, B, A)) {                     // small batches: longest units first
        hipLaunchKernelGGL(build_queues_kernel<float>, dim3(8), dim3(QB_THREADS), 0, st, Pr, B, W, A.x0, A.kparams, A.queue_order,
                           order_stride, A.work_counter);
        order = A.queue_order;
    }
#if IGT_DEV_KERNELS
    if (o3)
        hipLaunchKernelGGL((search_fast_kernel_o3<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.rec_sN,
                           A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else
#endif
    if (A.ckpt && A.ck_parts > 1)
        hipLaunchKernelGGL((search_fast_kernel_o2c<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf,
                           A.centre(), A.part_J, A.part_c, A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else if constexpr (CAND == CAND_TRACK)
        hipLaunchKernelGGL((search_fast_kernel_o2w<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter,
                           order, order_stride, A.ckpt, A.ck_parts, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.rec_sN,
                           A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b);
    else
        hipLaunchKernelGGL((search_fast_kernel_o2<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter,
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
template <>
hipError_t launch_search_records<float>(const KP& P, int B, const SolveArgs<float>& A, hipStream_t st) {
    return dispatch_search_fast<true>(P, B, A, st);
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
        const int n_units = B * (P.C / 64);
        // prune the list first (value_bound_kernel): scenarios with many feasible candidates -- the tracking and ramp-hold
        // families -- keep a fraction of their entries; short lists (the lattice's 14 per scenario) are left alone
        // value_bound_kernel: scenarios with many feasible candidates -- the tracking and ramp-hold families -- keep a fraction
        // of their entries; short lists (the lattice's 14 per scenario) are left alone.  IGT_DEV_FLAGS = 131072: no pruning.
        const int min_entries = (P.dev & 131072) ? 0x7fffffff : 48;
        unsigned* live_count = A.rec_count + 16;          // zeroed with rec_count before the search
        hipLaunchKernelGGL(value_bound_kernel, dim3(B), dim3(64), 0, st, B, P.C / 64, min_entries, net.n_hidden_mats, net, A.unit_seg,
                           A.rec_sN, A.rec_vN, A.rec_J, A.tv_sv, A.enc, A.prune_thr);
        hipLaunchKernelGGL(value_select_kernel, dim3(A.n_cu * 4), dim3(256), 0, st, n_units, P.C / 64, A.unit_seg, A.rec_J,
                           A.prune_thr, live_count, A.live_idx);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        if (net.n_hidden_mats > 1)
            hipLaunchKernelGGL(value_mfma_f64_kernel<2>, dim3(A.n_cu), dim3(512), lds, st, net, live_count, A.live_idx, A.rec_b,
                               A.rec_sN, A.rec_vN, A.rec_J, A.tv_sv, A.enc);
        else
            hipLaunchKernelGGL(value_mfma_f64_kernel<1>, dim3(A.n_cu), dim3(512), lds, st, net, live_count, A.live_idx, A.rec_b,
                               A.rec_sN, A.rec_vN, A.rec_J, A.tv_sv, A.enc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
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
    const int pairs = B * (P.n_obs > 1 ? P.n_obs : 1);
    hipLaunchKernelGGL((forecast_kernel<T>), dim3((pairs + 255) / 256), dim3(256), 0, st, P, pairs, routes, n_routes, ego_xyh, opp,
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
