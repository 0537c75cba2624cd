// igt_kernels_f64.hip -- gfx950 kernels of the float64 entry points (the reference's precision) and their launchers:
//   search_f64_kernel_*   persistent waves, one unit -- 64 candidates of one scenario -- at a time (igt_fast64.h); _cap: small
//                         batches, every unit also keeps its 64 trajectories
//   accel_rows_kernel     ahead of the search (batches above 1024 scenarios): the acceleration rows that can still win
//   emit_f64_kernel       one lane per scenario: the winner re-rolled with the same arithmetic -> x*[7,N+1], u*[2,N]
//   emit_gather_f64_kernel   small batches: the winner's trajectory copied out of the unit that rolled it
//   rollout_all_f64_kernel   debug / parity: every candidate's trajectory, cost and verdict bits
//   search_kernel / emit_kernel / rollout_all_kernel<ExactStepper<double>>   the oracle's operation order (IGT_DEV_FLAGS=1024)
//   search_literal_f64_kernel   the literal north_star mapping, a measurement variant (IGT_DEV_FLAGS=2048)
// Compiled on its own so that the two heavy translation units build in parallel.
#include "igt_device.h"
#include "igt_fast64.h"
#include "igt_launch.h"
#include "igt_kernels_common.h"

namespace igt {

#if IGT_DEV_KERNELS   // the oracle-order kernels (IGT_DEV_FLAGS = 1024)
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


#endif  // IGT_DEV_KERNELS

// ---------------------------------------------------------------------------------------
// float64 path (igt_fast64.h): one candidate per lane, W = C/64 units per scenario
// ---------------------------------------------------------------------------------------
// Generated G x G families with W * 64 == C and W | G: unit p takes G/W STEERING values (all G accelerations), handed out
// from the centre of the range outwards, as the float path does with its 128-candidate slices -- the extreme-steering
// units fail as a whole within a few steps and leave through the early exit.  Otherwise: chunks of 64 in index order
// (igt_kernels_common.h unit_layout / unit_candidate).
// The tracking family's steering is a feedback on the rolled state: its candidates hardly ever fail on |e_y|, they fail on
// what the ACCELERATION profile decides (collision, speed box, terminal set), so its units are cut along the acceleration
// axis instead -- unit p takes G/W consecutive acceleration offsets with all G steering offsets (tools/death_steps_track.py:
// 90.8 % of the wave-steps executed against 94.1 % with steering slices).
//
// Acceleration rows that cannot win.  In the generated families the acceleration sequence of candidate (i, j) depends on the
// row i alone, the speed is v_{k+1} = v_k + dt a_k, and two of the verdicts -- the speed box (mpc.py:316-317) and the terminal
// set on (v_{N-1}, a_{N-1}) (mpc.py:177-180) -- read nothing else: a row that fails one of them is infeasible in all of its
// G columns whatever the steering does.  accel_rows_kernel rolls the G scalar recurrences of a scenario ahead of the search
// (the same expressions as rollout_one, so the same bits) and leaves the rows that survive as a bit mask; the units are then
// made of live rows only:
//   * steering slices (lattice, ramp-hold): the scenario's G R live candidates are numbered column by column from the centre
//     of the steering range outwards and unit p takes numbers 64 p .. 64 p + 63 -- ceil(G R / 64) units instead of W, a column
//     may straddle two of them (benchmark batch: 10.7 of 16 rows live, 3.1 units instead of 4, 77 % of the wave-steps).  Where
//     the unit's steering table would not hold the floor(63 / R) + 2 columns such a unit can touch -- or with
//     IGT_DEV_FLAGS = 4194304 -- a unit holds floor(64 / R) whole columns instead (3.5 units, 83 %);
//   * acceleration-axis units (tracking): unit p takes the live rows of rank p G/W .. (p+1) G/W - 1: ceil(R W / G) units.
// Slices beyond that are empty (the wave moves on), lanes without a candidate idle with "lost" set from the start.
// A mask of all ones is the layout without this (IGT_DEV_FLAGS = 2097152; small batches, packs_live_rows).
//
// one lane per (scenario, acceleration row), 64 / G scenarios per wave: the row's (a_k, v_k) recurrence with the two verdicts that
// read nothing else -- the statements of rollout_one (igt_fast64.h), in its order; what load_scenario reads of the scenario for
// them (a_prev, v_0, the warm start, the refinement centre) is read per lane here
//
// QUEUES (batches whose search queues are sorted, B <= 4096 at 256 candidates): the first eight workgroups of the same launch
// sort one queue each (build_queue, eight trips of 256 threads) while the others roll the rows -- one launch instead of two
// (the queue builder's own launch cost 7 us of a 236 us solve).  The two kinds of workgroup do not talk to each other: the
// queues are sorted without the masks, so a slice beyond a scenario's live rows' units is sorted like any other slice of
// its class and the search wave that takes it finds it empty and moves on.  (Sorting them last needs the masks first: as
// a second launch that is the 7 us; inside this one it was built with tickets and a device-scope fence per workgroup and
// cost 20 us, profiles/r04_ab_fused_prelude.txt.)
constexpr int ROWS_THREADS = 256, ROWS_QB_TRIPS = QB_THREADS * QB_TRIPS / ROWS_THREADS;
template <int CAND, bool QUEUES>
__global__ __launch_bounds__(ROWS_THREADS) void accel_rows_kernel(KP P, int B, const double* __restrict__ x0,
                                                         const double* __restrict__ u_prev, const uint32_t* __restrict__ flags,
                                                         const double* __restrict__ cinf, Centre<double> cpar,
                                                         unsigned long long* __restrict__ row_mask, double* __restrict__ row_rem,
                                                         int W, const double* __restrict__ kparams, unsigned* __restrict__ order,
                                                         int order_stride, unsigned* __restrict__ work_counter) {
    if constexpr (QUEUES) {
        if (blockIdx.x < 8) {
            build_queue<double, ROWS_THREADS, ROWS_QB_TRIPS>(P, B, W, (int)blockIdx.x, x0, kparams, order, order_stride,
                                                             work_counter, nullptr, true);
            return;
        }
    }
    const int lane = threadIdx.x & 63, lg = __ffs(P.G) - 1;                  // G is a power of two (igt_api.hip)
    const int per_wave = 64 >> lg;                                           // scenarios per wave
    const int wave = ((int)blockIdx.x - (QUEUES ? 8 : 0)) * (ROWS_THREADS / 64) + (int)(threadIdx.x >> 6);
    const int b = wave * per_wave + (lane >> lg), i = lane & (P.G - 1);
    bool live = false;
    if (b < B) {
        const double a_prev = u_prev[(size_t)b * 2 + 0], df_prev = u_prev[(size_t)b * 2 + 1];
        const double* ws = (cpar.ws && (flags[b] & 2u)) ? cpar.ws + (size_t)b * 2 * P.N : nullptr;      // load_scenario's S.ws
        double c0 = 0.0, c2 = P.N * P.rate_a;                                                           // ... and S.cpar[0], [2]
        if (cpar.cpar) { c0 = cpar.cpar[(size_t)b * 4 + 0]; c2 = cpar.cpar[(size_t)b * 4 + 2]; }
        double da;
        if (CAND == CAND_LATTICE) da = -P.rate_a + (2 * P.rate_a) * (double)i / (double)(P.G - 1);
        else da = c0 + cand_m(i, P.G, P.refine_it == 0) * c2;
        double a = a_prev, v = x0[(size_t)b * 7 + 5], g = -1.0e300, travel = 0.0;
        unsigned viol = 0;
        bool twin = CAND == CAND_TRACK && i > 0;          // so far the row's accelerations are those of the row below it
        for (int k = 0; k < P.N; ++k) {
            if (CAND == CAND_LATTICE) {
                a = clampd(a + da, P.a_min, P.a_max);
            } else {
                double ba, bdf;
                ramp_base<double>(ws, P.N, k, a_prev, df_prev, ba, bdf);
                if (CAND == CAND_TRACK) {
                    a = track_accel_next(P, k, ba, da, v, a);
                } else {
                    const double ta = clampd(ba + da, P.a_min, P.a_max);
                    a = clampd(a + clampd(ta - a, -P.rate_a, P.rate_a), P.a_min, P.a_max);
                }
            }
            if (CAND == CAND_TRACK) {          // lane - 1: row i - 1 of the same scenario.  Every lane takes part in the exchange
                const double below = __shfl_up(a, 1, 64);      // whatever its own flag says: the lane above it reads what it holds
                twin = twin && (a == below);
            }
            g = fmax(g, fmax(P.v_min - v, v - P.v_max));                             // mpc.py:316-317 (k < N)
            if (k == P.N - 1) viol |= terminal_viol_x8(P, v, a, cinf);               // mpc.py:177-180
            const double vn = fma(P.dt, a, v);
            travel += fmax(fabs(v), fabs(vn));           // sum_k max(|v_k|, |v_k+1|): what the incumbent bound lets the row still gain
            v = vn;
        }
        // (igt_fast64.h BOUND reads it instead of rolling the recurrence ahead once more at the start of every unit: with the speed
        // cap in the targets that preamble had grown to a third of a pruned unit's instructions)
        if (CAND == CAND_TRACK && row_rem) row_rem[(size_t)b * P.G + i] = travel;
        // A tracking row whose N accelerations are, bit for bit, those of the row below it (both ride the speed cap, the envelope
        // or a box limit) rolls the same G candidates again: same trajectories, same costs, and the arg-min's tie rule gives them
        // to the lower index anyway -- the row is left out.  (With the speed cap the rows above the one that just reaches v_max
        // are all such twins; before it they failed the speed box and were left out for that.)
        live = !(g > P.tol) && viol == 0 && !twin;
    }
    const unsigned long long m = __ballot(live);
    if (b < B && i == 0) {
        row_mask[b] = (m >> (lane & ~(P.G - 1))) & (P.G >= 64 ? ~0ull : ((1ull << P.G) - 1ull));
        row_mask[B + b] = cost_key((double)INFINITY);      // the scenario's incumbent of this pass: none yet (igt_fast64.h BOUND)
    }
}

// launch-time bit of KP::dev (never taken from IGT_DEV_FLAGS): the masks of accel_rows_kernel sit behind the partials
constexpr int DEV_LIVE_ROWS = 1 << 30;
constexpr int DEV_NO_BOUND = 8388608;      // IGT_DEV_FLAGS: no incumbent bound in the tracking family's search (A/B runs, bitwise test)
__device__ __forceinline__ const unsigned long long* live_rows_of(const KP& P, int B, int W, const double* part_J) {
    return (P.dev & DEV_LIVE_ROWS) ? reinterpret_cast<const unsigned long long*>(part_J + (size_t)B * W) : nullptr;
}
// the incumbents sit behind the masks: [B] keys, set to "none" by accel_rows_kernel
template <int CAND, bool VALUE>
__device__ __forceinline__ unsigned long long* incumbents_of(const KP& P, int B, int W, double* part_J) {
    return (CAND == CAND_TRACK && !VALUE && (P.dev & DEV_LIVE_ROWS) && !(P.dev & (DEV_NO_BOUND | 262144)))
               ? reinterpret_cast<unsigned long long*>(part_J + (size_t)B * W) + B : nullptr;
}
// launch-time bit of KP::dev: the search pass leaves the unit winners' horizon checkpoints behind the incumbents
// ([B W][CK_RECORD] doubles) for emit_seg_f64_kernel
constexpr int DEV_CKPT = 1 << 29;
constexpr int DEV_NO_SEG_EMIT = 16777216;   // IGT_DEV_FLAGS: emit re-rolls the winner in one piece (A/B runs, bitwise test)
constexpr int CK_RECORD = (f64::CK_PARTS - 1) * f64::CK_FIELDS;
// behind the incumbents: the rows' travel sums [B G] (accel_rows_kernel, tracking family), then the checkpoint records
__device__ __forceinline__ double* row_rems_of(const KP& P, int B, int W, double* part_J) {
    return part_J + (size_t)B * W + 2 * (size_t)B;
}
__device__ __forceinline__ double* checkpoints_of(const KP& P, int B, int W, double* part_J) {
    return (P.dev & DEV_CKPT) ? part_J + (size_t)B * W + 2 * (size_t)B + (size_t)B * P.G : nullptr;
}
// queue items in unit-rank-major order when the tracking family's incumbents are in use and no order table was built
__device__ __forceinline__ bool rank_major_items(const KP& P, int cand, bool value) {
    return cand == CAND_TRACK && !value && (P.dev & DEV_LIVE_ROWS) && !(P.dev & (DEV_NO_BOUND | 262144));
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
template <int CAND, bool HI, bool VALUE, int NRK = 0, bool CAPTURE = false>
__device__ __forceinline__ void search_unit64(const KP& P, int W, int b, int p, const double* __restrict__ x0,
                                              const double* __restrict__ u_prev, const double* __restrict__ kparams,
                                              const uint32_t* __restrict__ flags, const double* __restrict__ obs,
                                              const double* __restrict__ table, const double* __restrict__ cinf,
                                              Centre<double> cpar, double* __restrict__ part_J,
                                              int32_t* __restrict__ part_c, double* __restrict__ rec_sN,
                                              double* __restrict__ rec_vN, double* __restrict__ rec_J,
                                              uint32_t* __restrict__ rec_viol, unsigned* __restrict__ rec_count,
                                              int32_t* __restrict__ rec_b, int2* __restrict__ unit_seg,
                                              const unsigned long long* __restrict__ row_mask = nullptr,
                                              double* __restrict__ traj = nullptr, unsigned long long* inc_all = nullptr,
                                              double* __restrict__ ck_all = nullptr, const double* __restrict__ rem_all = nullptr) {
    const int lane = threadIdx.x & 63;
    unsigned long long* inc = inc_all ? inc_all + b : nullptr;
    const double* rem_rows = (inc_all && rem_all) ? rem_all + (size_t)b * P.G : nullptr;
    const UnitLayout L = unit_layout(P, W, CAND, row_mask ? row_mask[b] : ~0ull);
    if (p >= L.n_units) {             // the scenario's live rows fit fewer units: this slice holds nothing
        if (lane == 0) {
            if (VALUE) unit_seg[b * W + p] = make_int2((b * W + p) * 64, 0);
            else { part_J[b * W + p] = 0.0; part_c[b * W + p] = -1; }
        }
        return;
    }
    Scenario<double> S;
    load_scenario<double>(S, P, b, x0, u_prev, kparams, flags, obs, cpar);
    NullSink sink;
    __shared__ int rank2row[64];
    rows_by_rank(L, lane, rank2row);
    int col = 0, r_first = 0, nj = 1;                        // steering slices: the unit's columns, the lane's among them
    const int c = unit_candidate(P, L, p, lane, rank2row, &col);   // -1: a lane without a candidate (rolls one, counts as lost)
    if (L.kind >= 2) unit_columns(P, L, p, r_first, nj);
    double J, sN, vN;
    unsigned viol;
    double* ck_lds = nullptr;
    if constexpr (CAPTURE) {      // small batches: every lane's trajectory is kept for emit_gather_f64_kernel (all rows, so no
                                  // Cartesian skip; same arithmetic as below, same bits)
        CaptureSink keep{traj + (size_t)(b * W + p) * traj_unit_doubles(P.N) + lane, P.N + 1};
        if (L.kind >= 2 && steer_table_fits(P, W, CAND)) {
            __shared__ double stabc[f64::STAB_MAX_ENTRIES * 3];
            f64::fill_steer_table<CAND>(P, S, nj, r_first, lane, P.lr_ratio, stabc);
            f64::rollout_one<CAND, HI, true, true, CaptureSink, true, true, NRK, true>(P, S, c, table, cinf, keep, J, viol, sN, vN,
                                                                                       stabc + col * 3, nj * 3);
            __syncthreads();
        } else {
            f64::rollout_one<CAND, HI, true, true, CaptureSink, true, false, NRK, true>(P, S, c, table, cinf, keep, J, viol, sN, vN);
        }
    } else {
    // steering slices of the families with state-independent steering: the slice's G/W steering columns are laid out in
    // LDS once per unit instead of being recomputed by each of their 64 W/G lanes at every step (igt_fast64.h)
    // 70 % of the benchmark's scenarios: the other vehicle is out of reach over the whole horizon (or filter_preds moved it
    // away), so the unit rolls without the Cartesian rows -- a sixth of the control step's instructions
    const bool far = !(P.dev & 65536) && obstacles_out_of_reach<double>(P, S, lane);
    // horizon checkpoints of every lane in LDS (igt_fast64.h SEGMODE 1); the unit winner's go to HBM below
    constexpr int SM = VALUE ? 0 : 1;
    constexpr int CKF = CAND == CAND_TRACK ? 8 : 5;
    __shared__ double ckl[VALUE ? 1 : (f64::CK_PARTS - 1) * CKF * 64];
    ck_lds = (!VALUE && ck_all) ? ckl + lane : nullptr;
    if (L.kind >= 2 && steer_table_fits(P, W, CAND)) {
        __shared__ double stab[f64::STAB_MAX_ENTRIES * 3];
        f64::fill_steer_table<CAND>(P, S, nj, r_first, lane, P.lr_ratio, stab);
        if (far)
            f64::rollout_one<CAND, HI, true, true, NullSink, true, true, NRK, false, SM>(P, S, c, table, cinf, sink, J, viol, sN, vN,
                                                                                         stab + col * 3, nj * 3, nullptr, ck_lds);
        else
            f64::rollout_one<CAND, HI, true, true, NullSink, true, true, NRK, true, SM>(P, S, c, table, cinf, sink, J, viol, sN, vN,
                                                                                        stab + col * 3, nj * 3, nullptr, ck_lds);
        __syncthreads();                                  // the next unit of this wave rewrites the table
    } else if (far) {
        f64::rollout_one<CAND, HI, true, true, NullSink, true, false, NRK, false, SM>(P, S, c, table, cinf, sink, J, viol, sN, vN, nullptr, 0, inc, ck_lds, 0, 0, rem_rows);
    } else {
        f64::rollout_one<CAND, HI, true, true, NullSink, true, false, NRK, true, SM>(P, S, c, table, cinf, sink, J, viol, sN, vN, nullptr, 0, inc, ck_lds, 0, 0, rem_rows);
    }
    }
    if (VALUE) {   // terminal value network (mpc.py:369): append the feasible candidates for value_mfma_f64_kernel; the
                   // unit's entries are contiguous, unit_seg remembers where (unit_reduce_kernel picks the unit's best)
        // The unit's entries go to the unit's own 64 slots: no counter is shared between the units (a returning atomicAdd on
        // one address retires every 11.4 ns on this part -- 262 144 units at B = 65 536 would queue for 3 ms); the dense list
        // the network runs over is built afterwards from the units' counts (value_select_kernel).
        const bool ok = viol == 0 && finite_d(J);
        const unsigned long long m = __ballot(ok);
        const unsigned n = __popcll(m);
        const unsigned base = (unsigned)(b * W + p) * 64u;
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
    if (ck_lds && bestC >= 0 && c == bestC) {          // the unit winner's lane: its checkpoints -> HBM, record [q][field]
        constexpr int CKF2 = CAND == CAND_TRACK ? 8 : 5;
        double* g = ck_all + (size_t)(b * W + p) * CK_RECORD;
#pragma unroll
        for (int q = 0; q < f64::CK_PARTS - 1; ++q)
#pragma unroll
            for (int f = 0; f < CKF2; ++f) g[q * f64::CK_FIELDS + f] = ck_lds[(size_t)(q * CKF2 + f) * 64];
    }
    if (lane == 0) {
        part_J[b * W + p] = bestJ; part_c[b * W + p] = bestC;
        // the unit's best feasible cost is the scenario's incumbent from now on (device-scope atomic: the later units of the
        // scenario may run on another XCD)
        if (inc && bestC >= 0) atomicMin(inc, cost_key(bestJ));
    }
}

// persistent waves on the per-XCD queues (search_waves), 2 or 3 per SIMD like the float kernels.  NRK = 4: the build for the
// reference's discretisation (num_rk4_steps = 4, evaluate.py:109), NRK = 0: any n_rk4 (same arithmetic, same bits)
template <int CAND, bool HI, bool VALUE, int NRK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void search_f64_kernel_o2(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE, NRK>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg, live_rows_of(P, B, W, part_J), nullptr,
                                       incumbents_of<CAND, VALUE>(P, B, W, part_J), checkpoints_of(P, B, W, part_J), row_rems_of(P, B, W, part_J));
    }, rank_major_items(P, CAND, VALUE));
}
template <int CAND, bool HI, bool VALUE, int NRK>      // held to 256 registers for the tracking family (see search_fast_kernel_o2w)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void search_f64_kernel_o2w(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE, NRK>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg, live_rows_of(P, B, W, part_J), nullptr,
                                       incumbents_of<CAND, VALUE>(P, B, W, part_J), checkpoints_of(P, B, W, part_J), row_rems_of(P, B, W, part_J));
    }, rank_major_items(P, CAND, VALUE));
}
// small batches (captures_trajectories): the same search, every unit also leaves its 64 trajectories in `traj`
template <int CAND, bool VALUE>
__global__ __launch_bounds__(64) void search_f64_kernel_cap(IGT_SEARCH64_ARGS, double* __restrict__ traj) {
    if (queues == 0) {      // every unit has a wave of its own (search_is_static): no queues, no counters
        search_unit64<CAND, false, VALUE, 4, true>(P, W, (int)(blockIdx.x / (unsigned)W), (int)(blockIdx.x % (unsigned)W), x0, u_prev,
                                                   kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN, rec_vN, rec_J,
                                                   rec_viol, rec_count, rec_b, unit_seg, nullptr, traj);
        return;
    }
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, false, VALUE, 4, true>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c,
                                                   rec_sN, rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg, nullptr, traj);
    });
}
// one 64-thread block per scenario: final arg-min over the W partials, then the winner's trajectory copied out of the unit
// that rolled it.  Which lane that was is asked of unit_candidate itself, so the two cannot drift apart.
template <int CAND>
__global__ __launch_bounds__(64) void emit_gather_f64_kernel(KP P, int B, int W, const double* __restrict__ part_J,
                                                             const int32_t* __restrict__ part_c,
                                                             const double* __restrict__ traj, double* __restrict__ cost_out,
                                                             int32_t* __restrict__ argmin_out, int32_t* __restrict__ status_out,
                                                             double* __restrict__ x_out, double* __restrict__ u_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double bestJ = 0.0;
    int c = -1, pw = 0;
    for (int w = 0; w < W; ++w) {      // (J, c) lexicographic, as emit_f64_kernel
        const int cw = part_c[(size_t)b * W + w];
        const double Jw = part_J[(size_t)b * W + w];
        if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; pw = w; }
    }
    if (lane == 0) {
        cost_out[b] = c >= 0 ? bestJ : (double)INFINITY;
        argmin_out[b] = c;
        status_out[b] = c >= 0 ? 0 : 1;
    }
    const int N1 = P.N + 1, nx = 7 * N1, nu = 2 * P.N;
    double* xo = x_out + (size_t)b * nx;
    double* uo = u_out + (size_t)b * nu;
    const UnitLayout L = unit_layout(P, W, CAND, ~0ull);      // batches that keep trajectories run the full layout
    const unsigned long long holder = __ballot(c >= 0 && unit_candidate(P, L, pw, lane, nullptr) == c);
    if (c < 0 || holder == 0ull) {     // is_opt False (mpc.py:402-406): no trajectory
        for (int i = lane; i < nx; i += 64) xo[i] = (double)NAN;
        for (int i = lane; i < nu; i += 64) uo[i] = (double)NAN;
        return;
    }
    const double* src = traj + (size_t)(b * W + pw) * traj_unit_doubles(P.N) + (__ffsll((long long)holder) - 1);
    for (int i = lane; i < nx; i += 64) xo[i] = src[(size_t)i * 64];
    for (int i = lane; i < nu; i += 64) uo[i] = src[(size_t)((7 + i / P.N) * N1 + i % P.N) * 64];
}
#if IGT_DEV_KERNELS
template <int CAND, bool HI, bool VALUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void search_f64_kernel_o3(IGT_SEARCH64_ARGS) {
    search_waves(P, B, W, queues, work_counter, order, order_stride, [&](int b, int p) {
        search_unit64<CAND, HI, VALUE>(P, W, b, p, x0, u_prev, kparams, flags, obs, table, cinf, cpar, part_J, part_c, rec_sN,
                                       rec_vN, rec_J, rec_viol, rec_count, rec_b, unit_seg, live_rows_of(P, B, W, part_J));
    });
}
#endif

// one lane per scenario: final arg-min over the W partials, then the winner re-rolled with the same arithmetic
// (general sub-step variant: the lanes of the wave belong to different scenarios) -> x*[7,N+1], u*[2,N]
template <int CAND, bool HI, int NRK>
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
    f64::rollout_one<CAND, HI, false, false, StoreSink<double>, false, false, NRK>(P, S, c, table, cinf, sink, J, viol, sN, vN);
}

// ---- emit in pieces (batches that do not keep trajectories; progress cost) ----
// One roll-out is a serial chain, and a wave issues a float64 instruction every four cycles however few of its lanes are
// active: emit_f64_kernel's 64 waves at B = 4096 take one roll-out of time, 51 us.  Here a workgroup of four waves owns S
// scenarios (lane = scenario): wave w rolls steps [w N / 4, (w + 1) N / 4) of each winner's Frenet rows from the checkpoint the
// search pass left (igt_fast64.h SEGMODE 2) into LDS -- states, controls and (sin, cos)(beta_k) -- , then the first wave rolls the
// Cartesian rows x, y, psi from those controls (cartesian_rows), and the workgroup writes its scenarios' x*[7, N+1] and u*[2, N]
// out of LDS as they lie in HBM: contiguous, every store of a wave a full line (emit_f64_kernel: one lane per scenario,
// 64 lines touched per store instruction, 2.6x the bytes at B = 65 536).  The same statements on the same numbers as the
// roll-out in one piece: bit-identical (tests: against IGT_DEV_FLAGS = 16777216 and against rollout-all).
struct SegSink {
    static constexpr bool kKeepsStates = true;
    double* X;      // this scenario's [7][N+1] in LDS
    double* U;      // [2][N]
    double* SB;     // [N] sin beta_k
    double* CB;     // [N] cos beta_k
    int N1;
    __device__ __forceinline__ void ctrl(int, int k, double a, double df) { U[k] = a; U[N1 - 1 + k] = df; }
    __device__ __forceinline__ void slip(int, int k, double sb, double cb) { SB[k] = sb; CB[k] = cb; }
    __device__ __forceinline__ void state(int, int k, const double (&st)[7]) {
        X[2 * N1 + k] = st[2]; X[3 * N1 + k] = st[3]; X[4 * N1 + k] = st[4]; X[5 * N1 + k] = st[5];
        if (k == 0) { X[0] = st[0]; X[N1] = st[1]; X[6 * N1] = st[6]; }
    }
};
__host__ __device__ inline int seg_doubles_per_scenario(int N) { return 7 * (N + 1) + 4 * N; }

struct LdsControls {         // tracking family: the controls of step k as the Frenet pieces left them in LDS
    const double* A; const double* SB; const double* CB;
    __device__ __forceinline__ void operator()(int k, double& a, double& sb, double& cb) const { a = A[k]; sb = SB[k]; cb = CB[k]; }
};
constexpr int SEG_THREADS = 320;      // four waves for the pieces + one for the Cartesian rows (idle in the tracking family's first phase)

template <int CAND, bool HI, int NRK>
__global__ __launch_bounds__(SEG_THREADS) void emit_seg_f64_kernel(KP P, int B, int W, int S, const double* __restrict__ x0,
                                                           const double* __restrict__ u_prev, const double* __restrict__ kparams,
                                                           const uint32_t* __restrict__ flags, const double* __restrict__ obs,
                                                           const double* __restrict__ table, const double* __restrict__ cinf,
                                                           Centre<double> cpar, const double* __restrict__ part_J,
                                                           const int32_t* __restrict__ part_c, double* __restrict__ ck_all,
                                                           double* __restrict__ cost_out, int32_t* __restrict__ argmin_out,
                                                           int32_t* __restrict__ status_out, double* __restrict__ x_out,
                                                           double* __restrict__ u_out) {
    extern __shared__ double seg_lds[];
    __shared__ int win[64];
    const int N1 = P.N + 1, per = seg_doubles_per_scenario(P.N);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b0 = blockIdx.x * S, b = b0 + lane;
    const bool active = lane < S && b < B;
    double bestJ = 0.0;
    int c = -1, pw = 0;
    if (active) {
        for (int w = 0; w < W; ++w) {      // (J, c) lexicographic: ties -> lowest candidate index whatever the slice order
            const int cw = part_c[(size_t)b * W + w];
            const double Jw = part_J[(size_t)b * W + w];
            if (cw >= 0 && (c < 0 || Jw < bestJ || (Jw == bestJ && cw < c))) { bestJ = Jw; c = cw; pw = w; }
        }
        if (wave == 0) {
            cost_out[b] = c >= 0 ? bestJ : (double)INFINITY;
            argmin_out[b] = c;
            status_out[b] = c >= 0 ? 0 : 1;
            win[lane] = c;
        }
    }
    double* blk = seg_lds + (size_t)(lane < S ? lane : 0) * per;
    Scenario<double> Sc;
    if (active && c >= 0) {
        load_scenario<double>(Sc, P, b, x0, u_prev, kparams, flags, obs, cpar);
        if (wave < f64::CK_PARTS) {
            SegSink sink{blk, blk + 7 * N1, blk + 7 * N1 + 2 * P.N, blk + 7 * N1 + 3 * P.N, N1};
            const int k0 = f64::ckpt_step(P.N, wave), k1 = f64::ckpt_step(P.N, wave + 1);
            double* ck = wave > 0 ? ck_all + ((size_t)b * W + pw) * CK_RECORD + (size_t)(wave - 1) * f64::CK_FIELDS : nullptr;
            double J, sN, vN;
            unsigned viol;
            f64::rollout_one<CAND, HI, false, false, SegSink, false, false, NRK, false, 2>(P, Sc, c, table, cinf, sink, J, viol, sN, vN,
                                                                                           nullptr, 0, nullptr, ck, k0, k1);
        } else if (CAND != CAND_TRACK) {      // beside the pieces: the Cartesian rows, the controls generated on the spot
            f64::StepControls<CAND> ctl(P, Sc, c, table);
            f64::cartesian_rows<HI, NRK>(P, Sc.x0[0], Sc.x0[1], Sc.x0[6], Sc.x0[5], ctl, blk, blk + N1, blk + 6 * N1, 1);
        }
    }
    __syncthreads();
    if (CAND == CAND_TRACK) {                  // behind the pieces: the steering is theirs to decide
        if (wave == f64::CK_PARTS && active && c >= 0) {
            LdsControls ctl{blk + 7 * N1, blk + 7 * N1 + 2 * P.N, blk + 7 * N1 + 3 * P.N};
            f64::cartesian_rows<HI, NRK>(P, Sc.x0[0], Sc.x0[1], Sc.x0[6], Sc.x0[5], ctl, blk, blk + N1, blk + 6 * N1, 1);
        }
        __syncthreads();
    }
    // the workgroup's scenarios as they lie in HBM (is_opt False, mpc.py:402-406: NaN)
    const int nS = B - b0 < S ? B - b0 : S, nx = 7 * N1, nu = 2 * P.N;
    double* xo = x_out + (size_t)b0 * nx;
    double* uo = u_out + (size_t)b0 * nu;
    for (int i = threadIdx.x; i < nS * nx; i += SEG_THREADS) {
        const int sc = i / nx, r = i - sc * nx;
        xo[i] = win[sc] >= 0 ? seg_lds[(size_t)sc * per + r] : (double)NAN;
    }
    for (int i = threadIdx.x; i < nS * nu; i += SEG_THREADS) {
        const int sc = i / nu, r = i - sc * nu;
        uo[i] = win[sc] >= 0 ? seg_lds[(size_t)sc * per + nx + r] : (double)NAN;
    }
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

#if IGT_DEV_KERNELS
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
    __device__ __forceinline__ void slip(int, int, double, double) {}
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

#endif  // IGT_DEV_KERNELS

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
#if IGT_DEV_KERNELS
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

#endif
bool search_builds_queues(const KP& P, int B, const SolveArgs<double>& A) {
    const int W = P.C / 64;
    return A.queue_order && W <= 256 && ((B + 7) / 8) * W <= QB_THREADS * QB_TRIPS && !(P.dev & 16) && !(P.dev & (1024 | 2048));
}

// Small batches keep the trajectories of the search pass (CaptureSink) when the kernel built for the reference's
// discretisation runs; IGT_DEV_FLAGS = 524288 switches it off for A/B runs.
static bool captures_trajectories(const KP& P, const SolveArgs<double>& A) {
    return A.traj && !P.hi_order && P.n_rk4 == 4 && !(P.dev & (32 | 1024 | 2048 | 524288));
}

// ... and such a batch has no more units than the chip has SIMDs (the kernel that keeps trajectories holds one wave per
// SIMD), so every unit is given its own wave: the queue builder (6.6 us and a launch) would only order units that all start
// at once.  Measured, search + emit, tracking family: 67 us against 84 at B = 1, 91 against 101 at B = 32.
// IGT_DEV_FLAGS = 1048576 keeps the queues for A/B runs; the unit trace (256) needs them.
bool search_is_static(const KP& P, int B, const SolveArgs<double>& A) {
    return captures_trajectories(P, A) && (size_t)B * (P.C / 64) <= (size_t)A.n_cu * 4 && !(P.dev & (256 | 1048576));
}

// Whether emit rolls the winner in four pieces from the search pass's checkpoints (emit_seg_f64_kernel): progress cost (with the
// value network the winner is only known after the network has run), batches that do not keep trajectories, horizons of at
// least 8 steps, workspace sized for the records (A.ck_ok).  The search launcher and the emit launcher both ask this.
// scenarios per workgroup: a full wave's 64 when their staging fits a compute unit's LDS (116 KB at N = 20; 160 KB per CU,
// the kernel asks for more than the default 64 KB: seg_lds_opt_in), else the largest power of two that does
constexpr size_t SEG_LDS_MAX = 150 * 1024;
static int seg_scenarios_per_block(const KP& P) {
    int S = 64;
    while (S > 0 && (size_t)S * seg_doubles_per_scenario(P.N) * 8 > SEG_LDS_MAX) S >>= 1;
    return S;
}
template <class K>
static hipError_t seg_lds_opt_in(K kernel) {      // once per instantiation and device (a function attribute)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEG_LDS_MAX);
}
// Not when the caller overlaps solves (igt_set_concurrency >= 3): there the emit pass of one solve runs under the search pass of
// another, its latency is hidden, and what matters is that it gets on the chip between two persistent search kernels -- one
// 64-thread wave without LDS does, a 320-thread workgroup with 116 KB of LDS waits for a compute unit to drain (measured on one
// box, three runs each: 20.3 against 20.9 M solves/s with four solves in flight).
static bool emits_in_pieces(const KP& P, const SolveArgs<double>& A) {
    return A.ck_ok && P.cost_mode == 0 && P.N >= 8 && !captures_trajectories(P, A) && seg_scenarios_per_block(P) >= 4 &&
           A.waves_per_simd != 1 && !(P.dev & (32 | 1024 | 2048 | DEV_NO_SEG_EMIT));
}

// Whether the search runs on units made of live acceleration rows only (accel_rows_kernel; unit_layout).  Only for batches of
// more than two rounds of units (B > 1024 at 256 candidates): below that the solve lasts as long as its longest unit, fewer
// units do not shorten it, and the extra launch costs 8 us (a closed loop of 512 problems per step: 0.30 against 0.28 ms).
static bool packs_live_rows(const KP& P, int B, const SolveArgs<double>& A) {
    const int W = P.C / 64;
    return A.row_mask && P.cand_mode != CAND_TABLE && P.G <= 64 && P.G * P.G == P.C && W * 64 == P.C && P.G % W == 0 &&
           !(P.dev & (1 | 1024 | 2048 | 2097152)) && !captures_trajectories(P, A) && (size_t)B * W > (size_t)A.n_cu * 16;
}

// float64 search: persistent waves on the per-XCD queues, one 64-candidate unit at a time
template <int CAND, bool HI, bool VALUE>
static hipError_t launch_search64(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    const int W = P.C / 64;
#if IGT_DEV_KERNELS
    if ((P.dev & 2048) && !VALUE && P.N <= LIT_MAX_N) {       // measurement variant: the literal wave-per-trajectory mapping
        hipLaunchKernelGGL((search_literal_f64_kernel<CAND, HI>), dim3(B), dim3(64 * LIT_WAVES), 0, st, P, B, W, A.x0, A.u_prev,
                           A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c);
        return hipGetLastError();
    }
#endif
    const size_t total = (size_t)B * W;
    // 2 waves per SIMD (232 VGPRs, no spill); the 3-per-SIMD build spills 244 B/lane and is 3-8 % behind at every batch
    // size (IGT_DEV_FLAGS = 32 selects it for A/B runs)
    const bool o3 = IGT_DEV_KERNELS && (P.dev & 32) != 0;
    const size_t slots = (size_t)A.n_cu * 4 * (o3 ? 3 : (A.waves_per_simd == 1 ? 1 : 2));
    const size_t grid = total < slots ? total : slots;
    const unsigned* order = nullptr;
    const int order_stride = ((B + 7) / 8) * W;
    if (!HI && search_is_static(P, B, A)) {
        hipLaunchKernelGGL((search_f64_kernel_cap<CAND, VALUE>), dim3(total), dim3(64), 0, st, P, B, W, 0, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg, A.traj);
        return hipGetLastError();
    }
    // the live-row masks live behind the partials (part_J[B W ..]): no further kernel argument -- the search kernels spill
    // scalar registers as it is, and every one more shows up as v_readlane in the control-step loop
    KP Pr = P;
    if (!VALUE && emits_in_pieces(P, A)) Pr.dev |= DEV_CKPT;
    const unsigned long long* rows = nullptr;
    bool queues_built = false;
    if constexpr (CAND != CAND_TABLE) {
        if (packs_live_rows(P, B, A)) {
            rows = reinterpret_cast<const unsigned long long*>(A.part_J + (size_t)B * W);
            const int per_block = (ROWS_THREADS / 64) * (64 / P.G);    // scenarios per workgroup
            const int n_groups = (B + per_block - 1) / per_block;
            // small batches: the same launch sorts the queues (longest units first)
            queues_built = search_builds_queues(P, B, A) && !(P.dev & 33554432);
#define IGT_LAUNCH_ROWS(QUEUES_)                                                                                              \
            hipLaunchKernelGGL((accel_rows_kernel<CAND, QUEUES_>), dim3(n_groups + (QUEUES_ ? 8 : 0)), dim3(ROWS_THREADS), 0, st, P, \
                               B, A.x0, A.u_prev, A.flags, A.cinf, A.centre(), const_cast<unsigned long long*>(rows),        \
                               A.part_J + (size_t)B * W + 2 * (size_t)B, W, A.kparams, A.queue_order, order_stride,           \
                               A.work_counter)
            if (queues_built) { IGT_LAUNCH_ROWS(true); order = A.queue_order; } else IGT_LAUNCH_ROWS(false);
#undef IGT_LAUNCH_ROWS
            Pr.dev |= DEV_LIVE_ROWS;
        }
    }
    if (!queues_built && search_builds_queues(P, B, A)) {    // small batches: longest units first
        hipLaunchKernelGGL(build_queues_kernel<double>, dim3(8), dim3(QB_THREADS), 0, st, P, B, W, A.x0, A.kparams,
                           A.queue_order, order_stride, A.work_counter, rows);
        order = A.queue_order;
    }
    if (!HI && captures_trajectories(P, A)) {
        hipLaunchKernelGGL((search_f64_kernel_cap<CAND, VALUE>), dim3(grid), dim3(64), 0, st, P, B, W, 8, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg, A.traj);
        return hipGetLastError();
    }
#if IGT_DEV_KERNELS
    if (o3)
        hipLaunchKernelGGL((search_f64_kernel_o3<CAND, HI, VALUE>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter, order,
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg);
    else
#endif
    {
        // the reference's discretisation (4 sub-steps, short polynomials) has its own build of the kernel
        constexpr int NRK4 = HI ? 0 : 4;
        const bool rk4 = NRK4 == 4 && P.n_rk4 == 4;
#define IGT_LAUNCH_S64(KERNEL, NRK_)                                                                                          \
        hipLaunchKernelGGL((KERNEL<CAND, HI, VALUE, NRK_>), dim3(grid), dim3(64), 0, st, Pr, B, W, 8, A.work_counter, order,    \
                           order_stride, A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, \
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol, A.rec_count, A.rec_b, A.unit_seg)
        if constexpr (CAND == CAND_TRACK) {
            if (rk4) IGT_LAUNCH_S64(search_f64_kernel_o2w, NRK4); else IGT_LAUNCH_S64(search_f64_kernel_o2w, 0);
        } else {
            if (rk4) IGT_LAUNCH_S64(search_f64_kernel_o2, NRK4); else IGT_LAUNCH_S64(search_f64_kernel_o2, 0);
        }
#undef IGT_LAUNCH_S64
    }
    return hipGetLastError();
}
template <bool VALUE>
static hipError_t dispatch_search64(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
#if IGT_DEV_KERNELS
    if (P.dev & 1024) return launch_search_exact<VALUE>(P, B, A, st);     // developer switch: oracle-order kernels
#endif
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
hipError_t launch_search_records<double>(const KP& P, int B, const SolveArgs<double>& A, hipStream_t st) {
    return dispatch_search64<true>(P, B, A, st);
}

template <int CAND, bool HI>
static hipError_t launch_emit64(const KP& P, int B, int W, const SolveArgs<double>& A, hipStream_t st) {
    constexpr int NRK4 = HI ? 0 : 4;
    if (!HI && captures_trajectories(P, A)) {
        hipLaunchKernelGGL((emit_gather_f64_kernel<CAND>), dim3(B), dim3(64), 0, st, P, B, W, A.part_J, A.part_c, A.traj, A.cost_out,
                           A.argmin_out, A.status_out, A.x_out, A.u_out);
        return hipGetLastError();
    }
    if (emits_in_pieces(P, A)) {
        const int S = seg_scenarios_per_block(P);
        const size_t lds = (size_t)S * seg_doubles_per_scenario(P.N) * 8;
        double* ck = A.part_J + (size_t)B * W + 2 * (size_t)B + (size_t)B * P.G;
        if (NRK4 == 4 && P.n_rk4 == 4)
            hipLaunchKernelGGL((emit_seg_f64_kernel<CAND, HI, NRK4>), dim3((B + S - 1) / S), dim3(SEG_THREADS), lds, st, P, B, W, S, A.x0, A.u_prev,
                               A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, ck, A.cost_out, A.argmin_out,
                               A.status_out, A.x_out, A.u_out);
        else
            hipLaunchKernelGGL((emit_seg_f64_kernel<CAND, HI, 0>), dim3((B + S - 1) / S), dim3(SEG_THREADS), lds, st, P, B, W, S, A.x0, A.u_prev,
                               A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, ck, A.cost_out, A.argmin_out,
                               A.status_out, A.x_out, A.u_out);
        return hipGetLastError();
    }
    if (NRK4 == 4 && P.n_rk4 == 4)
        hipLaunchKernelGGL((emit_f64_kernel<CAND, HI, NRK4>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, W, A.x0, A.u_prev, A.kparams,
                           A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.cost_out, A.argmin_out, A.status_out,
                           A.x_out, A.u_out);
    else
        hipLaunchKernelGGL((emit_f64_kernel<CAND, HI, 0>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, W, A.x0, A.u_prev, A.kparams,
                           A.flags, A.obs, A.table, A.cinf, A.centre(), A.part_J, A.part_c, A.cost_out, A.argmin_out, A.status_out,
                           A.x_out, A.u_out);
    return hipGetLastError();
}
// dynamic LDS above 64 KB has to be asked for, per kernel and device: once per handle at igt_create -- not on the launch path,
// which must stay a pure sequence of stream operations (stream capture)
template <int CAND>
static hipError_t seg_opt_in_family() {
    hipError_t e = seg_lds_opt_in(emit_seg_f64_kernel<CAND, false, 4>);
    if (e == hipSuccess) e = seg_lds_opt_in(emit_seg_f64_kernel<CAND, false, 0>);
    if (e == hipSuccess) e = seg_lds_opt_in(emit_seg_f64_kernel<CAND, true, 0>);
    return e;
}
hipError_t prepare_emit_kernels() {
    hipError_t e = seg_opt_in_family<CAND_LATTICE>();
    if (e == hipSuccess) e = seg_opt_in_family<CAND_TABLE>();
    if (e == hipSuccess) e = seg_opt_in_family<CAND_RAMP_HOLD>();
    if (e == hipSuccess) e = seg_opt_in_family<CAND_TRACK>();
    return e;
}

template <>
hipError_t launch_emit<double>(const KP& P, int B, int W, const SolveArgs<double>& A, hipStream_t st) {
#if IGT_DEV_KERNELS
    if (P.dev & 1024) {     // developer switch: oracle-order kernels (argmin_out is already final there)
        hipLaunchKernelGGL((emit_kernel<ExactStepper<double>, double>), dim3((B + 63) / 64), dim3(64), 0, st, P, B, A.x0,
                           A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), A.argmin_out, A.x_out, A.u_out);
        return hipGetLastError();
    }
#endif
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
#if IGT_DEV_KERNELS
    if (P.dev & 1024) {     // developer switch: oracle-order kernels
        hipLaunchKernelGGL((rollout_all_kernel<ExactStepper<double>, double>), dim3((B + 3) / 4), dim3(256), 0, st, P, B,
                           A.x0, A.u_prev, A.kparams, A.flags, A.obs, A.table, A.cinf, A.centre(), X_all, U_all, cost_all, viol_all,
                           A.rec_sN, A.rec_vN, A.rec_J, A.rec_viol);
        return hipGetLastError();
    }
#endif
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

}  // namespace igt
