// igt_kernels_common.h -- device code shared by the kernel translation units (igt_kernels.hip: float path, value network,
// forecast, model steps; igt_kernels_f64.hip: the float64 search / emit / rollout-all kernels): scenario loading, the
// trajectory sink, the per-XCD work queues and the persistent-wave loop.
#pragma once
#include "igt_device.h"
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

template <typename T>
struct StoreSink {
    static constexpr bool kKeepsStates = true;
    T* x;   // [7, N+1] of this scenario/candidate (may be null)
    T* u;   // [2, N]
    int N;
    __device__ __forceinline__ void ctrl(int, int k, double a, double df) {
        if (u) { u[k] = (T)a; u[N + k] = (T)df; }
    }
    __device__ __forceinline__ void slip(int, int, double, double) {}
    __device__ __forceinline__ void state(int, int k, const double (&st)[7]) {
        if (x) {
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i * (N + 1) + k] = (T)st[i];
        }
    }
};

// Small batches: the search pass itself keeps every candidate's trajectory (unit-major, then field, step, lane: each store of
// a wave is one 512-byte line), and emit copies the winner's instead of rolling it again -- the second roll is a serial chain
// of N steps on one lane, 60 us of a 150 us solve at N = 20.  9 (N+1) 64 doubles per unit: 97 KB at N = 20, 400 MB of
// stores at B = 1024 -- which is why this is for batches with at most one unit per SIMD only (igt_api.hip solve_impl).
struct CaptureSink {
    static constexpr bool kKeepsStates = true;
    double* base;   // this unit's block + lane
    int N1;         // N + 1
    __device__ __forceinline__ void ctrl(int, int k, double a, double df) {
        base[(size_t)(7 * N1 + k) * 64] = a;
        base[(size_t)(8 * N1 + k) * 64] = df;
    }
    __device__ __forceinline__ void slip(int, int, double, double) {}
    __device__ __forceinline__ void state(int, int k, const double (&st)[7]) {
#pragma unroll
        for (int i = 0; i < 7; ++i) base[(size_t)(i * N1 + k) * 64] = st[i];
    }
};

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
// ---- units of the float64 search (igt_kernels_f64.hip: "Acceleration rows that cannot win") ----
struct UnitLayout {
    int kind;                       // 0: 64 candidates in index order, 1: acceleration-axis units, 2: whole steering columns per
                                    // unit, 3: 64 live candidates per unit in column-major order (columns may straddle units)
    int per;                        // kind 1: rows per unit, kind 2: steering columns per unit
    int n_units, R;
    unsigned long long mask;        // live acceleration rows
};
// x / d for 0 <= x < 4096, 1 <= d <= 64 through the float reciprocal: (x + 1/2) / d is at least 1/128 away from an integer, the
// float error is below 1e-3 of that -- exact, and a tenth of the instructions of an integer division by a run-time value
__device__ __forceinline__ int small_div(int x, int d) {
    return (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}
// whether the steering table of a slice fits LDS with the layout's own G / W columns (then the live-row layouts keep within it)
__device__ __forceinline__ bool steer_table_fits(const KP& P, int W, int cand) {
    return (cand == CAND_LATTICE || cand == CAND_RAMP_HOLD) && (P.G / W) * P.N <= STEER_TABLE_MAX_ENTRIES && !(P.dev & 4);
}
// UNIFORM: the caller's lanes all ask about the same scenario (a search / emit wave): wave-uniform results are moved to scalar
// registers.  The queue builder asks per lane about DIFFERENT scenarios and must not (round 4: it did, so every lane of a builder
// wave took the unit count of lane 0's scenario, and a scenario with more units than that had its last unit sorted with the
// empty slices, at the very end of the queue -- 576 arc units of 30-60 us started after 100 us at B = 4096).
template <bool UNIFORM = true>
__device__ __forceinline__ UnitLayout unit_layout(const KP& P, int W, int cand, unsigned long long mask) {
    UnitLayout L;
    const bool slices = cand != CAND_TABLE && P.G * P.G == P.C && W * 64 == P.C && P.G % W == 0 && !(P.dev & 1);
    if (!slices) { L.kind = 0; L.per = 64; L.n_units = W; L.R = P.G; L.mask = ~0ull; return L; }
    L.mask = mask & (P.G >= 64 ? ~0ull : ((1ull << P.G) - 1ull));
    L.R = __popcll(L.mask);
    const int nj = 64 / P.G;                                 // = G / W (C = G^2 = 64 W; G is a power of two, igt_api.hip)
    if (cand == CAND_TRACK && !(P.dev & 262144)) {
        L.kind = 1; L.per = nj;
        L.n_units = (L.R + nj - 1) / nj;                     // nj is a power of two as well
        return L;
    }
    const bool table = steer_table_fits(P, W, cand);
    // 64 live candidates per unit, column by column from the centre outwards: a unit touches at most floor(63 / R) + 2 columns
    if (L.R > 0 && !(P.dev & 4194304) && (!table || (small_div(63, L.R) + 2) * P.N <= STEER_TABLE_MAX_ENTRIES)) {
        L.kind = 3; L.per = 0;
        L.n_units = (P.G * L.R + 63) >> 6;
        if (UNIFORM) L.n_units = __builtin_amdgcn_readfirstlane(L.n_units);
        return L;
    }
    L.kind = 2;
    int per = L.R > 0 ? small_div(64, L.R) : P.G;
    if (per > P.G) per = P.G;
    if (table) { const int cap = small_div(STEER_TABLE_MAX_ENTRIES, P.N); if (per > cap) per = cap; }
    // the float detour leaves these in vector registers although they are the same on every lane: back to scalars, or the
    // stride of the steering table is recomputed on the vector ALU at every control step
    L.per = per;
    L.n_units = L.R > 0 ? small_div(P.G + per - 1, per) : 0;
    if (UNIFORM) { L.per = __builtin_amdgcn_readfirstlane(L.per); L.n_units = __builtin_amdgcn_readfirstlane(L.n_units); }
    return L;
}
// the steering columns (ranks, centre outwards) unit p touches: first and how many
__device__ __forceinline__ void unit_columns(const KP& P, const UnitLayout& L, int p, int& r_first, int& ncol) {
    if (L.kind == 3) {
        r_first = __builtin_amdgcn_readfirstlane(small_div(64 * p, L.R));
        int r_last = __builtin_amdgcn_readfirstlane(small_div(64 * p + 63, L.R));
        if (r_last > P.G - 1) r_last = P.G - 1;
        ncol = r_last - r_first + 1;
    } else {
        r_first = p * L.per;
        ncol = P.G - r_first < L.per ? P.G - r_first : L.per;
    }
}
// rank -> row of the live acceleration rows, laid out in LDS by the unit's wave (one workgroup = one wave): lane i, if row i is
// live, stores i at the number of live rows below it
__device__ __forceinline__ void rows_by_rank(const UnitLayout& L, int lane, int* __restrict__ rank2row) {
    if (L.kind != 0) {
        if ((L.mask >> lane) & 1ull) rank2row[__popcll(L.mask & ((1ull << lane) - 1ull))] = lane;
    }
    __syncthreads();
}
// candidate of lane `lane` of unit p, or -1 for a lane that holds none (rank2row: rows_by_rank's table, or null: searched);
// col_off: the lane's steering column, counted from the unit's first (steering slices)
__device__ __forceinline__ int unit_candidate(const KP& P, const UnitLayout& L, int p, int lane, const int* __restrict__ rank2row,
                                              int* col_off = nullptr) {
    if (L.kind == 0) return p * 64 + lane;
    const int lg = __ffs(P.G) - 1;                           // G is a power of two
    int rank, j, off = 0;
    if (L.kind == 1) {                                       // lane = il * G + j; unit 0 holds the HIGHEST live rows (the ones that
        rank = L.R - 1 - (p * L.per + (lane >> lg));         // make most progress -- where the winner usually is: its cost is the
        j = lane & (P.G - 1);                                // incumbent the later units are pruned against, igt_fast64.h BOUND)
        if (rank < 0) rank = L.R;
    } else {
        int r;
        if (L.kind == 3) {                                   // candidate number g = r R + q of the scenario's live ones
            const int g = 64 * p + lane;
            r = small_div(g, L.R);
            rank = g - r * L.R;
            off = r - small_div(64 * p, L.R);
        } else {                                             // lane = il * per + jl
            const int il = small_div(lane, L.per);
            off = lane - il * L.per;
            r = p * L.per + off;
            rank = il;
        }
        if (r >= P.G) rank = L.R;
        j = (r & 1) ? P.G / 2 - 1 - (r >> 1) : P.G / 2 + (r >> 1);      // column rank r, handed out from the centre outwards
    }
    if (col_off) *col_off = off;
    if (rank >= L.R) return -1;
    int row;
    if (rank2row) row = rank2row[rank];
    else { unsigned long long m = L.mask; for (int t = 0; t < rank; ++t) m &= m - 1ull; row = __ffsll((long long)m) - 1; }
    return (row << lg) + j;
}
__device__ __forceinline__ int units_of_live_rows(const KP& P, int W, unsigned long long mask) {      // per lane (queue builder)
    return unit_layout<false>(P, W, P.cand_mode, mask).n_units;
}

// One workgroup per queue sorts its units into QC cost classes, most expensive first, keeping the scenario order
// inside a class (a stable counting sort, so the order is a function of the inputs alone).
// order[q][k] = (scenario ordinal in the queue) * 256 + slice.
constexpr int QC = 8, QB_THREADS = 1024, QB_TRIPS = 2;     // up to 2048 units per queue (B <= 8192 at W = 2)
// The body for one queue q, run by one workgroup of QB_THREADS_ threads in QB_TRIPS_ trips (QB_THREADS_ * QB_TRIPS_ = 2048): the
// stand-alone kernel below takes 1024 x 2; the float64 prelude (igt_kernels_f64.hip accel_rows_kernel, whose first eight
// workgroups sort the queues while the others roll the acceleration rows) 256 x 8.
// row_mask: the scenarios' live acceleration rows when they are known at this point (a slice beyond the live rows' units is
// a hole and sorts last); null: not known -- such slices are sorted as if they held candidates and the search skips them
// when it gets there.  by_rows: the tracking family's units are cut along the acceleration axis (unit-rank-major order).
template <typename T, int QB_THREADS_, int QB_TRIPS_>
__device__ __forceinline__ void build_queue(const KP& P, int B, int W, int q, const T* __restrict__ x0,
                                            const T* __restrict__ kparams, unsigned* __restrict__ order, int stride,
                                            unsigned* __restrict__ work_counter,
                                            const unsigned long long* __restrict__ row_mask, bool by_rows) {
    constexpr int QB_THREADS = QB_THREADS_, QB_TRIPS = QB_TRIPS_;
    __shared__ int cnt[QB_TRIPS][QB_THREADS / 64][QC];   // [trip][wave][class] counts, then exclusive offsets
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) work_counter[q * 64] = 0u;             // this queue's unit counter (saves the memset node)
    const int n_scen = (B + 7) / 8, n = n_scen * W;
    // 1. the classes, one thread per scenario (its W units share what is read of it): cls_s[j W + p]
    __shared__ unsigned char cls_s[QB_THREADS * QB_TRIPS];
#pragma unroll 2
    for (int j = tid; j < n_scen; j += QB_THREADS) {
        const int b = queue_scenario(q, j);
        int nu = 0;                                      // a hole of the last block of 8: sorts last, skipped by the search
        float frac_out = 0.95f, weight = 1.0f;
        if (b < B) {
            nu = row_mask ? units_of_live_rows(P, W, row_mask[b]) : W;
            const float s0 = (float)x0[(size_t)b * 7 + 2], v0 = (float)x0[(size_t)b * 7 + 5];
            const float b0 = (float)kparams[(size_t)b * 3 + 0], b1 = (float)kparams[(size_t)b * 3 + 1], kv = (float)kparams[(size_t)b * 3 + 2];
            const float reach = s0 + 1.5f * fmaxf(v0, 1.0f) * (float)(P.N * P.dt);
            const bool arc = kv != 0.0f && reach >= b0 && s0 <= b1;
            // share of the horizon rolled: the central slice lives to the end; the outer ones leave the lane the sooner the
            // faster the vehicle is -- except where the route bends within reach: there the outer steering columns are the
            // ones that can follow it, and every slice may live long; and the tracking family's units (acceleration rows,
            // steering by feedback) all do (91 % of their wave-steps are executed)
            frac_out = (arc || P.cand_mode == CAND_TRACK) ? 0.95f : fminf(fmaxf(1.05f - 0.15f * v0, 0.4f), 0.95f);
            weight = arc ? 1.9f : 1.0f;
        }
        // float64 tracking units (cut along the acceleration axis, highest rows in unit 0): unit-rank-major -- every scenario's
        // unit 0 before any unit 1 -- so that a later unit of a scenario starts when the earlier ones have left their best cost
        // as its incumbent (igt_fast64.h BOUND)
        const bool by_rank = P.cand_mode == CAND_TRACK && (by_rows || (sizeof(T) == 4 && (P.dev & (1 << 30)))) && !(P.dev & 262144);
        for (int p = 0; p < W; ++p) {
            int c = QC - 1;
            if (p < nu) {
                const float cost = (p == 0 ? 0.95f : frac_out) * weight;                              // 0.4 .. 1.8
                c = (int)((1.85f - cost) * ((float)QC / 1.5f));
                c = c < 0 ? 0 : (c > QC - 1 ? QC - 1 : c);
                if (by_rank) c = W <= QC ? p : (p * QC) / W;
            }
            cls_s[j * W + p] = (unsigned char)c;
        }
    }
    __syncthreads();
    // 2. an item's class and its rank among its wave's items of that class: in registers over two trips; over more (the
    // prelude's 256 threads) in LDS, the trips not unrolled
    constexpr bool IN_LDS = QB_TRIPS > 2;
    constexpr int UNROLL = IN_LDS ? 1 : QB_TRIPS;
    __shared__ unsigned short cls_rank[IN_LDS ? QB_THREADS * QB_TRIPS : 1];
    int cls[IN_LDS ? 1 : QB_TRIPS], rank[IN_LDS ? 1 : QB_TRIPS];
#pragma unroll UNROLL
    for (int t = 0; t < QB_TRIPS; ++t) {
        const int i = t * QB_THREADS + tid;
        const int c = i < n ? (int)cls_s[i] : -1;
        int r = 0;
#pragma unroll
        for (int cc = 0; cc < QC; ++cc) {
            const unsigned long long m = __ballot(c == cc);
            if (lane == 0) cnt[t][wv][cc] = __popcll(m);
            if (c == cc) r = __popcll(m & ((1ull << lane) - 1ull));
        }
        if constexpr (IN_LDS) cls_rank[t * QB_THREADS + tid] = (unsigned short)(((c < 0 ? 15 : c) << 8) | r);
        else { cls[t] = c; rank[t] = r; }
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
#pragma unroll UNROLL
    for (int t = 0; t < QB_TRIPS; ++t) {
        int c, r;
        if constexpr (IN_LDS) { const int cr = cls_rank[t * QB_THREADS + tid]; c = (cr >> 8) == 15 ? -1 : (cr >> 8); r = cr & 255; }
        else { c = cls[t]; r = rank[t]; }
        if (c < 0) continue;
        int off = cnt[t][wv][c] + r;
        for (int cc = 0; cc < c; ++cc) off += total[cc];      // the more expensive classes come first
        const int i = t * QB_THREADS + tid, j = i / W, p = i - j * W;
        order[(size_t)q * stride + off] = (unsigned)j * 256u + (unsigned)p;
    }
}
template <typename T>
__global__ __launch_bounds__(QB_THREADS) void build_queues_kernel(KP P, int B, int W, const T* __restrict__ x0,
                                                                  const T* __restrict__ kparams,
                                                                  unsigned* __restrict__ order, int stride,
                                                                  unsigned* __restrict__ work_counter,
                                                                  const unsigned long long* __restrict__ row_mask = nullptr) {
    build_queue<T, QB_THREADS, QB_TRIPS>(P, B, W, (int)blockIdx.x, x0, kparams, order, stride, work_counter, row_mask,
                                         row_mask != nullptr);
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
// p_major (no order table): item k of a queue is unit rank k / n_scen of scenario ordinal k mod n_scen -- every scenario's
// unit 0 before any unit 1 (the float64 tracking family's incumbents) -- instead of scenario-major.
template <class Unit>
__device__ __forceinline__ void search_waves(const KP& P, int B, int W, int queues, unsigned* __restrict__ work_counter,
                                             const unsigned* __restrict__ order, int order_stride, const Unit& unit,
                                             bool p_major = false) {
    const unsigned q = blockIdx.x % (unsigned)queues, uW = (unsigned)W;
    const unsigned n_scen = ((unsigned)B + 7u) / 8u;          // blocks of 8 scenarios; queue q takes one of each
    const unsigned K = n_scen * uW;
    const bool lane0 = (threadIdx.x & 63) == 0;
    // (round 4: 8 items per wave instead of 4 -- at B = 4096 that is every item, i.e. no index is fetched ahead: an item reserved
    // by a wave that is 100 us into an arc unit starts when that unit ends, while waves go idle from 75 % of the span on;
    // profiles/r04_hold_sweep.txt: search 0.228 -> 0.220 ms, one solve at a time 16.1 -> 16.7 M solves/s)
    const unsigned hold = ((P.dev >> 12) & 15u ? (P.dev >> 12) & 15u : 8u) * (gridDim.x / (unsigned)queues + 1u);
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
            unsigned j, p;
            if (ord) { j = item >> 8; p = item & 255u; }
            else if (p_major) { p = k / n_scen; j = k - p * n_scen; }
            else { j = k / uW; p = k - j * uW; }
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

}  // namespace igt
