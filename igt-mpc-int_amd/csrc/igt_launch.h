// igt_launch.h -- launcher declarations shared by igt_kernels.hip and igt_api.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "igt_device.h"
#include "igt_value_net.h"

namespace igt {

// trajectories kept by the search pass of a small double batch (igt_kernels_common.h CaptureSink)
constexpr int TRAJ_FIELDS = 9;                    // the 7 states, then a and delta_f
__host__ __device__ inline size_t traj_unit_doubles(int N) { return (size_t)TRAJ_FIELDS * (N + 1) * 64; }

template <typename T>
struct SolveArgs {   // all device pointers
    const T* x0;
    const T* u_prev;
    const T* kparams;
    const uint32_t* flags;
    const T* obs;
    const T* tv_sv;
    const T* enc;
    const double* table;   // [C,2,N] or null
    const double* cinf;    // [F,3] or null
    const double* cpar;    // [B,4] ramp-hold centre offset / span of a refinement pass, or null (first pass)
    const T* u_ws;         // [B,2,N] warm start (previous solution shifted by one step), or null; used where flags & 2
    Centre<T> centre() const { return Centre<T>{cpar, u_ws}; }
    T* x_out;
    T* u_out;
    T* cost_out;
    int32_t* argmin_out;
    int32_t* status_out;
    double* part_J;        // [B, W] per-slice best cost        (workspace)
    int32_t* part_c;       // [B, W] per-slice best candidate
    // value-net cost: per-candidate records written by the search pass, consumed by value_kernel
    T* rec_sN;             // [B, C]
    T* rec_vN;             // [B, C]
    double* rec_J;         // [B, C] cost without the terminal term
    uint32_t* rec_viol;    // [B, C]
    T* p_vec;              // [B, 128] first-layer offsets
    // float path: feasible candidates appended by the search pass (count, scenario, candidate in rec_viol's storage)
    unsigned* work_counter;         // search_fast_kernel's 8 unit counters, 256 B apart (zeroed before every launch)
    int n_cu;                       // compute units (sizes the persistent search grid)
    int waves_per_simd;             // of the persistent search grid: 2, or 1 when solves overlap (igt_set_concurrency); 0 = 2
    double* ckpt;                   // [ck_parts-1][B*Wk] horizon checkpoints of the search pass for emit (null: none)
    int ck_parts;                   // pieces the horizon is emitted in (igt_device.h Seg / Ckpt)
    unsigned* queue_order;          // [8][ceil(B/8) W] longest-first unit order of the search queues (null: index order)
    unsigned* rec_count;
    int32_t* rec_b;
    unsigned long long* best_key;   // [B] per-scenario (orderable cost, candidate) minimum
    int2* unit_seg;                 // double path: [B, C/64] (first entry, count) of each unit's entries in the compact list
    double* prune_thr;              // double path: [B] cost above which an entry cannot win (value_bound_kernel)
    unsigned* live_idx;             // double path: entries left for the network after value_prune_kernel
    unsigned long long* row_mask;   // double path: [B] live acceleration rows of the generated families (accel_rows_kernel)
    double* traj;                   // double path, small batches: [B C/64][9][N+1][64] kept by the search pass (null: none)
    bool ck_ok;                     // double path: part_J is followed by [B] masks, [B] incumbents, [B G] row travel sums and [B C/64][24] checkpoint records
};
constexpr int CK_RECORD_DOUBLES = 24;   // igt_fast64.h (CK_PARTS - 1) * CK_FIELDS

bool search_builds_queues(const KP& P, int B, const SolveArgs<float>& A);   // then no memset of the counters is needed
bool search_builds_queues(const KP& P, int B, const SolveArgs<double>& A);
bool search_is_static(const KP& P, int B, const SolveArgs<double>& A);      // one wave per unit: no queues, no counters
inline bool search_is_static(const KP&, int, const SolveArgs<float>&) { return false; }
template <typename T> hipError_t launch_search(const KP& P, int B, const SolveArgs<T>& A, int nc, hipStream_t st);
template <typename T> hipError_t launch_emit(const KP& P, int B, int W, const SolveArgs<T>& A, hipStream_t st);
// value-net cost: search pass that writes per-candidate records, then prep + MLP + per-chunk arg-min
template <typename T> hipError_t launch_search_records(const KP& P, int B, const SolveArgs<T>& A, hipStream_t st);
template <typename T>
hipError_t launch_value(const KP& P, int B, const DevNet<T>& net, const SolveArgs<T>& A, T* cost_all,
                        uint32_t* viol_all, hipStream_t st);
hipError_t prepare_value_kernels(int n_hidden_mats);   // once per igt_set_value_net: dynamic-LDS function attributes
hipError_t prepare_emit_kernels();                     // once per igt_create: the same for the float64 emit in pieces
template <typename T> hipError_t launch_reduce(int B, int W, const SolveArgs<T>& A, hipStream_t st);
// ramp-hold refinement: winner of the pass just finished -> centre/span of the next pass (cpar[B,4])
template <typename T>
hipError_t launch_refine(const KP& P, int B, int W, const SolveArgs<T>& A, double* cpar, int first, hipStream_t st);
template <typename T>
hipError_t launch_rollout_all(const KP& P, int B, const SolveArgs<T>& A, T* X_all, T* U_all, T* cost_all,
                              uint32_t* viol_all, hipStream_t st);   // value mode: also fills A.rec_*
template <typename T>
hipError_t launch_frenet_step(const KP& P, int n, const T* x, const T* u, const T* kparams, T* x_next, hipStream_t st);
template <typename T>
hipError_t launch_forecast(const KP& P, int B, const double* routes, int n_routes, const T* ego_xyh, const T* opp,
                           const T* opp_a, const int32_t* opp_route, const T* plan_x, const T* plan_u,
                           const int32_t* has_plan, T* obs_xy, T* tv_sv, hipStream_t st);
template <typename T> hipError_t launch_first_controls(int B, int N, const T* u, T* u0, hipStream_t st);
template <typename T>
hipError_t launch_cartesian(int n, int steps, double dt, double l_r, double l_f, const T* z0, const T* u, T* z_out,
                            hipStream_t st);

}  // namespace igt
