/*
 * igtmpc.h -- C ABI of the MI355X-native batched MPC rollout + shooting solver.
 *
 * The reference (hansungkim98122/IGT-MPC-INT) is pure Python and has NO FFI for
 * this path: its boundary is the Python object API that evaluate.py consumes
 * (MPC_Planner.update_initial_condition / update_predictions / solve,
 * mpc.py:241-294, 383-406).  Each entry point below therefore cites the
 * reference lines whose work it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-allocated buffers.
 *   - every function returns 0 on success, <0 on error (IGT_E_*); the message is
 *     available from igt_last_error() (thread-local).  Nothing throws.
 *   - `mem` says where the caller's buffers live: IGT_MEM_DEVICE (HBM pointers,
 *     work is enqueued on `stream` and the call returns without synchronising)
 *     or IGT_MEM_HOST (the library stages through its own device buffers and
 *     synchronises before returning).
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own non-blocking stream; pass
 *     hipStreamLegacy, (void*)1, to name the legacy default stream).
 *   - state order everywhere: [x, y, s, ey, epsi, v, psi]      (mpc.py:163)
 *   - array layouts are C-contiguous with the shapes written in the comments.
 *   - one handle per (device, thread); handles share no mutable state.  A handle owns its workspace: two solves
 *     on one handle must not overlap in time (use one handle per stream).
 *   - IGT_MEM_DEVICE calls only enqueue work (kernels and memset nodes; no allocation once the workspace has its
 *     size, no host synchronisation), so after one warm-up call a solve can be captured in a stream graph.
 */
#ifndef IGTMPC_H
#define IGTMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGT_VERSION 201

enum {
    IGT_OK = 0,
    IGT_E_INVALID = -1,   /* bad argument / unsupported parameter combination */
    IGT_E_HIP = -2,       /* a HIP runtime call failed                        */
    IGT_E_NOMEM = -3,
    IGT_E_STATE = -4      /* e.g. value-net cost requested but no net loaded  */
};

enum { IGT_MEM_DEVICE = 0, IGT_MEM_HOST = 1 };

/* candidate control-sequence families (build-defined: the reference's NLP has no
 * candidates; SURVEY.md section 8d defines the lattice used for the benchmark) */
enum {
    IGT_CAND_LATTICE = 0,  /* G x G lattice of constant per-step (da, ddf) increments */
    IGT_CAND_TABLE = 1,    /* explicit table U[C,2,N] shared by every scenario         */
    IGT_CAND_RAMP_HOLD = 2,/* G x G: a and df track, at the rate limits, base sequence + one of G offsets each; the
                              base is u_prev held over the horizon or -- igt_solve_batch_ws_* -- the warm start (the
                              previous solution shifted by one step); offsets are dense around 0 (first pass) and, with
                              refine_iters > 0, re-centred on the previous pass's winner with its grid cell's spacing */
    IGT_CAND_TRACK = 3     /* G x G: acceleration as IGT_CAND_RAMP_HOLD (offset i from the base sequence) with the target
                              kept under the envelope E_k = track_env dt^2 (N - k - 1/2) / (2 w_u) -- the acceleration
                              beyond which one more unit costs more effort (mpc.py:362) than it buys progress
                              (mpc.py:372); STEERING is a
                              state feedback evaluated inside the roll-out: beta_cmd = clamp(-epsi - track_ke * ey + off_j),
                              df tracks atan(tan(beta_cmd) (l_f + l_r) / l_r) at the steering-rate limit.  Every
                              acceleration profile thereby gets the steering that belongs to where it actually is;
                              the realised (a_k, df_k) are returned as u_out like any other candidate's            */
};

/* cost (mpc.py:356-373) */
enum {
    IGT_COST_PROGRESS = 0, /* ... - (s_N - s_0)                 eval_mode mpc,    mpc.py:372 */
    IGT_COST_VALUE_NET = 1 /* ... - (V(Wn(x_N-mu))*sig + mu_t)  eval_mode gt_mpc, mpc.py:369 */
};

/* per-scenario flag bits (flags[B]) */
#define IGT_FLAG_ABS_HEADING 1u /* ego route in {'32','41'}: psi_0 = |psi_0|  (mpc.py:231-234, 282-285) */
#define IGT_FLAG_WARM 2u        /* the scenario's row of u_ws holds a warm start (igt_solve_batch_ws_*)          */

/* violation bits reported by igt_rollout_batch_* (viol_out) */
#define IGT_VIOL_BOX_V 1u      /* mpc.py:316-317  k = 0..N-1 */
#define IGT_VIOL_BOX_U 2u      /* mpc.py:318-321            */
#define IGT_VIOL_RATE 4u       /* mpc.py:301-312  (table candidates only; lattices satisfy it by construction) */
#define IGT_VIOL_EY 8u         /* mpc.py:296-299  k = 0..N   */
#define IGT_VIOL_TERMINAL 16u  /* mpc.py:177-180  C_inf * [v_{N-1}; a_{N-1}] <= b */
#define IGT_VIOL_COLLISION 32u /* mpc.py:223-226  k = 1..N   */
#define IGT_VIOL_NONFINITE 64u

#define IGT_MAX_N 64
#define IGT_MAX_CINF 256
#define IGT_MAX_OBS 4

typedef struct igt_handle igt_handle;

/* The constants MPC_Planner.__init__ hard-codes (mpc.py:45-62) plus the
 * discretisation (N, dt, n_rk4) and the candidate family. */
typedef struct igt_params {
    int32_t N;         /* horizon                      (mpc.py:38; evaluate.py:69)  */
    int32_t n_rk4;     /* RK4 sub-steps per control step (evaluate.py:109 -> 4)     */
    int32_t C;         /* candidates per solve; multiple of 64; lattices need C = G*G */
    int32_t n_obs;     /* obstacles per scenario = M-1  (mpc.py:83)                 */
    int32_t cand_mode; /* IGT_CAND_*  */
    int32_t cost_mode; /* IGT_COST_*  */
    double dt;         /* fourwayint.yaml:2  */
    double l_r, l_f;   /* mpc.py:49-50       */
    double v_min, v_max, a_min, a_max, df_max; /* mpc.py:57-62 */
    double jerk_limit;       /* mpc.py:56 */
    double steer_rate_limit; /* mpc.py:55 */
    double ey_lim;           /* mpc.py:61 */
    double d_min;            /* 2*ca_radius, mpc.py:45 */
    double w_u;              /* 0.05, mpc.py:362 */
    double feas_tol;         /* inequality verdicts are g <= feas_tol */
    int32_t refine_iters;    /* IGT_CAND_RAMP_HOLD / IGT_CAND_TRACK: extra search passes around the winner (0..4) */
    int32_t reserved;
    double track_ke;         /* IGT_CAND_TRACK: lateral-error gain of the steering feedback [1/m]          (0.3)  */
    double track_span;       /* IGT_CAND_TRACK: the G slip-angle offsets span +-track_span [rad]            (0.1)  */
    double track_beta_lim;   /* IGT_CAND_TRACK: |beta_cmd| limit [rad]                                      (0.7)  */
    double track_env;        /* IGT_CAND_TRACK: scale of the acceleration envelope E_k; 0 = no envelope     (1.0)
                                Applied with IGT_COST_PROGRESS only: E_k is derived from the progress term
                                (mpc.py:372), which the IGT_COST_VALUE_NET cost does not have (mpc.py:367-370) */
    double track_vcap;       /* IGT_CAND_TRACK: > 0: the acceleration targets also stay under the speed cap -- the largest
                                a_k from which a jerk-limited ramp to a = 0 (mpc.py:301-304) still keeps v <= v_max
                                (mpc.py:316-317): a candidate with a large offset accelerates at the limits and arrives
                                at v_max with a = 0 instead of failing the speed box; 0 = off                  (1.0)  */
} igt_params;

/* Fills *p with the reference's numbers: N=20, dt=0.1, n_rk4=4, C=256, n_obs=1,
 * lattice candidates, progress cost, mpc.py:45-62 limits, feas_tol=1e-6. */
int igt_params_default(igt_params* p);

const char* igt_last_error(void);
int igt_version(void);

/* Replaces MPC_Planner.__init__ (mpc.py:21-160) for a whole batch: no NLP is
 * built; the handle owns a stream, staging buffers and the constant tables. */
int igt_create(const igt_params* p, int device, igt_handle** out);
int igt_destroy(igt_handle* h);
int igt_get_params(const igt_handle* h, igt_params* out);

/* Terminal set C_inf as half-planes A[F,2] * (v, a) <= b[F]
 * (mpc.py:88-104 builds it with polytope; 177-180 applies it). F = 0 disables. */
int igt_set_cinf(igt_handle* h, const double* A, const double* b, int32_t F);

/* Explicit candidate table U[C,2,N] (row 0 = a_k, row 1 = df_k) for IGT_CAND_TABLE. */
int igt_set_candidate_table(igt_handle* h, const double* U);

/* Terminal value network (mpc.py:108-127, 367-369; model.py:14-51).
 * n_layers Linear layers, dims[n_layers+1] = {6,128,...,1}; weights holds, per
 * layer, W[out,in] row-major followed by b[out].  Wn[6,6], mu_f[6]: input
 * whitening x -> Wn (x - mu_f); sigma_t, mu_t: target de-normalisation. */
int igt_set_value_net(igt_handle* h, int32_t n_layers, const int32_t* dims, const double* weights,
                      const double* Wn, const double* mu_f, double sigma_t, double mu_t);

/* The batched solve.  Replaces, per scenario b, one
 *   update_initial_condition (mpc.py:280-294) + update_predictions (241-278) + solve (383-406)
 * of the reference by: generate C candidate control sequences from u_prev[b],
 * roll each through the RK4 Frenet bicycle model
 * (kinematic_bicycle_model_frenet.py:70-127 == mpc.py:201-209), evaluate the cost
 * (mpc.py:356-373) and every constraint (mpc.py:177-180, 223-226, 296-321), and
 * return the feasible arg-min (ties -> lowest candidate index).
 *
 *   x0      [B,7]              initial state                       (mpc.py:280-292)
 *   u_prev  [B,2]              previously applied (a, df)          (mpc.py:286)
 *   kparams [B,3]              curvature (b0, b1, Kv): K(s)=Kv on [b0,b1) else 0;
 *                              straight routes pass (inf, inf, 0)  (mpc.py:183-200)
 *   flags   [B]                IGT_FLAG_*
 *   obs_xy  [B,n_obs,2,N+1]    obstacle x / y predictions, already passed through
 *                              filter_preds (utils.py:365-388); k = 1..N are read
 *   tv_sv   [B,2]  enc [B,2]   value-net cost only (may be NULL otherwise):
 *                              (s,v) of the other vehicle's last raw prediction and
 *                              (e_ego, e_tv) scenario encodings     (mpc.py:326-338)
 *   x_out   [B,7,N+1]  u_out [B,2,N]   arg-min trajectory / controls (mpc.py:401)
 *   cost_out[B]  argmin_out[B]  status_out[B]
 *        status 0 <-> is_opt True; 1 <-> no feasible candidate (is_opt False,
 *        mpc.py:402-406): then argmin = -1, cost = +inf, x_out/u_out = NaN.
 *
 * _f64: double everywhere -- the reference's precision.  The RK4 stages are evaluated in factorised form
 *       (csrc/igt_fast64.h); <= 1e-9 of the float64 oracle on every trajectory (measured ~1e-14).
 * _f32: float storage, float stage derivatives, double state accumulators; within 1e-5*max(1,|ref|) of the
 *       float64 oracle except where a curvature switch is decided inside float32 noise.
 */
int igt_solve_batch_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev,
                        const float* kparams, const uint32_t* flags, const float* obs_xy,
                        const float* tv_sv, const float* enc, float* x_out, float* u_out,
                        float* cost_out, int32_t* argmin_out, int32_t* status_out, int mem,
                        void* stream);
int igt_solve_batch_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev,
                        const double* kparams, const uint32_t* flags, const double* obs_xy,
                        const double* tv_sv, const double* enc, double* x_out, double* u_out,
                        double* cost_out, int32_t* argmin_out, int32_t* status_out, int mem,
                        void* stream);

/* The same solve with a warm start -- what the reference passes to solve(x_sol_prev, u_sol_prev) after
 * augment_prev_sol (utils.py:354-363; call site evaluate.py:478-482): the previous solution's controls shifted by one
 * step and extended by repeating the last one.  IPOPT starts its iterations there; the shooting solver centres its
 * candidate set there: IGT_CAND_RAMP_HOLD targets become u_ws[b,:,k] + offset (instead of u_prev + offset), so the
 * shifted previous plan is itself candidate (G/2, G/2).  The state part x_sol_prev has no counterpart (shooting states
 * are implied by the controls).
 *   u_ws [B,2,N]   rows are read only where flags[b] & IGT_FLAG_WARM; NULL = no warm start anywhere.
 * Needs cand_mode IGT_CAND_RAMP_HOLD or IGT_CAND_TRACK (acceleration base only) when u_ws != NULL. */
int igt_solve_batch_ws_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev,
                           const float* kparams, const uint32_t* flags, const float* obs_xy,
                           const float* tv_sv, const float* enc, const float* u_ws, float* x_out,
                           float* u_out, float* cost_out, int32_t* argmin_out, int32_t* status_out,
                           int mem, void* stream);
int igt_solve_batch_ws_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev,
                           const double* kparams, const uint32_t* flags, const double* obs_xy,
                           const double* tv_sv, const double* enc, const double* u_ws, double* x_out,
                           double* u_out, double* cost_out, int32_t* argmin_out, int32_t* status_out,
                           int mem, void* stream);

/* Debug / parity entry: every candidate of every scenario.
 *   X_all [B,C,7,N+1] (may be NULL)   U_all [B,C,2,N] (may be NULL)
 *   cost_all [B,C]   viol_all [B,C] (IGT_VIOL_* bits; 0 = feasible) */
int igt_rollout_batch_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev,
                          const float* kparams, const uint32_t* flags, const float* obs_xy,
                          const float* tv_sv, const float* enc, float* X_all, float* U_all,
                          float* cost_all, uint32_t* viol_all, int mem, void* stream);
int igt_rollout_batch_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev,
                          const double* kparams, const uint32_t* flags, const double* obs_xy,
                          const double* tv_sv, const double* enc, double* X_all, double* U_all,
                          double* cost_all, uint32_t* viol_all, int mem, void* stream);

int igt_rollout_batch_ws_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev,
                             const float* kparams, const uint32_t* flags, const float* obs_xy,
                             const float* tv_sv, const float* enc, const float* u_ws, float* X_all,
                             float* U_all, float* cost_all, uint32_t* viol_all, int mem, void* stream);
int igt_rollout_batch_ws_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev,
                             const double* kparams, const uint32_t* flags, const double* obs_xy,
                             const double* tv_sv, const double* enc, const double* u_ws, double* X_all,
                             double* U_all, double* cost_all, uint32_t* viol_all, int mem, void* stream);

/* One control step of the RK4 Frenet bicycle model for n independent states -- the model
 * object the reference's driver calls directly for warm-start extension and for the
 * brake fallback (kinematic_bicycle_model_frenet.py:16-192 numpy branch; call sites
 * utils.py:358, evaluate.py:520).  Uses the handle's dt, n_rk4, l_r, l_f.
 *   x [n,7]   u [n,2] = (a, df)   kparams [n,3]   x_next [n,7]
 * _f64 follows the reference operation for operation (<= 1e-12 of the golden transitions).  _f32 uses the float
 * arithmetic of the solver with its long stage-offset polynomials: within 1e-5*max(1,|ref|) for |v| <= 25 m/s. */
int igt_frenet_step_f32(igt_handle* h, int32_t n, const float* x, const float* u, const float* kparams,
                        float* x_next, int mem, void* stream);
int igt_frenet_step_f64(igt_handle* h, int32_t n, const double* x, const double* u, const double* kparams,
                        double* x_next, int mem, void* stream);

/* ---- opponent forecast on the device ------------------------------
 * Route geometry for igt_forecast_batch_*: n_routes x 12 doubles per route
 *   (p0x, p0y, tx, ty, cx, cy, b0, b1, R, endx, endy, straight)
 * = start point of s, unit tangent, side the arc bends to, arc interval [b0,b1], radius, the frozen end
 * coordinates the reference uses after the arc, straight flag (utils.py:532-586 frenet2global, table-driven). */
int igt_set_routes(igt_handle* h, int32_t n_routes, const double* table);

/* What each ego planner is told about its opponent at a timestep, for B problems at once.  Replaces
 * ConstantAccelerationModel.predict (constant_acceleration_model.py:18-82), share_motion_forecasts
 * (utils.py:339-352: an opponent that solved last step shares its plan, shifted by one step and extended by
 * one predicted step; a = 0 if that step would exceed v = 5) and filter_preds (utils.py:365-388: an opponent
 * behind the ego is moved to (-20,-20)).  Scenes of M = n_obs + 1 vehicles (mpc.py:82-83 is written for any M; the shipped
 * fourwayint.yaml has M = 2): every per-opponent array carries an n_obs axis behind B, the opponents of a scene are forecast,
 * shared and filtered independently (with n_obs = 1 the shapes are the two-vehicle ones, [B,4], [B], ...).
 *   ego_xyh  [B,3]  ego x, y, heading                opp [B,n_obs,4]  opponent x, y, s, v (current)
 *   opp_a    [B,n_obs]  opponent's last applied a    opp_route [B,n_obs]  route id (row of the igt_set_routes table)
 *   plan_x [B,n_obs,7,N+1], plan_u [B,n_obs,2,N], has_plan [B,n_obs] (int32, 0 = no shared plan)  -- all three may be NULL
 *   obs_xy [B,n_obs,2,N+1] (out)   tv_sv [B,n_obs,2] (out): (s, v) of every opponent's last forecast state (mpc.py:263-276; the
 *   value network's features read the one of a two-vehicle scene, mpc.py:330) */
int igt_forecast_batch_f32(igt_handle* h, int32_t B, const float* ego_xyh, const float* opp, const float* opp_a,
                           const int32_t* opp_route, const float* plan_x, const float* plan_u,
                           const int32_t* has_plan, float* obs_xy, float* tv_sv, int mem, void* stream);
int igt_forecast_batch_f64(igt_handle* h, int32_t B, const double* ego_xyh, const double* opp, const double* opp_a,
                           const int32_t* opp_route, const double* plan_x, const double* plan_u,
                           const int32_t* has_plan, double* obs_xy, double* tv_sv, int mem, void* stream);

/* 4-state Cartesian forward-Euler bicycle (kinematic_bicycle_model.py:15-50), the
 * model ReferenceGen.py steps to lay out reference paths.
 *   z0 [n,4] = (x, y, psi, v)   u [n,2,T] (a, df)   z_out [n,4,T+1] */
int igt_cartesian_euler_f32(igt_handle* h, int32_t n, int32_t T, const float* z0, const float* u,
                            float* z_out, int mem, void* stream);
int igt_cartesian_euler_f64(igt_handle* h, int32_t n, int32_t T, const double* z0, const double* u,
                            double* z_out, int mem, void* stream);

/* ---- multi-GPU: the one exchange of the path ---------------------------------------------------
 * Scenarios are independent (evaluate.py:469-558: the agents of a timestep solve against the same predictions), so a
 * batch shards into contiguous blocks, one process and one handle per GPU, with NO collective on the data path.  The
 * only exchange is an all-gather of the first-step controls u*[:, :, 0] -- the (a, df) every agent applies
 * (evaluate.py:492) -- so that each rank holds the whole action vector.  RCCL (ncclAllGather over xGMI) is bound at run
 * time with dlopen: a single-GPU user needs no RCCL, and a host that already carries one (torch ships librccl.so.1,
 * and `torch.distributed` is how bench.py and igtmpc/sharding.py do the same exchange) is not given a second copy.
 *   igt_comm_unique_id : rank 0 creates the 128-byte id; the caller distributes it (MPI_Bcast, a file, a socket).
 *   igt_comm_init      : every rank, same id; one communicator per handle.
 *   igt_allgather_controls_* : u_out [B_local,2,N] (device) -> u0_all [world*B_local,2] (device), rank-major, enqueued
 *                        on `stream`.  Equal shards (ncclAllGather); ragged shards are padded by the caller: with a
 *                        communicator EVERY rank must call with the same B_local > 0 -- an empty shard or a size
 *                        that differs from the communicator's first call returns IGT_E_INVALID instead of leaving the
 *                        peers waiting inside the collective.  Without a communicator (world = 1) it reduces to the
 *                        strided copy u0_all = u_out[:, :, 0] (B_local = 0: nothing to do). */
#define IGT_COMM_ID_BYTES 128
int igt_comm_unique_id(void* id_out /* IGT_COMM_ID_BYTES */);
int igt_comm_init(igt_handle* h, int32_t world, int32_t rank, const void* id);
int igt_comm_destroy(igt_handle* h);
int igt_allgather_controls_f32(igt_handle* h, int32_t B_local, const float* u_out, float* u0_all, void* stream);
int igt_allgather_controls_f64(igt_handle* h, int32_t B_local, const double* u_out, double* u0_all, void* stream);

/* How many solves the caller keeps in flight on this device at a time -- on other handles and streams -- including this
 * handle's (default 1).  The search kernels are persistent: with 1 a kernel starts as many waves as the device holds, two per
 * SIMD, which is fastest for a solve that has the device to itself.  When solves overlap, a second kernel only gets wave
 * slots as the first one drains, and a slot that has been vacated stays idle for tens of microseconds before the next
 * workgroup runs there; with `solves_in_flight` >= 3 a search kernel takes one wave per SIMD, so that two kernels are
 * resident side by side and each one's waves run faster while the other's are being replaced (measured at B = 4096 with four
 * solves in flight: 0.187 -> 0.176 ms per step; one solve alone: 0.277 -> 0.324, hence a setting and not the default; with
 * two in flight the streams of this runtime end up on one hardware queue and run one after the other, so 2 is treated as 1).
 * From 3 on, the float64 emit pass also stays one lean kernel (one wave per 64 scenarios, no LDS) instead of the five-wave,
 * LDS-staged emit in pieces that a solve with the device to itself uses to cut the winner roll-out's latency: between two
 * persistent search kernels only the lean one gets on the chip.
 * Results do not depend on it.  IGT_E_INVALID unless 1 <= solves_in_flight <= 64. */
int igt_set_concurrency(igt_handle* h, int32_t solves_in_flight);

/* Workspace (owned by the handle, grown on the first solve of a batch size, never shrunk; growth synchronises the stream and is
 * refused under stream capture with IGT_E_STATE).  Per scenario, C = 256: 48 B of slice partials, 16 B of live-row masks and
 * incumbent keys (float64; float32: 8 B), 128 B of the acceleration rows' travel sums (float64), 768 B of horizon checkpoints of
 * the unit winners (float64, progress cost, N >= 8),
 * 288 B of queue order / counters; value-network cost: + 36 B per candidate (the list of feasible candidates).  Small float64
 * batches (no more 64-candidate units than the device has SIMDs: B <= 256 at C = 256) keep every candidate's trajectory for
 * the emit pass: 9 (N + 1) 64 doubles per unit = 97 KB per unit at N = 20, i.e. up to 99 MB per handle at B C / 64 = 1024. */

/* Per-kernel timing with HIP events on the launch stream (used by bench.py for the
 * roofline line).  While enabled, every solve records events around its kernels;
 * igt_get_kernel_ms synchronises on them and returns the last call's durations. */
int igt_set_profiling(igt_handle* h, int enable);
int igt_get_kernel_ms(igt_handle* h, float* search_ms, float* emit_ms);

/* Algorithmic HBM bytes one solve moves: reads + writes. */
int igt_algorithmic_bytes_per_solve(const igt_handle* h, int elem_size, int64_t* read_bytes,
                                    int64_t* write_bytes);

#ifdef __cplusplus
}
#endif
#endif /* IGTMPC_H */
