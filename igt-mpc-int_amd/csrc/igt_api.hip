// igt_api.hip -- the C ABI declared in include/igtmpc.h (handle management, argument
// validation, host staging, launches).  No exceptions cross the boundary.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "igt_launch.h"
#include "igtmpc.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(IGT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

}  // namespace

struct igt_handle {
    igt_params p;
    igt::KP kp;
    int device;
    hipStream_t stream;
    double* d_cinf;
    double* d_table;
    bool table_set;
    bool net_set;
    void* d_stage;
    size_t stage_bytes;
    void* h_stage;         // pinned mirror of the first PACK_BYTES of d_stage (small host-mode solves: one copy each way)
    void* d_work;          // workspace: per-slice partial arg-min, value-net records
    size_t work_bytes;
    double* d_routes;      // [n_routes, 12] route geometry for the forecast kernel
    int n_routes;
    void* d_net;           // value-net parameters (float block followed by double block)
    igt::DevNet<float> net_f;
    igt::DevNet<double> net_d;
    bool prof;
    hipEvent_t ev[3];
    bool ev_recorded;
    int nc;
    int n_cu;              // compute units of the device (sizes the persistent search grid)
    int concurrency;       // solves the caller keeps in flight on the device (igt_set_concurrency)
    void* comm;            // RCCL communicator of igt_comm_init (null: none)
    int comm_world, comm_rank;
    int32_t comm_B_local;  // shard size of the communicator's first all-gather (0 = none yet); later calls must match
    void* d_u0;            // [B_local,2] first-step controls staged for the all-gather
    size_t u0_bytes;
    int dev_ckpt;          // IGT_DEV_CKPT / IGT_DEV_TRAJ_MAX (threshold sweeps), read once at igt_create; -1: not set
    int dev_traj_max;
};

namespace {

// RCCL is bound at run time (dlopen), not at link time: a single-GPU user needs no RCCL at all, and a process that
// already carries one (torch ships its own librccl.so.1) must not get a second copy.
// Lifetime: the binding is opened by the first user -- igt_comm_unique_id for the length of the call, igt_comm_init for as
// long as its communicator lives -- and closed (dlclose) when the last communicator is destroyed (igt_comm_destroy /
// igt_destroy).  Nothing is left to static destructors: `g_rccl` is plain data, and at process exit no handle of the
// library refers to a DSO whose own teardown order it does not control ("exit-time abort").
struct IgtNcclId { char internal[IGT_COMM_ID_BYTES]; };
struct Rccl {
    void* lib;
    int users;
    int (*GetUniqueId)(IgtNcclId*);
    int (*CommInitRank)(void**, int, IgtNcclId, int);
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
    int (*CommDestroy)(void*);
    const char* (*GetErrorString)(int);
};
Rccl g_rccl = {nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
std::mutex g_rccl_mu;          // handles are per thread-group; two of them may reach igt_comm_* together

// one more user of the binding (opens it when there is none); null: RCCL cannot be loaded
Rccl* rccl_acquire() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    Rccl& r = g_rccl;
    if (!r.lib) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // one the process already mapped
        for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) return nullptr;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
            dlclose(r.lib);
            r.lib = nullptr;
            return nullptr;
        }
    }
    ++r.users;
    return &r;
}
// the binding of a live communicator (its handle holds a use: never null while h->comm is set)
Rccl* rccl() { return g_rccl.lib ? &g_rccl : nullptr; }
void rccl_release() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    Rccl& r = g_rccl;
    if (r.users > 0 && --r.users == 0 && r.lib) {
        dlclose(r.lib);          // drops this library's reference; a copy torch mapped stays mapped for torch
        r.lib = nullptr;
        r.GetUniqueId = nullptr; r.CommInitRank = nullptr; r.AllGather = nullptr; r.CommDestroy = nullptr; r.GetErrorString = nullptr;
    }
}
int rccl_fail(Rccl* r, const char* what, int rc) {
    return fail(IGT_E_HIP, std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

int isqrt_exact(int c) {
    int g = (int)std::lround(std::sqrt((double)c));
    return g * g == c ? g : -1;
}

int validate(const igt_params& p, std::string& why) {
    if (p.N < 1 || p.N > IGT_MAX_N) { why = "N must be in [1, IGT_MAX_N]"; return -1; }
    if (p.n_rk4 < 1 || p.n_rk4 > 64) { why = "n_rk4 must be in [1, 64]"; return -1; }
    if (p.C < 64 || p.C % 64) { why = "C must be a positive multiple of 64"; return -1; }
    if (p.n_obs < 0 || p.n_obs > IGT_MAX_OBS) { why = "n_obs must be in [0, IGT_MAX_OBS]"; return -1; }
    if (!(p.dt > 0) || !(p.l_r > 0) || !(p.l_f > 0)) { why = "dt, l_r, l_f must be positive"; return -1; }
    if (p.cand_mode == IGT_CAND_LATTICE) {
        const int g = isqrt_exact(p.C);
        if (g < 2 || 64 % g) { why = "lattice candidates need C = G*G with G in {2,4,8,16,32,64}; use IGT_CAND_TABLE"; return -1; }
    } else if (p.cand_mode == IGT_CAND_RAMP_HOLD || p.cand_mode == IGT_CAND_TRACK) {
        const int g = isqrt_exact(p.C);
        if (g < 4 || 64 % g) { why = "ramp-hold candidates need C = G*G with G in {4,8,16,32,64}"; return -1; }
    } else if (p.cand_mode != IGT_CAND_TABLE) {
        why = "unknown cand_mode"; return -1;
    }
    if (p.refine_iters < 0 || p.refine_iters > 4) { why = "refine_iters must be in [0, 4]"; return -1; }
    if (p.refine_iters > 0 && p.cand_mode != IGT_CAND_RAMP_HOLD && p.cand_mode != IGT_CAND_TRACK) { why = "refine_iters needs IGT_CAND_RAMP_HOLD or IGT_CAND_TRACK"; return -1; }
    if (p.cand_mode == IGT_CAND_TRACK && (!(p.track_ke >= 0) || !(p.track_span >= 0) || !(p.track_beta_lim > 0) || !(p.track_beta_lim < 1.5) || !(p.track_env >= 0) || !(p.track_env < 1e6) || !(p.track_vcap >= 0) || !(p.track_vcap < 1e6))) { why = "track_ke, track_span, track_env, track_vcap must be >= 0 (and finite) and 0 < track_beta_lim < 1.5"; return -1; }
    if (p.cost_mode != IGT_COST_PROGRESS && p.cost_mode != IGT_COST_VALUE_NET) { why = "unknown cost_mode"; return -1; }
    if (!(p.v_min <= p.v_max) || !(p.a_min <= p.a_max) || !(p.df_max >= 0)) { why = "inconsistent limits"; return -1; }
    if (!(p.feas_tol >= 0)) { why = "feas_tol must be >= 0"; return -1; }
    return 0;
}

igt::KP make_kp(const igt_params& p, int F) {
    igt::KP k;
    k.N = p.N; k.n_rk4 = p.n_rk4; k.C = p.C; k.n_obs = p.n_obs;
    k.cand_mode = p.cand_mode; k.cost_mode = p.cost_mode; k.F = F;
    k.G = p.cand_mode == IGT_CAND_TABLE ? 1 : isqrt_exact(p.C);
    k.refine_it = 0;
    k.df_small = p.df_max < 0.78 ? 1 : 0;
    { const char* e = getenv("IGT_DEV_FLAGS"); k.dev = (e ? atoi(e) : 0) & 0x1FFFFFFF; }      // bits 29 and 30 are set by the launchers
    k.dt = p.dt;
    k.h = p.dt / p.n_rk4;                      // frenet.py:93
    k.l_r = p.l_r;
    k.lr_ratio = p.l_r / (p.l_f + p.l_r);      // frenet.py:72
    k.v_min = p.v_min; k.v_max = p.v_max; k.a_min = p.a_min; k.a_max = p.a_max; k.df_max = p.df_max;
    k.rate_a = p.dt * p.jerk_limit;            // mpc.py:304
    k.rate_df = p.dt * p.steer_rate_limit;     // mpc.py:306
    k.ey_lim = p.ey_lim;
    k.dmin2 = p.d_min * p.d_min;               // mpc.py:226
    k.w_u = p.w_u;
    k.tol = p.feas_tol;
    k.trk_ke = p.track_ke; k.trk_span = p.track_span; k.trk_blim = p.track_beta_lim;
    // slope of the acceleration envelope (igt_device.h track_accel_target_uncapped); products and one quotient only, so the
    // oracle's track_env_slope() gives the same bits
    // The envelope is the stationary point of  w_u a_k^2 - (s_N - s_0)  (mpc.py:362, 372).  The gt_mpc cost has no
    // progress term (mpc.py:367-370: the value network stands there), so with IGT_COST_VALUE_NET it is not applied.
    k.trk_env = (p.track_env > 0 && p.w_u > 0 && p.cost_mode == IGT_COST_PROGRESS)
                    ? p.track_env * p.dt * p.dt / (2 * p.w_u) : (double)INFINITY;
    // speed cap of the acceleration targets (igt_device.h track_speed_cap): the speed it looks ahead to -- the box less a margin
    // far above a float32 roll-out's rounding (oracle: TRACK_VCAP_MARGIN) --, +inf when off
    k.trk_vmax = p.track_vcap > 0 ? p.v_max - 1e-4 : (double)INFINITY;
    k.inv_dt = 1.0 / p.dt;
    k.inv_rate_a = 1.0 / (p.dt * p.jerk_limit);
    // stage-offset polynomials: short form while h * (largest angular rate a candidate can reach) stays
    // small (igt_fast.h small_sincos2); v up to v_max + 2, |K| up to 0.25, sin(beta)/l_r <= 0.7/l_r
    const double vhi = std::fmax(std::fabs(p.v_min), std::fabs(p.v_max)) + 2.0;
    k.hi_order = (k.h * vhi * (0.7 / p.l_r + 0.25) > 0.12) ? 1 : 0;
    return k;
}

// A host-mode solve of a few scenarios is a dozen sub-kilobyte copies around ~120 us of kernels, and every pageable
// hipMemcpyAsync costs ~7 us whatever its size.  Up to PACK_BYTES the inputs are gathered into a pinned mirror of the
// staging arena and cross the bus in ONE copy, the outputs come back in one; beyond that the copies go directly (a second
// pass over megabytes on the host would cost more than the calls).
constexpr size_t PACK_BYTES = 256 * 1024;

int ensure_stage(igt_handle* h, size_t bytes) {
    if (!h->h_stage) HIPCHK(hipHostMalloc(&h->h_stage, PACK_BYTES, hipHostMallocDefault));
    if (bytes <= h->stage_bytes) return 0;
    if (h->d_stage) { HIPCHK(hipFree(h->d_stage)); h->d_stage = nullptr; h->stage_bytes = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&h->d_stage, want));
    h->stage_bytes = want;
    return 0;
}

// true while `st` is recording a stream graph: nothing that synchronises or allocates may be issued then
bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

int ensure_work(igt_handle* h, size_t bytes, hipStream_t st) {
    if (bytes <= h->work_bytes) return 0;
    if (capturing(st))
        return fail(IGT_E_STATE, "workspace too small for stream capture: run one eager solve of this batch size (and "
                                 "cost / candidate mode) on this handle first");
    HIPCHK(hipStreamSynchronize(st));
    if (h->d_work) { HIPCHK(hipFree(h->d_work)); h->d_work = nullptr; h->work_bytes = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&h->d_work, want));
    h->work_bytes = want;
    return 0;
}

template <typename T> const igt::DevNet<T>& net_of(const igt_handle* h);
template <> const igt::DevNet<float>& net_of<float>(const igt_handle* h) { return h->net_f; }
template <> const igt::DevNet<double>& net_of<double>(const igt_handle* h) { return h->net_d; }

struct Arena {   // carves 256-byte aligned pieces out of the staging buffer
    char* base;
    size_t off;
    template <typename U> U* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        U* p = reinterpret_cast<U*>(base + off);
        off += n * sizeof(U);
        return p;
    }
};

// the queue builder of small float batches zeroes the search kernel's unit counters itself
template <typename T>
inline bool counters_by_builder(const igt::KP& kp, int B, const igt::SolveArgs<T>& A) {
    return igt::search_is_static(kp, B, A) || igt::search_builds_queues(kp, B, A);      // ... or the search uses none
}

template <typename T>
int solve_impl(igt_handle* h, int32_t B, const T* x0, const T* u_prev, const T* kparams, const uint32_t* flags,
               const T* obs_xy, const T* tv_sv, const T* enc, const T* u_ws, T* x_out, T* u_out, T* cost_out,
               int32_t* argmin_out, int32_t* status_out, int mem, void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (B < 0) return fail(IGT_E_INVALID, "B < 0");
    if (B == 0) return IGT_OK;
    if (!x0 || !u_prev || !kparams || !flags || !x_out || !u_out || !cost_out || !argmin_out || !status_out)
        return fail(IGT_E_INVALID, "null buffer");
    const igt_params& p = h->p;
    if (p.n_obs > 0 && !obs_xy) return fail(IGT_E_INVALID, "obs_xy is null but n_obs > 0");
    if (p.cand_mode == IGT_CAND_TABLE && !h->table_set) return fail(IGT_E_STATE, "candidate table not set");
    if (u_ws && p.cand_mode != IGT_CAND_RAMP_HOLD && p.cand_mode != IGT_CAND_TRACK)
        return fail(IGT_E_INVALID, "a warm start needs IGT_CAND_RAMP_HOLD or IGT_CAND_TRACK (the families whose targets are centred on it)");
    const bool value = p.cost_mode == IGT_COST_VALUE_NET;
    if (value) {
        if (!h->net_set) return fail(IGT_E_STATE, "value net not set (igt_set_value_net)");
        if (!tv_sv || !enc) return fail(IGT_E_INVALID, "tv_sv / enc required for the value-net cost");
    }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t n_x = (size_t)B * 7, n_u = (size_t)B * 2, n_k = (size_t)B * 3;
    const size_t n_obs = (size_t)B * p.n_obs * 2 * (p.N + 1);
    const size_t n_xo = (size_t)B * 7 * (p.N + 1), n_uo = (size_t)B * 2 * p.N;

    igt::SolveArgs<T> A{};
    bool packed = false;                 // host mode, small: one copy each way through the pinned mirror (ensure_stage)
    size_t out_begin = 0, out_end = 0;
    A.table = h->table_set ? h->d_table : nullptr;
    A.cinf = h->kp.F > 0 ? h->d_cinf : nullptr;
    if (mem == IGT_MEM_DEVICE) {
        A.x0 = x0; A.u_prev = u_prev; A.kparams = kparams; A.flags = flags; A.obs = obs_xy;
        A.tv_sv = tv_sv; A.enc = enc; A.u_ws = u_ws;
        A.x_out = x_out; A.u_out = u_out; A.cost_out = cost_out; A.argmin_out = argmin_out; A.status_out = status_out;
    } else if (mem == IGT_MEM_HOST) {
        const size_t bytes = (n_x + 3 * n_u + n_k + n_obs + n_xo + 2 * n_uo + B) * sizeof(T) + (size_t)B * 12 + 24 * 256;
        if (int rc = ensure_stage(h, bytes)) return rc;
        Arena ar{(char*)h->d_stage, 0};
        // inputs first, outputs behind them: each group is one contiguous span of the arena
        T* dx0 = ar.take<T>(n_x); T* dup = ar.take<T>(n_u); T* dk = ar.take<T>(n_k);
        uint32_t* dfl = ar.take<uint32_t>(B); T* dob = ar.take<T>(n_obs ? n_obs : 1);
        T* dtv = ar.take<T>(n_u); T* den = ar.take<T>(n_u);
        T* dws = u_ws ? ar.take<T>(n_uo) : nullptr;
        const size_t in_span = ar.off;
        T* dxo = ar.take<T>(n_xo);
        out_begin = (size_t)((char*)dxo - (char*)h->d_stage);
        T* duo = ar.take<T>(n_uo); T* dco = ar.take<T>(B);
        int32_t* dam = ar.take<int32_t>(B); int32_t* dst = ar.take<int32_t>(B);
        out_end = ar.off;
        packed = bytes <= PACK_BYTES;
        if (packed) {
            char* hs = (char*)h->h_stage;
            auto put = [&](const void* src, const void* d, size_t nbytes) {
                std::memcpy(hs + ((const char*)d - (const char*)h->d_stage), src, nbytes);
            };
            put(x0, dx0, n_x * sizeof(T)); put(u_prev, dup, n_u * sizeof(T)); put(kparams, dk, n_k * sizeof(T));
            put(flags, dfl, (size_t)B * 4);
            if (n_obs) put(obs_xy, dob, n_obs * sizeof(T));
            if (value) { put(tv_sv, dtv, n_u * sizeof(T)); put(enc, den, n_u * sizeof(T)); }
            if (u_ws) put(u_ws, dws, n_uo * sizeof(T));
            HIPCHK(hipMemcpyAsync(h->d_stage, hs, in_span, hipMemcpyHostToDevice, st));
        } else {
            if (value) {
                HIPCHK(hipMemcpyAsync(dtv, tv_sv, n_u * sizeof(T), hipMemcpyHostToDevice, st));
                HIPCHK(hipMemcpyAsync(den, enc, n_u * sizeof(T), hipMemcpyHostToDevice, st));
            }
            if (u_ws) HIPCHK(hipMemcpyAsync(dws, u_ws, n_uo * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dx0, x0, n_x * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dup, u_prev, n_u * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dk, kparams, n_k * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dfl, flags, (size_t)B * 4, hipMemcpyHostToDevice, st));
            if (n_obs) HIPCHK(hipMemcpyAsync(dob, obs_xy, n_obs * sizeof(T), hipMemcpyHostToDevice, st));
        }
        A.tv_sv = dtv; A.enc = den;
        if (u_ws) A.u_ws = dws;
        A.x0 = dx0; A.u_prev = dup; A.kparams = dk; A.flags = dfl; A.obs = dob;
        A.x_out = dxo; A.u_out = duo; A.cost_out = dco; A.argmin_out = dam; A.status_out = dst;
    } else {
        return fail(IGT_E_INVALID, "mem must be IGT_MEM_DEVICE or IGT_MEM_HOST");
    }

    // workspace (grows on first use, never shrinks): per-slice partial arg-min [+ value-net records]
    // per-scenario partials: float value path reduces to ONE per scenario (compact list + atomicMin),
    // double value path keeps one per 64-candidate chunk, progress cost one per 128-candidate slice
    const bool compact = value && sizeof(T) == 4;
    // units per scenario of the search kernel: 128-candidate slices (float, two candidates per lane) or 64 (double)
    const size_t Wk = sizeof(T) == 4 ? ((size_t)p.C + 127) / 128 : (size_t)p.C / 64;
    const size_t W = compact ? 1 : (value ? (size_t)p.C / 64 : Wk);
    const bool exact64 = sizeof(T) == 8 && (h->kp.dev & 1024);      // developer switch: oracle-order double kernels
    // small float batches: emit in 4 pieces from checkpoints of the search pass.  Measured (search + emit): +15 % at
    // B = 1024, +12 % at 2048, +3 % at 4096, -1 % at 8192 -- but the records are ~150 MB of HBM writes per B = 4096 solve
    // against ~1 MB of algorithmic traffic, so they are only spent where emit latency dominates.
    int ck_parts = 1;
    if (sizeof(T) == 4 && B <= 2048 && p.N % 4 == 0 && p.N >= 8) ck_parts = 4;
    else if (sizeof(T) == 4 && B <= 2048 && p.N % 2 == 0 && p.N >= 4) ck_parts = 2;
    if (h->dev_ckpt >= 0) {
        const int v = h->dev_ckpt;
        if (sizeof(T) == 4 && v >= 1 && v <= igt::SEG_MAX_PARTS && p.N % v == 0 && p.N >= 2 * v) ck_parts = v;
    }
    const bool use_ckpt = ck_parts > 1;
    const size_t ckpt_bytes = use_ckpt ? (size_t)(ck_parts - 1) * B * Wk * igt::SEG_UNIT_DOUBLES * 8 + 256 : 0;
    // small double batches -- no more units than the chip has SIMDs, B <= 256 at 256 candidates -- : the search pass keeps
    // every trajectory and emit copies the winner's (igt_kernels_common.h CaptureSink; 97 KB of stores per unit at N = 20,
    // hidden behind a lone wave's dependent chains; with several units per SIMD they are not: a closed loop of 512
    // problems per step ran 0.32 ms per step with them against 0.30 without, 0.43 against 0.36 at 1024)
    size_t traj_max_units = (size_t)h->n_cu * 4;
    if (h->dev_traj_max >= 0) traj_max_units = (size_t)std::min(h->dev_traj_max, 8192) * Wk;   // sweeps: a batch size
    const bool capture = sizeof(T) == 8 && !exact64 && (size_t)B * Wk <= traj_max_units && (p.C % 64) == 0;
    const size_t traj_doubles = capture ? (size_t)B * Wk * igt::traj_unit_doubles(p.N) : 0;
    double* d_cpar = nullptr;
    {
        const size_t n_rec = value ? (size_t)B * p.C : 0;
        // double path, progress cost: the unit winners' horizon checkpoints for the emit in pieces (24 doubles per unit)
        const bool ck_ok = sizeof(T) == 8 && !value && !exact64 && p.N >= 8;
        const size_t ck_doubles = ck_ok ? (size_t)B * W * igt::CK_RECORD_DOUBLES : 0;
        // double path: the acceleration rows' travel sums [B, G] (accel_rows_kernel -> the tracking family's incumbent bound)
        const size_t rows_doubles = sizeof(T) == 8 ? (size_t)B * (p.cand_mode == IGT_CAND_TABLE ? 1 : isqrt_exact(p.C)) : 0;
        const size_t need = traj_doubles * 8 + 256 + (size_t)B * 16 + (ck_doubles + rows_doubles) * 8 + 256 + (size_t)B * W * 12 + (size_t)B * 56 + 8192 + ckpt_bytes + (size_t)(B + 8) * Wk * 36 + 256 + n_rec * (2 * sizeof(T) + 16) +
                            (value ? (size_t)B * igt::VN_H * sizeof(T) + (size_t)B * Wk * 8 + (size_t)B * 8 + n_rec * 4 + 512 : 0) + 20 * 256;
        if (int rc = ensure_work(h, need, st)) return rc;
        Arena wa{(char*)h->d_work, 0};
        A.part_J = wa.take<double>((size_t)B * W + (sizeof(T) == 8 ? (size_t)2 * B : (size_t)B) + rows_doubles + ck_doubles);  // double path: + [B] live-row masks + [B] incumbents + [B G] row sums + checkpoint records (igt_kernels_f64.hip); float path: + [B] incumbents
        A.ck_ok = ck_ok;
        A.part_c = wa.take<int32_t>((size_t)B * W);
        d_cpar = wa.take<double>((size_t)B * 4);
        const bool trace = (h->kp.dev & 256) != 0;      // developer trace: 32 B per unit behind the counters
        A.work_counter = wa.take<unsigned>(1024 + (trace ? (size_t)((B + 7) / 8) * 8 * Wk * 8 : 0));
        A.n_cu = h->n_cu;
        A.waves_per_simd = h->concurrency >= 3 ? 1 : 2;      // two solves in flight do not overlap on this runtime (igtmpc.h)
        // small batches: the search pass leaves horizon checkpoints, emit rolls the winner's four quarters at once
        A.ckpt = use_ckpt ? wa.take<double>((size_t)(ck_parts - 1) * B * Wk * igt::SEG_UNIT_DOUBLES) : nullptr;
        A.ck_parts = ck_parts;
        // small batches: the search queues are sorted longest unit first (build_queues_kernel; 3-5 % up to B = 4096,
        // nothing from 8192 on)
        A.queue_order = B <= 6144 ? wa.take<unsigned>((size_t)((B + 7) / 8) * 8 * Wk) : nullptr;
        if constexpr (sizeof(T) == 8) A.traj = capture ? wa.take<double>(traj_doubles) : nullptr;
        if constexpr (sizeof(T) == 8) A.row_mask = reinterpret_cast<unsigned long long*>(A.part_J + (size_t)B * W);
        if (value) {
            A.rec_J = wa.take<double>(n_rec);
            A.rec_sN = wa.take<T>(n_rec);
            A.rec_vN = wa.take<T>(n_rec);
            A.rec_viol = wa.take<uint32_t>(n_rec);
            A.p_vec = wa.take<T>((size_t)B * igt::VN_H);
            A.rec_b = wa.take<int32_t>(n_rec);
            A.rec_count = wa.take<unsigned>(64);
            A.best_key = wa.take<unsigned long long>((size_t)B);
            A.unit_seg = wa.take<int2>((size_t)B * Wk);
            if (sizeof(T) == 8) {
                A.prune_thr = wa.take<double>((size_t)B);
                A.live_idx = wa.take<unsigned>(n_rec);
            }
        }
    }
    if (h->prof) HIPCHK(hipEventRecord(h->ev[0], st));
    igt::KP kp = h->kp;
    for (int it = 0; it <= p.refine_iters; ++it) {
        kp.refine_it = it;
        A.cpar = it == 0 ? nullptr : d_cpar;
        if (value) {
            if (compact) {
                if (!counters_by_builder(kp, B, A)) HIPCHK(hipMemsetAsync(A.work_counter, 0, 8 * 256, st));
                HIPCHK(hipMemsetAsync(A.rec_count, 0, 256, st));
                HIPCHK(hipMemsetAsync(A.best_key, 0xff, (size_t)B * 8, st));
            }
            else if (!exact64) {      // double path: compact list of feasible candidates as well
                if (!counters_by_builder(kp, B, A)) HIPCHK(hipMemsetAsync(A.work_counter, 0, 8 * 256, st));
                HIPCHK(hipMemsetAsync(A.rec_count, 0, 256, st));
            }
            HIPCHK(igt::launch_search_records<T>(kp, B, A, st));
            HIPCHK(igt::launch_value<T>(kp, B, net_of<T>(h), A, nullptr, nullptr, st));
            if (exact64) HIPCHK(igt::launch_reduce<T>(B, (int)W, A, st));
        } else {
            if (!exact64 && !counters_by_builder(kp, B, A)) HIPCHK(hipMemsetAsync(A.work_counter, 0, 8 * 256, st));
            HIPCHK(igt::launch_search<T>(kp, B, A, h->nc, st));
        }
        if (it < p.refine_iters) {   // winner of this pass -> centre / span of the next one
            igt::SolveArgs<T> R = A;
            int Wr = (int)W;
            if (exact64 && !value) {   // oracle-order double kernels, progress cost: the search already reduced to one winner
                R.part_J = reinterpret_cast<double*>(A.cost_out);
                R.part_c = A.argmin_out;
                Wr = 1;
            }
            HIPCHK(igt::launch_refine<T>(kp, B, Wr, R, d_cpar, it == 0 ? 1 : 0, st));
        }
    }
    if (h->prof) HIPCHK(hipEventRecord(h->ev[1], st));
    HIPCHK(igt::launch_emit<T>(kp, B, (int)W, A, st));
    if (h->prof) { HIPCHK(hipEventRecord(h->ev[2], st)); h->ev_recorded = true; }

    if (h->kp.dev & 256) {      // developer trace -> $IGT_DEV_TRACE (binary u64[units][4])
        if (const char* path = std::getenv("IGT_DEV_TRACE")) {
            HIPCHK(hipStreamSynchronize(st));
            const size_t n = (size_t)((B + 7) / 8) * 8 * Wk * 4;
            std::vector<unsigned long long> tr(n);
            HIPCHK(hipMemcpy(tr.data(), reinterpret_cast<char*>(A.work_counter) + 4096, n * 8, hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(path, "wb")) { std::fwrite(tr.data(), 8, n, f); std::fclose(f); }
        }
    }
    if (mem == IGT_MEM_HOST && packed) {
        char* hs = (char*)h->h_stage;
        HIPCHK(hipMemcpyAsync(hs + out_begin, (char*)h->d_stage + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        auto get = [&](void* dst_, const void* d, size_t nbytes) {
            std::memcpy(dst_, hs + ((const char*)d - (const char*)h->d_stage), nbytes);
        };
        get(x_out, A.x_out, n_xo * sizeof(T)); get(u_out, A.u_out, n_uo * sizeof(T)); get(cost_out, A.cost_out, (size_t)B * sizeof(T));
        get(argmin_out, A.argmin_out, (size_t)B * 4); get(status_out, A.status_out, (size_t)B * 4);
    } else if (mem == IGT_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(x_out, A.x_out, n_xo * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(u_out, A.u_out, n_uo * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(cost_out, A.cost_out, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(argmin_out, A.argmin_out, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(status_out, A.status_out, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return IGT_OK;
}

template <typename T>
int rollout_impl(igt_handle* h, int32_t B, const T* x0, const T* u_prev, const T* kparams, const uint32_t* flags,
                 const T* obs_xy, const T* tv_sv, const T* enc, const T* u_ws, T* X_all, T* U_all, T* cost_all,
                 uint32_t* viol_all, int mem, void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (B < 0) return fail(IGT_E_INVALID, "B < 0");
    if (B == 0) return IGT_OK;
    if (!x0 || !u_prev || !kparams || !flags || !cost_all || !viol_all) return fail(IGT_E_INVALID, "null buffer");
    const igt_params& p = h->p;
    if (p.n_obs > 0 && !obs_xy) return fail(IGT_E_INVALID, "obs_xy is null but n_obs > 0");
    if (p.cand_mode == IGT_CAND_TABLE && !h->table_set) return fail(IGT_E_STATE, "candidate table not set");
    if (u_ws && p.cand_mode != IGT_CAND_RAMP_HOLD && p.cand_mode != IGT_CAND_TRACK)
        return fail(IGT_E_INVALID, "a warm start needs IGT_CAND_RAMP_HOLD or IGT_CAND_TRACK (the families whose targets are centred on it)");
    const bool value = p.cost_mode == IGT_COST_VALUE_NET;
    if (value) {
        if (!h->net_set) return fail(IGT_E_STATE, "value net not set (igt_set_value_net)");
        if (!tv_sv || !enc) return fail(IGT_E_INVALID, "tv_sv / enc required for the value-net cost");
    }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t n_x = (size_t)B * 7, n_u = (size_t)B * 2, n_k = (size_t)B * 3;
    const size_t n_obs = (size_t)B * p.n_obs * 2 * (p.N + 1);
    const size_t n_X = (size_t)B * p.C * 7 * (p.N + 1), n_U = (size_t)B * p.C * 2 * p.N, n_c = (size_t)B * p.C;
    igt::SolveArgs<T> A{};
    A.table = h->table_set ? h->d_table : nullptr;
    A.cinf = h->kp.F > 0 ? h->d_cinf : nullptr;
    T *dX = X_all, *dU = U_all, *dc = cost_all;
    uint32_t* dv = viol_all;
    if (mem == IGT_MEM_DEVICE) {
        A.x0 = x0; A.u_prev = u_prev; A.kparams = kparams; A.flags = flags; A.obs = obs_xy;
        A.tv_sv = tv_sv; A.enc = enc; A.u_ws = u_ws;
    } else if (mem == IGT_MEM_HOST) {
        const size_t n_ws = u_ws ? (size_t)B * 2 * p.N : 0;
        const size_t bytes = (n_x + 3 * n_u + n_k + n_obs + (X_all ? n_X : 0) + (U_all ? n_U : 0) + n_c + n_ws) * sizeof(T) +
                             n_c * 4 + (size_t)B * 4 + 24 * 256;
        if (int rc = ensure_stage(h, bytes)) return rc;
        Arena ar{(char*)h->d_stage, 0};
        T* dx0 = ar.take<T>(n_x); T* dup = ar.take<T>(n_u); T* dk = ar.take<T>(n_k);
        uint32_t* dfl = ar.take<uint32_t>(B); T* dob = ar.take<T>(n_obs ? n_obs : 1);
        dX = X_all ? ar.take<T>(n_X) : nullptr;
        dU = U_all ? ar.take<T>(n_U) : nullptr;
        dc = ar.take<T>(n_c); dv = ar.take<uint32_t>(n_c);
        T* dtv = ar.take<T>(n_u); T* den = ar.take<T>(n_u);
        if (value) {
            HIPCHK(hipMemcpyAsync(dtv, tv_sv, n_u * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(den, enc, n_u * sizeof(T), hipMemcpyHostToDevice, st));
        }
        A.tv_sv = dtv; A.enc = den;
        if (u_ws) {
            T* dws = ar.take<T>(n_ws);
            HIPCHK(hipMemcpyAsync(dws, u_ws, n_ws * sizeof(T), hipMemcpyHostToDevice, st));
            A.u_ws = dws;
        }
        HIPCHK(hipMemcpyAsync(dx0, x0, n_x * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dup, u_prev, n_u * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dk, kparams, n_k * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dfl, flags, (size_t)B * 4, hipMemcpyHostToDevice, st));
        if (n_obs) HIPCHK(hipMemcpyAsync(dob, obs_xy, n_obs * sizeof(T), hipMemcpyHostToDevice, st));
        A.x0 = dx0; A.u_prev = dup; A.kparams = dk; A.flags = dfl; A.obs = dob;
    } else {
        return fail(IGT_E_INVALID, "mem must be IGT_MEM_DEVICE or IGT_MEM_HOST");
    }
    if (value) {   // records -> value_kernel fills cost_all / viol_all with the terminal term included
        const size_t need = n_c * (2 * sizeof(T) + 12) + (size_t)B * igt::VN_H * sizeof(T) + 8 * 256;
        if (int rc = ensure_work(h, need, st)) return rc;
        Arena wa{(char*)h->d_work, 0};
        A.rec_J = wa.take<double>(n_c);
        A.rec_sN = wa.take<T>(n_c);
        A.rec_vN = wa.take<T>(n_c);
        A.rec_viol = wa.take<uint32_t>(n_c);
        A.p_vec = wa.take<T>((size_t)B * igt::VN_H);
    }
    HIPCHK(igt::launch_rollout_all<T>(h->kp, B, A, dX, dU, dc, dv, st));
    if (value) HIPCHK(igt::launch_value<T>(h->kp, B, net_of<T>(h), A, dc, dv, st));
    if (mem == IGT_MEM_HOST) {
        if (X_all) HIPCHK(hipMemcpyAsync(X_all, dX, n_X * sizeof(T), hipMemcpyDeviceToHost, st));
        if (U_all) HIPCHK(hipMemcpyAsync(U_all, dU, n_U * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(cost_all, dc, n_c * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(viol_all, dv, n_c * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return IGT_OK;
}

template <typename T>
int frenet_step_impl(igt_handle* h, int32_t n, const T* x, const T* u, const T* kparams, T* x_next, int mem,
                     void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (n < 0) return fail(IGT_E_INVALID, "negative size");
    if (n == 0) return IGT_OK;
    if (!x || !u || !kparams || !x_next) return fail(IGT_E_INVALID, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t n_x = (size_t)n * 7, n_u = (size_t)n * 2, n_k = (size_t)n * 3;
    const T *dx = x, *du = u, *dk = kparams;
    T* dout = x_next;
    if (mem == IGT_MEM_HOST) {
        if (int rc = ensure_stage(h, (2 * n_x + n_u + n_k) * sizeof(T) + 8 * 256)) return rc;
        Arena ar{(char*)h->d_stage, 0};
        T* a = ar.take<T>(n_x); T* b = ar.take<T>(n_u); T* c = ar.take<T>(n_k); dout = ar.take<T>(n_x);
        HIPCHK(hipMemcpyAsync(a, x, n_x * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(b, u, n_u * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c, kparams, n_k * sizeof(T), hipMemcpyHostToDevice, st));
        dx = a; du = b; dk = c;
    } else if (mem != IGT_MEM_DEVICE) {
        return fail(IGT_E_INVALID, "mem must be IGT_MEM_DEVICE or IGT_MEM_HOST");
    }
    HIPCHK(igt::launch_frenet_step<T>(h->kp, n, dx, du, dk, dout, st));
    if (mem == IGT_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(x_next, dout, n_x * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return IGT_OK;
}

template <typename T>
int forecast_impl(igt_handle* h, int32_t B, const T* ego_xyh, const T* opp, const T* opp_a, const int32_t* opp_route,
                  const T* plan_x, const T* plan_u, const int32_t* has_plan, T* obs_xy, T* tv_sv, int mem, void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (B < 0) return fail(IGT_E_INVALID, "B < 0");
    if (B == 0) return IGT_OK;
    if (!h->d_routes) return fail(IGT_E_STATE, "route table not set (igt_set_routes)");
    if (h->p.n_obs < 1) return fail(IGT_E_INVALID, "the forecast entry needs at least one other vehicle (n_obs >= 1)");
    if (!ego_xyh || !opp || !opp_a || !opp_route || !obs_xy || !tv_sv) return fail(IGT_E_INVALID, "null buffer");
    const bool plans = plan_x && plan_u && has_plan;
    if (!plans && (plan_x || plan_u || has_plan)) return fail(IGT_E_INVALID, "plan_x, plan_u, has_plan go together");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const int N = h->p.N;
    const size_t M1 = (size_t)h->p.n_obs;            // other vehicles per scene: every per-opponent array is [B, n_obs, ...]
    const size_t n_e = (size_t)B * 3, n_o = (size_t)B * M1 * 4, n_px = (size_t)B * M1 * 7 * (N + 1), n_pu = (size_t)B * M1 * 2 * N;
    const size_t n_out = (size_t)B * M1 * 2 * (N + 1), n_tv = (size_t)B * M1 * 2;
    const size_t n_pairs = (size_t)B * M1;
    const T *de = ego_xyh, *dop = opp, *da = opp_a, *dpx = plan_x, *dpu = plan_u;
    const int32_t *dr = opp_route, *dhp = has_plan;
    T *dout = obs_xy, *dtv = tv_sv;
    if (mem == IGT_MEM_HOST) {
        const size_t bytes = (n_e + n_o + n_pairs + (plans ? n_px + n_pu : 0) + n_out + n_tv) * sizeof(T) + n_pairs * 8 + 16 * 256;
        if (int rc = ensure_stage(h, bytes)) return rc;
        Arena ar{(char*)h->d_stage, 0};
        T* a0 = ar.take<T>(n_e); T* a1 = ar.take<T>(n_o); T* a2 = ar.take<T>(n_pairs);
        int32_t* a3 = ar.take<int32_t>(n_pairs); int32_t* a4 = ar.take<int32_t>(n_pairs);
        T* a5 = ar.take<T>(plans ? n_px : 1); T* a6 = ar.take<T>(plans ? n_pu : 1);
        dout = ar.take<T>(n_out); dtv = ar.take<T>(n_tv);
        HIPCHK(hipMemcpyAsync(a0, ego_xyh, n_e * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(a1, opp, n_o * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(a2, opp_a, n_pairs * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(a3, opp_route, n_pairs * 4, hipMemcpyHostToDevice, st));
        de = a0; dop = a1; da = a2; dr = a3;
        if (plans) {
            HIPCHK(hipMemcpyAsync(a4, has_plan, n_pairs * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(a5, plan_x, n_px * sizeof(T), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(a6, plan_u, n_pu * sizeof(T), hipMemcpyHostToDevice, st));
            dhp = a4; dpx = a5; dpu = a6;
        }
    } else if (mem != IGT_MEM_DEVICE) {
        return fail(IGT_E_INVALID, "mem must be IGT_MEM_DEVICE or IGT_MEM_HOST");
    }
    HIPCHK(igt::launch_forecast<T>(h->kp, B, h->d_routes, h->n_routes, de, dop, da, dr, plans ? dpx : nullptr,
                                   plans ? dpu : nullptr, plans ? dhp : nullptr, dout, dtv, st));
    if (mem == IGT_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(obs_xy, dout, n_out * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(tv_sv, dtv, n_tv * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return IGT_OK;
}

template <typename T>
int cartesian_impl(igt_handle* h, int32_t n, int32_t steps, const T* z0, const T* u, T* z_out, int mem, void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (n < 0 || steps < 0) return fail(IGT_E_INVALID, "negative size");
    if (n == 0) return IGT_OK;
    if (!z0 || !z_out || (steps > 0 && !u)) return fail(IGT_E_INVALID, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t n_z = (size_t)n * 4, n_u = (size_t)n * 2 * steps, n_o = (size_t)n * 4 * (steps + 1);
    const T *dz = z0, *du = u;
    T* dout = z_out;
    if (mem == IGT_MEM_HOST) {
        if (int rc = ensure_stage(h, (n_z + n_u + n_o) * sizeof(T) + 8 * 256)) return rc;
        Arena ar{(char*)h->d_stage, 0};
        T* a = ar.take<T>(n_z); T* b = ar.take<T>(n_u ? n_u : 1); dout = ar.take<T>(n_o);
        HIPCHK(hipMemcpyAsync(a, z0, n_z * sizeof(T), hipMemcpyHostToDevice, st));
        if (n_u) HIPCHK(hipMemcpyAsync(b, u, n_u * sizeof(T), hipMemcpyHostToDevice, st));
        dz = a; du = b;
    } else if (mem != IGT_MEM_DEVICE) {
        return fail(IGT_E_INVALID, "mem must be IGT_MEM_DEVICE or IGT_MEM_HOST");
    }
    HIPCHK(igt::launch_cartesian<T>(n, steps, h->p.dt, h->p.l_r, h->p.l_f, dz, du, dout, st));
    if (mem == IGT_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(z_out, dout, n_o * sizeof(T), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return IGT_OK;
}

template <typename T>
int allgather_impl(igt_handle* h, int32_t B_local, const T* u_out, T* u0_all, void* stream) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (B_local < 0) return fail(IGT_E_INVALID, "B_local < 0");
    if (!h->comm) {         // no communicator = one shard: the gathered vector is the local one
        if (B_local == 0) return IGT_OK;
        if (!u_out || !u0_all) return fail(IGT_E_INVALID, "null buffer");
        HIPCHK(hipSetDevice(h->device));
        hipStream_t st0 = stream ? (hipStream_t)stream : h->stream;
        HIPCHK(igt::launch_first_controls<T>(B_local, h->p.N, u_out, u0_all, st0));
        return IGT_OK;
    }
    // With a communicator every rank enters ncclAllGather with the SAME count or the peers hang / read garbage: an
    // empty shard is refused (pad it, as the header says), and the count of the first call is the communicator's.
    if (B_local == 0) return fail(IGT_E_INVALID, "B_local == 0 with a communicator: every rank must take part with the same padded B_local");
    if (h->comm_B_local == 0) h->comm_B_local = B_local;
    else if (h->comm_B_local != B_local)
        return fail(IGT_E_INVALID, "B_local changed since the communicator's first all-gather (all ranks must pass one padded shard size)");
    if (!u_out || !u0_all) return fail(IGT_E_INVALID, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t bytes = (size_t)B_local * 2 * sizeof(T);
    if (bytes > h->u0_bytes) {
        if (capturing(st)) return fail(IGT_E_STATE, "staging buffer too small for stream capture: run one eager call first");
        HIPCHK(hipStreamSynchronize(st));
        if (h->d_u0) { HIPCHK(hipFree(h->d_u0)); h->d_u0 = nullptr; h->u0_bytes = 0; }
        HIPCHK(hipMalloc(&h->d_u0, bytes + 256));
        h->u0_bytes = bytes;
    }
    HIPCHK(igt::launch_first_controls<T>(B_local, h->p.N, u_out, reinterpret_cast<T*>(h->d_u0), st));
    Rccl* r = rccl();
    if (int rc = r->AllGather(h->d_u0, u0_all, (size_t)B_local * 2, sizeof(T) == 4 ? 7 /* ncclFloat */ : 8 /* ncclDouble */,
                              h->comm, st))
        return rccl_fail(r, "ncclAllGather", rc);
    return IGT_OK;
}

}  // namespace

extern "C" {

int igt_version(void) { return IGT_VERSION; }

const char* igt_last_error(void) { return g_err.c_str(); }

int igt_params_default(igt_params* p) {
    if (!p) return fail(IGT_E_INVALID, "null params");
    std::memset(p, 0, sizeof(*p));
    p->N = 20; p->n_rk4 = 4; p->C = 256; p->n_obs = 1;
    p->cand_mode = IGT_CAND_LATTICE; p->cost_mode = IGT_COST_PROGRESS;
    p->dt = 0.1;
    p->l_r = 4.47 / 2; p->l_f = 4.47 / 2;                 /* mpc.py:48-50 */
    p->v_min = 0; p->v_max = 5; p->a_min = -4; p->a_max = 3; p->df_max = 1;  /* mpc.py:57-62 */
    p->jerk_limit = 0.9; p->steer_rate_limit = 0.7;       /* mpc.py:55-56 */
    p->ey_lim = 0.2;                                      /* mpc.py:61 */
    p->d_min = 2 * 2.8;                                   /* mpc.py:45, fourwayint.yaml:9 */
    p->w_u = 0.05;                                        /* mpc.py:362 */
    p->feas_tol = 1e-6;
    p->refine_iters = 0;
    p->track_ke = 0.3; p->track_span = 0.1; p->track_beta_lim = 0.7; p->track_env = 1.0; p->track_vcap = 1.0;
    return IGT_OK;
}

int igt_create(const igt_params* p, int device, igt_handle** out) {
    if (!p || !out) return fail(IGT_E_INVALID, "null argument");
    std::string why;
    if (validate(*p, why)) return fail(IGT_E_INVALID, why);
#if !IGT_DEV_KERNELS
    if (const char* e = getenv("IGT_DEV_FLAGS"))
        if (atoi(e) & IGT_DEV_KERNEL_FLAGS)
            return fail(IGT_E_INVALID, "IGT_DEV_FLAGS selects developer kernels (32 / 1024 / 2048) that this library is built without; "
                                       "load libigtmpc_dev.so (built with IGT_DEV_KERNELS=1)");
#endif
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(IGT_E_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    igt_handle* h = new (std::nothrow) igt_handle();
    if (!h) return fail(IGT_E_NOMEM, "out of host memory");
    h->p = *p;
    h->kp = make_kp(*p, 0);
    h->device = device;
    h->d_cinf = nullptr; h->d_table = nullptr; h->table_set = false; h->net_set = false;
    h->d_stage = nullptr; h->stage_bytes = 0; h->h_stage = nullptr;
    h->d_work = nullptr; h->work_bytes = 0;
    h->d_net = nullptr;
    h->d_routes = nullptr; h->n_routes = 0;
    h->prof = false; h->ev_recorded = false;
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0; h->comm_B_local = 0; h->d_u0 = nullptr; h->u0_bytes = 0;
    h->nc = 2;
    if (const char* e = std::getenv("IGT_NC")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4) h->nc = v;
    }
    while ((p->C / 64) % h->nc) h->nc /= 2;
    h->n_cu = 256;
    h->concurrency = 1;
    h->dev_ckpt = -1; h->dev_traj_max = -1;      // developer sweeps: the environment is read here, not on every solve
    if (const char* e = std::getenv("IGT_DEV_CKPT")) h->dev_ckpt = std::max(std::atoi(e), 0);
    if (const char* e = std::getenv("IGT_DEV_TRAJ_MAX")) h->dev_traj_max = std::max(std::atoi(e), 0);
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) h->n_cu = v; }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(IGT_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    for (int i = 0; i < 3; ++i) {
        e = hipEventCreate(&h->ev[i]);
        if (e != hipSuccess) { delete h; return fail(IGT_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    }
    e = igt::prepare_emit_kernels();
    if (e != hipSuccess) { (void)igt_destroy(h); return fail(IGT_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
    *out = h;
    return IGT_OK;
}

int igt_destroy(igt_handle* h) {
    if (!h) return IGT_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->d_cinf) (void)hipFree(h->d_cinf);
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->d_work) (void)hipFree(h->d_work);
    if (h->d_net) (void)hipFree(h->d_net);
    if (h->d_routes) (void)hipFree(h->d_routes);
    if (h->d_u0) (void)hipFree(h->d_u0);
    if (h->comm) { if (Rccl* r = rccl()) (void)r->CommDestroy(h->comm); h->comm = nullptr; rccl_release(); }
    for (int i = 0; i < 3; ++i) (void)hipEventDestroy(h->ev[i]);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return IGT_OK;
}

int igt_get_params(const igt_handle* h, igt_params* out) {
    if (!h || !out) return fail(IGT_E_INVALID, "null argument");
    *out = h->p;
    return IGT_OK;
}

int igt_set_cinf(igt_handle* h, const double* A, const double* b, int32_t F) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (F < 0 || F > IGT_MAX_CINF) return fail(IGT_E_INVALID, "F must be in [0, IGT_MAX_CINF]");
    if (F > 0 && (!A || !b)) return fail(IGT_E_INVALID, "null table");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_cinf) { HIPCHK(hipFree(h->d_cinf)); h->d_cinf = nullptr; }
    if (F > 0) {
        std::vector<double> packed((size_t)F * 3);
        for (int m = 0; m < F; ++m) {
            packed[m * 3 + 0] = A[m * 2 + 0];
            packed[m * 3 + 1] = A[m * 2 + 1];
            packed[m * 3 + 2] = b[m];
        }
        HIPCHK(hipMalloc((void**)&h->d_cinf, packed.size() * sizeof(double)));
        HIPCHK(hipMemcpy(h->d_cinf, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    h->kp.F = F;
    return IGT_OK;
}

int igt_set_candidate_table(igt_handle* h, const double* U) {
    if (!h || !U) return fail(IGT_E_INVALID, "null argument");
    if (h->p.cand_mode != IGT_CAND_TABLE) return fail(IGT_E_STATE, "handle was not created with IGT_CAND_TABLE");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->p.C * 2 * h->p.N;
    if (!h->d_table) HIPCHK(hipMalloc((void**)&h->d_table, n * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_table, U, n * sizeof(double), hipMemcpyHostToDevice));
    h->table_set = true;
    return IGT_OK;
}

int igt_set_value_net(igt_handle* h, int32_t n_layers, const int32_t* dims, const double* weights, const double* Wn,
                      const double* mu_f, double sigma_t, double mu_t) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (!dims || !weights || !Wn || !mu_f) return fail(IGT_E_INVALID, "null argument");
    if (n_layers != 3 && n_layers != 4) return fail(IGT_E_INVALID, "value net must have 2 or 3 hidden layers (3 or 4 Linear layers)");
    const int H = igt::VN_H;
    if (dims[0] != 6 || dims[n_layers] != 1) return fail(IGT_E_INVALID, "value net must map 6 -> 1");
    for (int l = 1; l < n_layers; ++l)
        if (dims[l] != H) return fail(IGT_E_INVALID, "hidden width must be 128");
    const int nm = n_layers - 2;   // hidden -> hidden matrices
    // unpack [W(out,in) row-major, b(out)] per layer
    const double* W1 = weights;                 // [H,6]
    const double* b1 = W1 + (size_t)H * 6;
    const double* Wh[2] = {nullptr, nullptr};
    const double* bh[2] = {nullptr, nullptr};
    const double* cur = b1 + H;
    for (int m = 0; m < nm; ++m) { Wh[m] = cur; bh[m] = cur + (size_t)H * H; cur = bh[m] + H; }
    const double* Wo = cur;                     // [1,H]
    const double bo = Wo[H];
    // host block: A1[H*6] c1[H] (WT[H*H] bias[H]) x nm, wout[H]
    const size_t n = (size_t)H * 6 + H + (size_t)nm * ((size_t)H * H + H) + H;
    std::vector<double> blk(n);
    double* A1 = blk.data();
    double* c1 = A1 + (size_t)H * 6;
    for (int i = 0; i < H; ++i) {
        double acc = b1[i];
        for (int k = 0; k < 6; ++k) {
            double a = 0.0;
            for (int j = 0; j < 6; ++j) a += W1[i * 6 + j] * Wn[j * 6 + k];     // A1 = W1 Wn
            A1[i * 6 + k] = a;
            acc -= a * mu_f[k];                                                 // c1 = b1 - A1 mu_f
        }
        c1[i] = acc;
    }
    double* q = c1 + H;
    size_t offWT[2] = {0, 0}, offB[2] = {0, 0};
    for (int m = 0; m < nm; ++m) {
        offWT[m] = (size_t)(q - blk.data());
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < H; ++j) q[(size_t)i * H + j] = Wh[m][(size_t)j * H + i];   // [i][j] = W[j][i]
        q += (size_t)H * H;
        offB[m] = (size_t)(q - blk.data());
        for (int j = 0; j < H; ++j) q[j] = bh[m][j];
        q += H;
    }
    const size_t offWo = (size_t)(q - blk.data());
    for (int j = 0; j < H; ++j) q[j] = Wo[j];
    // float block = the plain arrays followed by the MFMA fragment block (igt_value_net.h, value_mfma_kernel)
    const size_t nfrag = (size_t)igt::frag_floats(nm);
    std::vector<float> blkf(n + nfrag);
    for (size_t i = 0; i < n; ++i) blkf[i] = (float)blk[i];
    {
        float* f = blkf.data() + n;
        for (int t = 0; t < 4; ++t)                       // A1F: [A1 | c1 | 0] in A-operand order
            for (int s4 = 0; s4 < 4; ++s4)
                for (int l = 0; l < 64; ++l) {
                    const int i = 32 * t + (l & 31), k = 2 * s4 + (l >> 5);
                    *f++ = (float)(k < 6 ? A1[i * 6 + k] : (k == 6 ? c1[i] : 0.0));
                }
        for (int m = 0; m < nm; ++m)                      // WF[m]: W[j][i], i in the accumulator's row order
            for (int t = 0; t < 4; ++t)
                for (int ks = 0; ks < 64; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const int j = 32 * t + (l & 31), i = 32 * (ks >> 4) + igt::frag_row(ks & 15, l >> 5);
                        *f++ = (float)Wh[m][(size_t)j * H + i];
                    }
        for (int m = 0; m < nm; ++m)                      // BF[m]
            for (int t = 0; t < 4; ++t)
                for (int r = 0; r < 16; ++r)
                    for (int hh = 0; hh < 2; ++hh) *f++ = (float)bh[m][32 * t + igt::frag_row(r, hh)];
        for (int t = 0; t < 4; ++t)                       // WOF
            for (int r = 0; r < 16; ++r)
                for (int hh = 0; hh < 2; ++hh) *f++ = (float)Wo[32 * t + igt::frag_row(r, hh)];
    }
    // double block = the plain arrays followed by the f64 MFMA fragment block (igt_value_net.h, value_mfma_f64_kernel):
    // lane (i = l & 15, g = l >> 4) of k-step (T, r) holds W[16 To + i][16 T + 4 r + g]
    const size_t nfragd = (size_t)igt::fragd_doubles(nm);
    blk.resize(n + nfragd);
    A1 = blk.data(); c1 = A1 + (size_t)H * 6;            // (the resize may have moved the block)
    {
        double* f = blk.data() + n;
        for (int t = 0; t < 8; ++t)                       // A1F: [A1 | c1 | 0]
            for (int s2 = 0; s2 < 2; ++s2)
                for (int l = 0; l < 64; ++l) {
                    const int i = 16 * t + (l & 15), k = 4 * s2 + (l >> 4);
                    *f++ = k < 6 ? A1[i * 6 + k] : (k == 6 ? c1[i] : 0.0);
                }
        for (int m = 0; m < nm; ++m)                      // WF[m]
            for (int t = 0; t < 8; ++t)
                for (int ti = 0; ti < 8; ++ti)
                    for (int r = 0; r < 4; ++r)
                        for (int l = 0; l < 64; ++l)
                            *f++ = Wh[m][(size_t)(16 * t + (l & 15)) * H + (16 * ti + 4 * r + (l >> 4))];
        for (int m = 0; m < nm; ++m)                      // BF[m]: bias of row 16 t + g + 4 r
            for (int t = 0; t < 8; ++t)
                for (int r = 0; r < 4; ++r)
                    for (int g = 0; g < 4; ++g) *f++ = bh[m][16 * t + g + 4 * r];
        for (int t = 0; t < 8; ++t)                       // WOF
            for (int r = 0; r < 4; ++r)
                for (int g = 0; g < 4; ++g) *f++ = Wo[16 * t + g + 4 * r];
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_net) { HIPCHK(hipFree(h->d_net)); h->d_net = nullptr; }
    const size_t bytes_f = (((n + nfrag) * 4 + 255) / 256) * 256;
    HIPCHK(hipMalloc(&h->d_net, bytes_f + (n + nfragd) * 8));
    float* df = reinterpret_cast<float*>(h->d_net);
    double* dd = reinterpret_cast<double*>(reinterpret_cast<char*>(h->d_net) + bytes_f);
    HIPCHK(hipMemcpy(df, blkf.data(), (n + nfrag) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dd, blk.data(), (n + nfragd) * 8, hipMemcpyHostToDevice));
    auto fill = [&](auto& net, auto* base) {
        net.A1 = base; net.c1 = base + (size_t)H * 6;
        for (int m = 0; m < 2; ++m) { net.WT[m] = m < nm ? base + offWT[m] : nullptr; net.bias[m] = m < nm ? base + offB[m] : nullptr; }
        net.wout = base + offWo;
        net.n_hidden_mats = nm;
    };
    fill(h->net_f, df);
    fill(h->net_d, dd);
    h->net_f.frag = df + n; h->net_f.fragd = nullptr;
    h->net_d.frag = nullptr; h->net_d.fragd = dd + n;
    h->net_f.bout = (float)bo; h->net_f.sigma_t = (float)sigma_t; h->net_f.mu_t = (float)mu_t;
    h->net_d.bout = bo; h->net_d.sigma_t = sigma_t; h->net_d.mu_t = mu_t;
    HIPCHK(igt::prepare_value_kernels(nm));
    h->net_set = true;
    return IGT_OK;
}

int igt_solve_batch_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev, const float* kparams,
                        const uint32_t* flags, const float* obs_xy, const float* tv_sv, const float* enc, float* x_out,
                        float* u_out, float* cost_out, int32_t* argmin_out, int32_t* status_out, int mem, void* stream) {
    return solve_impl<float>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, nullptr, x_out, u_out, cost_out,
                             argmin_out, status_out, mem, stream);
}
int igt_solve_batch_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev, const double* kparams,
                        const uint32_t* flags, const double* obs_xy, const double* tv_sv, const double* enc,
                        double* x_out, double* u_out, double* cost_out, int32_t* argmin_out, int32_t* status_out,
                        int mem, void* stream) {
    return solve_impl<double>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, nullptr, x_out, u_out, cost_out,
                              argmin_out, status_out, mem, stream);
}
int igt_solve_batch_ws_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev, const float* kparams,
                           const uint32_t* flags, const float* obs_xy, const float* tv_sv, const float* enc,
                           const float* u_ws, float* x_out, float* u_out, float* cost_out, int32_t* argmin_out,
                           int32_t* status_out, int mem, void* stream) {
    return solve_impl<float>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws, x_out, u_out, cost_out,
                             argmin_out, status_out, mem, stream);
}
int igt_solve_batch_ws_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev, const double* kparams,
                           const uint32_t* flags, const double* obs_xy, const double* tv_sv, const double* enc,
                           const double* u_ws, double* x_out, double* u_out, double* cost_out, int32_t* argmin_out,
                           int32_t* status_out, int mem, void* stream) {
    return solve_impl<double>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws, x_out, u_out, cost_out,
                              argmin_out, status_out, mem, stream);
}

int igt_rollout_batch_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev, const float* kparams,
                          const uint32_t* flags, const float* obs_xy, const float* tv_sv, const float* enc,
                          float* X_all, float* U_all, float* cost_all, uint32_t* viol_all, int mem, void* stream) {
    return rollout_impl<float>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, nullptr, X_all, U_all, cost_all,
                               viol_all, mem, stream);
}
int igt_rollout_batch_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev, const double* kparams,
                          const uint32_t* flags, const double* obs_xy, const double* tv_sv, const double* enc,
                          double* X_all, double* U_all, double* cost_all, uint32_t* viol_all, int mem, void* stream) {
    return rollout_impl<double>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, nullptr, X_all, U_all, cost_all,
                                viol_all, mem, stream);
}
int igt_rollout_batch_ws_f32(igt_handle* h, int32_t B, const float* x0, const float* u_prev, const float* kparams,
                             const uint32_t* flags, const float* obs_xy, const float* tv_sv, const float* enc,
                             const float* u_ws, float* X_all, float* U_all, float* cost_all, uint32_t* viol_all, int mem,
                             void* stream) {
    return rollout_impl<float>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws, X_all, U_all, cost_all,
                               viol_all, mem, stream);
}
int igt_rollout_batch_ws_f64(igt_handle* h, int32_t B, const double* x0, const double* u_prev, const double* kparams,
                             const uint32_t* flags, const double* obs_xy, const double* tv_sv, const double* enc,
                             const double* u_ws, double* X_all, double* U_all, double* cost_all, uint32_t* viol_all,
                             int mem, void* stream) {
    return rollout_impl<double>(h, B, x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws, X_all, U_all, cost_all,
                                viol_all, mem, stream);
}

int igt_frenet_step_f32(igt_handle* h, int32_t n, const float* x, const float* u, const float* kparams, float* x_next,
                        int mem, void* stream) {
    return frenet_step_impl<float>(h, n, x, u, kparams, x_next, mem, stream);
}
int igt_frenet_step_f64(igt_handle* h, int32_t n, const double* x, const double* u, const double* kparams,
                        double* x_next, int mem, void* stream) {
    return frenet_step_impl<double>(h, n, x, u, kparams, x_next, mem, stream);
}

int igt_set_routes(igt_handle* h, int32_t n_routes, const double* table) {
    if (!h || !table) return fail(IGT_E_INVALID, "null argument");
    if (n_routes < 1 || n_routes > 64) return fail(IGT_E_INVALID, "n_routes must be in [1, 64]");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_routes) { HIPCHK(hipFree(h->d_routes)); h->d_routes = nullptr; h->n_routes = 0; }
    HIPCHK(hipMalloc((void**)&h->d_routes, (size_t)n_routes * 12 * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_routes, table, (size_t)n_routes * 12 * sizeof(double), hipMemcpyHostToDevice));
    h->n_routes = n_routes;
    return IGT_OK;
}

int igt_forecast_batch_f32(igt_handle* h, int32_t B, const float* ego_xyh, const float* opp, const float* opp_a,
                           const int32_t* opp_route, const float* plan_x, const float* plan_u, const int32_t* has_plan,
                           float* obs_xy, float* tv_sv, int mem, void* stream) {
    return forecast_impl<float>(h, B, ego_xyh, opp, opp_a, opp_route, plan_x, plan_u, has_plan, obs_xy, tv_sv, mem, stream);
}
int igt_forecast_batch_f64(igt_handle* h, int32_t B, const double* ego_xyh, const double* opp, const double* opp_a,
                           const int32_t* opp_route, const double* plan_x, const double* plan_u, const int32_t* has_plan,
                           double* obs_xy, double* tv_sv, int mem, void* stream) {
    return forecast_impl<double>(h, B, ego_xyh, opp, opp_a, opp_route, plan_x, plan_u, has_plan, obs_xy, tv_sv, mem, stream);
}

int igt_cartesian_euler_f32(igt_handle* h, int32_t n, int32_t T, const float* z0, const float* u, float* z_out,
                            int mem, void* stream) {
    return cartesian_impl<float>(h, n, T, z0, u, z_out, mem, stream);
}
int igt_cartesian_euler_f64(igt_handle* h, int32_t n, int32_t T, const double* z0, const double* u, double* z_out,
                            int mem, void* stream) {
    return cartesian_impl<double>(h, n, T, z0, u, z_out, mem, stream);
}

int igt_comm_unique_id(void* id_out) {
    if (!id_out) return fail(IGT_E_INVALID, "null argument");
    Rccl* r = rccl_acquire();
    if (!r) return fail(IGT_E_STATE, "RCCL (librccl.so.1) could not be loaded");
    IgtNcclId id;
    const int rc = r->GetUniqueId(&id);
    const int ret = rc ? rccl_fail(r, "ncclGetUniqueId", rc) : IGT_OK;
    rccl_release();
    if (ret) return ret;
    std::memcpy(id_out, id.internal, IGT_COMM_ID_BYTES);
    return IGT_OK;
}

int igt_comm_init(igt_handle* h, int32_t world, int32_t rank, const void* id) {
    if (!h || !id) return fail(IGT_E_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(IGT_E_INVALID, "need 0 <= rank < world");
    if (h->comm) return fail(IGT_E_STATE, "communicator already initialised (igt_comm_destroy first)");
    Rccl* r = rccl_acquire();          // held until igt_comm_destroy / igt_destroy
    if (!r) return fail(IGT_E_STATE, "RCCL (librccl.so.1) could not be loaded");
    if (hipError_t e = hipSetDevice(h->device)) { rccl_release(); return fail(IGT_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
    IgtNcclId nid;
    std::memcpy(nid.internal, id, IGT_COMM_ID_BYTES);
    void* comm = nullptr;
    if (int rc = r->CommInitRank(&comm, world, nid, rank)) { const int ret = rccl_fail(r, "ncclCommInitRank", rc); rccl_release(); return ret; }
    h->comm = comm; h->comm_world = world; h->comm_rank = rank; h->comm_B_local = 0;
    return IGT_OK;
}

int igt_comm_destroy(igt_handle* h) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (!h->comm) return IGT_OK;
    Rccl* r = rccl();
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));          // no all-gather of this handle is still in flight on its own stream
    const int rc = r ? r->CommDestroy(h->comm) : 0;
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0; h->comm_B_local = 0;
    const int ret = rc ? rccl_fail(r, "ncclCommDestroy", rc) : IGT_OK;
    rccl_release();                                   // the last communicator closes the binding
    return ret;
}


int igt_allgather_controls_f32(igt_handle* h, int32_t B_local, const float* u_out, float* u0_all, void* stream) {
    return allgather_impl<float>(h, B_local, u_out, u0_all, stream);
}
int igt_allgather_controls_f64(igt_handle* h, int32_t B_local, const double* u_out, double* u0_all, void* stream) {
    return allgather_impl<double>(h, B_local, u_out, u0_all, stream);
}

int igt_set_concurrency(igt_handle* h, int32_t solves_in_flight) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (solves_in_flight < 1 || solves_in_flight > 64) return fail(IGT_E_INVALID, "solves_in_flight must lie in [1, 64]");
    h->concurrency = solves_in_flight;
    return IGT_OK;
}

int igt_set_profiling(igt_handle* h, int enable) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    h->prof = enable != 0;
    h->ev_recorded = false;
    return IGT_OK;
}

int igt_get_kernel_ms(igt_handle* h, float* search_ms, float* emit_ms) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (!h->prof || !h->ev_recorded) return fail(IGT_E_STATE, "no profiled solve recorded");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev[2]));
    float a = 0, b = 0;
    HIPCHK(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIPCHK(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
    if (search_ms) *search_ms = a;
    if (emit_ms) *emit_ms = b;
    return IGT_OK;
}

int igt_algorithmic_bytes_per_solve(const igt_handle* h, int elem_size, int64_t* read_bytes, int64_t* write_bytes) {
    if (!h) return fail(IGT_E_INVALID, "null handle");
    if (elem_size != 4 && elem_size != 8) return fail(IGT_E_INVALID, "elem_size must be 4 or 8");
    const igt_params& p = h->p;
    int64_t r = (int64_t)(7 + 2 + 3) * elem_size + 4 + (int64_t)p.n_obs * 2 * (p.N + 1) * elem_size;
    if (p.cost_mode == IGT_COST_VALUE_NET) r += 4 * elem_size;
    const int64_t w = (int64_t)(7 * (p.N + 1) + 2 * p.N) * elem_size + elem_size + 4 + 4;
    if (read_bytes) *read_bytes = r;
    if (write_bytes) *write_bytes = w;
    return IGT_OK;
}

}  // extern "C"
