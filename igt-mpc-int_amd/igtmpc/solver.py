"""BatchSolver -- Python face of the C ABI (include/igtmpc.h).

One BatchSolver replaces, for a whole batch of (scenario, ego) problems, what the
reference does one problem at a time with MPC_Planner.update_initial_condition /
update_predictions / solve (mpc.py:241-294, 383-406).

Buffers may be numpy arrays (host mode: the library stages them through HBM and
synchronises) or torch tensors on the solver's GPU (device mode: pointers are
handed over as they are and the work is enqueued on the given / current stream)."""
import ctypes as ct

import numpy as np

from . import _lib as L

_DT = {'f32': np.float32, 'f64': np.float64}


def _is_torch(x):
    return hasattr(x, 'data_ptr') and hasattr(x, 'is_cuda')


class BatchSolver:
    def __init__(self, N=20, dt=0.1, n_rk4=4, C=256, n_obs=1, device=0, dtype='f32',
                 cand_mode='lattice', cost_mode='progress', **limits):
        self._h = None
        self.lib = L.load()          # the shipped library -- or libigtmpc_dev.so when IGT_DEV_FLAGS asks for developer kernels
        if dtype not in _DT:
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.dtype = dtype
        self.np_dtype = _DT[dtype]
        p = L.igt_params()
        self._check(self.lib.igt_params_default(ct.byref(p)))
        p.N, p.dt, p.n_rk4, p.C, p.n_obs = N, dt, n_rk4, C, n_obs
        p.cand_mode = {'lattice': L.IGT_CAND_LATTICE, 'table': L.IGT_CAND_TABLE,
                       'ramp_hold': L.IGT_CAND_RAMP_HOLD, 'track': L.IGT_CAND_TRACK}[cand_mode]
        p.cost_mode = {'progress': L.IGT_COST_PROGRESS, 'value_net': L.IGT_COST_VALUE_NET}[cost_mode]
        for k, v in limits.items():
            if not hasattr(p, k):
                raise TypeError(f'unknown parameter {k!r}')
            setattr(p, k, v)
        self.params = p
        self.device = device
        h = ct.c_void_p()
        self._check(self.lib.igt_create(ct.byref(p), device, ct.byref(h)))
        self._h = h
        self.N, self.C, self.n_obs = N, C, n_obs
        self._solve = getattr(self.lib, f'igt_solve_batch_ws_{dtype}')
        self._rollout = getattr(self.lib, f'igt_rollout_batch_ws_{dtype}')
        self._cart = getattr(self.lib, f'igt_cartesian_euler_{dtype}')
        self._fstep = getattr(self.lib, f'igt_frenet_step_{dtype}')
        self._fcast = getattr(self.lib, f'igt_forecast_batch_{dtype}')
        self._routes_set = False

    def _check(self, rc):
        L.check(rc, self.lib)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if self._h is not None:
            self.lib.igt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ tables
    def set_cinf(self, A, b):
        """C_inf half-planes A[F,2] (v,a) <= b[F]   (mpc.py:88-104, 177-180)."""
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1, 2)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        if len(A) != len(b):
            raise ValueError('A and b disagree')
        self._check(self.lib.igt_set_cinf(self._h, A.ctypes.data, b.ctypes.data, len(b)))

    def set_candidate_table(self, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        if U.shape != (self.C, 2, self.N):
            raise ValueError(f'candidate table must be [{self.C},2,{self.N}]')
        self._check(self.lib.igt_set_candidate_table(self._h, U.ctypes.data))

    def set_value_net(self, layers, Wn=None, mu_f=None, sigma_t=1.0, mu_t=0.0):
        """Terminal value network of the gt_mpc cost (mpc.py:108-127, 367-369; model.py:14-51).
        layers = [(W[out,in], b[out]), ...] (3 or 4 Linear layers, 6 -> 128 -> ... -> 1);
        Wn[6,6], mu_f[6]: input whitening x -> Wn (x - mu_f); sigma_t, mu_t: target de-normalisation.
        The reference's statistics come from a dataset that is not shipped: identity by default."""
        Wn = np.eye(6) if Wn is None else np.asarray(Wn, dtype=np.float64)
        mu_f = np.zeros(6) if mu_f is None else np.asarray(mu_f, dtype=np.float64)
        dims = [int(np.asarray(layers[0][0]).shape[1])] + [int(np.asarray(W).shape[0]) for W, _ in layers]
        flat = np.concatenate([np.concatenate([np.asarray(W, np.float64).ravel(), np.asarray(b, np.float64).ravel()])
                               for W, b in layers])
        dims_a = np.asarray(dims, dtype=np.int32)
        Wn_c, mu_c = np.ascontiguousarray(Wn), np.ascontiguousarray(mu_f)
        self._check(self.lib.igt_set_value_net(self._h, len(layers), dims_a.ctypes.data, flat.ctypes.data,
                                           Wn_c.ctypes.data, mu_c.ctypes.data, float(sigma_t), float(mu_t)))

    def set_concurrency(self, solves_in_flight):
        """Solves the caller keeps in flight on the device at a time, on other handles / streams (igtmpc.h igt_set_concurrency):
        from 3 on, the persistent search kernels take one wave per SIMD instead of two so that two of them run side by side."""
        self._check(self.lib.igt_set_concurrency(self._h, int(solves_in_flight)))

    def set_profiling(self, on=True):
        self._check(self.lib.igt_set_profiling(self._h, int(on)))

    def kernel_ms(self):
        a, b = ct.c_float(), ct.c_float()
        self._check(self.lib.igt_get_kernel_ms(self._h, ct.byref(a), ct.byref(b)))
        return a.value, b.value

    def algorithmic_bytes_per_solve(self):
        r, w = ct.c_int64(), ct.c_int64()
        es = 4 if self.dtype == 'f32' else 8
        self._check(self.lib.igt_algorithmic_bytes_per_solve(self._h, es, ct.byref(r), ct.byref(w)))
        return r.value, w.value

    # ------------------------------------------------------------------ marshalling
    def _prep(self, arrs, shapes, dtypes):
        """-> (mode, ptrs, keepalive).  All numpy (host) or all torch-cuda (device)."""
        torch_mode = any(_is_torch(a) for a in arrs if a is not None)
        ptrs, keep = [], []
        for a, shp, dt in zip(arrs, shapes, dtypes):
            if a is None:
                ptrs.append(None)
                continue
            if torch_mode:
                import torch
                if not _is_torch(a) or not a.is_cuda:
                    raise TypeError('device mode needs every buffer to be a CUDA torch tensor')
                want = {np.float32: torch.float32, np.float64: torch.float64, np.uint32: torch.int32,
                        np.int32: torch.int32}[dt]
                if a.dtype != want and not (dt is np.uint32 and a.dtype in (torch.int32, torch.uint32)):
                    raise TypeError(f'tensor dtype {a.dtype} does not match {dt.__name__}')
                if a.device.index != self.device:
                    raise ValueError('tensor is on another device')
                if not a.is_contiguous():
                    raise ValueError('device buffers must be contiguous')
                if tuple(a.shape) != tuple(shp):
                    raise ValueError(f'shape {tuple(a.shape)} != {tuple(shp)}')
                ptrs.append(a.data_ptr())
                keep.append(a)
            else:
                b = np.ascontiguousarray(a, dtype=dt)
                if b.shape != tuple(shp):
                    raise ValueError(f'shape {b.shape} != {tuple(shp)}')
                ptrs.append(b.ctypes.data)
                keep.append(b)
        return (L.IGT_MEM_DEVICE if torch_mode else L.IGT_MEM_HOST), ptrs, keep

    def _stream_ptr(self, stream, torch_mode):
        """hipStream_t for the C ABI.  Device mode: the given stream, else torch's current stream.  torch's default
        stream has handle 0, which the C ABI reads as "the handle's own non-blocking stream" -- the kernels would
        then be unordered with the torch ops that produce their inputs and consume their outputs.  So in device
        mode ANY resolved handle of 0 (implicit, `stream=torch.cuda.default_stream()`, `stream=0`) names the legacy
        default stream explicitly (hipStreamLegacy == (hipStream_t)1).  Host mode keeps NULL: the library stages
        and synchronises on its own stream."""
        if stream is not None:
            ptr = int(getattr(stream, 'cuda_stream', stream) or 0)
        elif torch_mode:
            import torch
            ptr = int(torch.cuda.current_stream(self.device).cuda_stream or 0)
        else:
            return None
        if torch_mode and ptr == 0:
            ptr = 1
        return ptr or None

    # ------------------------------------------------------------------ the solve
    def solve(self, x0, u_prev, kparams, flags, obs_xy=None, tv_sv=None, enc=None, out=None, stream=None, u_ws=None):
        """x0[B,7] u_prev[B,2] kparams[B,3] flags[B] obs_xy[B,n_obs,2,N+1]
        -> dict(x[B,7,N+1], u[B,2,N], cost[B], argmin[B], status[B]).
        u_ws[B,2,N]: warm start (previous solution shifted by one step, utils.py:354-363), read where
        flags & IGT_FLAG_WARM; ramp-hold candidates are then centred on it instead of on u_prev."""
        B = int(x0.shape[0])
        dt, N, no = self.np_dtype, self.N, self.n_obs
        torch_mode = _is_torch(x0)
        if out is None:
            if torch_mode:
                import torch
                td = torch.float32 if self.dtype == 'f32' else torch.float64
                dev = x0.device
                out = dict(x=torch.empty((B, 7, N + 1), dtype=td, device=dev),
                           u=torch.empty((B, 2, N), dtype=td, device=dev),
                           cost=torch.empty((B,), dtype=td, device=dev),
                           argmin=torch.empty((B,), dtype=torch.int32, device=dev),
                           status=torch.empty((B,), dtype=torch.int32, device=dev))
            else:
                out = dict(x=np.empty((B, 7, N + 1), dt), u=np.empty((B, 2, N), dt), cost=np.empty((B,), dt),
                           argmin=np.empty((B,), np.int32), status=np.empty((B,), np.int32))
        arrs = [x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws,
                out['x'], out['u'], out['cost'], out['argmin'], out['status']]
        shapes = [(B, 7), (B, 2), (B, 3), (B,), (B, no, 2, N + 1), (B, 2), (B, 2), (B, 2, N),
                  (B, 7, N + 1), (B, 2, N), (B,), (B,), (B,)]
        dts = [dt, dt, dt, np.uint32, dt, dt, dt, dt, dt, dt, dt, np.int32, np.int32]
        mode, ptrs, keep = self._prep(arrs, shapes, dts)
        if mode == L.IGT_MEM_HOST:
            for k, i in (('x', 8), ('u', 9), ('cost', 10), ('argmin', 11), ('status', 12)):
                if keep_is_copy(out[k], ptrs[i]):
                    raise ValueError(f'out[{k!r}] must be a contiguous array of the solver dtype')
        self._check(self._solve(self._h, B, *ptrs, mode, self._stream_ptr(stream, mode == L.IGT_MEM_DEVICE)))
        return out

    def rollout_all(self, x0, u_prev, kparams, flags, obs_xy=None, tv_sv=None, enc=None, want_X=True, want_U=True,
                    stream=None, u_ws=None):
        """Every candidate of every scenario (parity/debug):
        -> dict(X[B,C,7,N+1] | None, U[B,C,2,N] | None, cost[B,C], viol[B,C])."""
        B = int(x0.shape[0])
        dt, N, no, Cn = self.np_dtype, self.N, self.n_obs, self.C
        if _is_torch(x0):
            raise TypeError('rollout_all is a host-mode debug entry (numpy buffers)')
        X = np.empty((B, Cn, 7, N + 1), dt) if want_X else None
        U = np.empty((B, Cn, 2, N), dt) if want_U else None
        cost = np.empty((B, Cn), dt)
        viol = np.empty((B, Cn), np.uint32)
        arrs = [x0, u_prev, kparams, flags, obs_xy, tv_sv, enc, u_ws, X, U, cost, viol]
        shapes = [(B, 7), (B, 2), (B, 3), (B,), (B, no, 2, N + 1), (B, 2), (B, 2), (B, 2, N),
                  (B, Cn, 7, N + 1), (B, Cn, 2, N), (B, Cn), (B, Cn)]
        dts = [dt, dt, dt, np.uint32, dt, dt, dt, dt, dt, dt, dt, np.uint32]
        mode, ptrs, keep = self._prep(arrs, shapes, dts)
        self._check(self._rollout(self._h, B, *ptrs, mode, self._stream_ptr(stream, False)))
        return dict(X=X, U=U, cost=cost, viol=viol)

    def set_routes(self, table=None):
        """Route geometry for forecast(); default: the four-way intersection of igtmpc.routes."""
        if table is None:
            from . import routes as R
            T = R.TABLES
            table = np.column_stack([T['p0'], T['t'], T['c'], T['b0'], T['b1'], T['R'], T['end'],
                                     (T['turn'] == 0).astype(np.float64)])
        table = np.ascontiguousarray(table, dtype=np.float64)
        if table.ndim != 2 or table.shape[1] != 12:
            raise ValueError('route table must be [n_routes, 12]')
        self._check(self.lib.igt_set_routes(self._h, len(table), table.ctypes.data))
        self._routes_set = True

    def forecast(self, ego_xyh, opp, opp_a, opp_route, plan_x=None, plan_u=None, has_plan=None, stream=None):
        """Opponent forecast + V2V sharing + filter_preds on the device (constant_acceleration_model.py:18-82,
        utils.py:339-352, 365-388): -> (obs_xy[B,n_obs,2,N+1], tv_sv[B,2]).
        Two-vehicle scenes (n_obs = 1): opp[B,4], opp_a[B], opp_route[B], plan_x[B,7,N+1], plan_u[B,2,N], has_plan[B].
        M > 2 (mpc.py:82-83 takes any M): the same arrays with an n_obs axis behind B -- opp[B,n_obs,4], opp_a[B,n_obs], ... -- and
        tv_sv[B,n_obs,2] (the last raw prediction of every other vehicle, mpc.py:263-276)."""
        if not self._routes_set:
            self.set_routes()
        B, N, M1 = int(ego_xyh.shape[0]), self.N, self.n_obs
        dt = self.np_dtype
        tv_shape = (B, 2) if M1 == 1 else (B, M1, 2)
        if _is_torch(ego_xyh):
            import torch
            obs = torch.empty((B, M1, 2, N + 1), dtype=ego_xyh.dtype, device=ego_xyh.device)
            tv = torch.empty(tv_shape, dtype=ego_xyh.dtype, device=ego_xyh.device)
        else:
            obs, tv = np.empty((B, M1, 2, N + 1), dt), np.empty(tv_shape, dt)
        ax = () if M1 == 1 else (M1,)
        arrs = [ego_xyh, opp, opp_a, opp_route, plan_x, plan_u, has_plan, obs, tv]
        shapes = [(B, 3), (B, *ax, 4), (B, *ax), (B, *ax), (B, *ax, 7, N + 1), (B, *ax, 2, N), (B, *ax), (B, M1, 2, N + 1), tv_shape]
        dts = [dt, dt, dt, np.int32, dt, dt, np.int32, dt, dt]
        mode, ptrs, keep = self._prep(arrs, shapes, dts)
        self._check(self._fcast(self._h, B, *ptrs, mode, self._stream_ptr(stream, mode == L.IGT_MEM_DEVICE)))
        return obs, tv

    def frenet_step(self, x, u, kparams, stream=None):
        """One control step of the RK4 Frenet model for n states: x[n,7], u[n,2], kparams[n,3]
        -> x_next[n,7]   (kinematic_bicycle_model_frenet.py:70-127)."""
        n = int(x.shape[0])
        dt = self.np_dtype
        if _is_torch(x):
            import torch
            out = torch.empty((n, 7), dtype=x.dtype, device=x.device)
        else:
            out = np.empty((n, 7), dt)
        mode, ptrs, keep = self._prep([x, u, kparams, out], [(n, 7), (n, 2), (n, 3), (n, 7)], [dt] * 4)
        self._check(self._fstep(self._h, n, *ptrs, mode, self._stream_ptr(stream, mode == L.IGT_MEM_DEVICE)))
        return out

    def cartesian_euler(self, z0, u, stream=None):
        """z0[n,4]=(x,y,psi,v), u[n,2,T] -> z[n,4,T+1]   (kinematic_bicycle_model.py:15-50)."""
        n, T = int(z0.shape[0]), int(u.shape[2])
        dt = self.np_dtype
        if _is_torch(z0):
            import torch
            out = torch.empty((n, 4, T + 1), dtype=z0.dtype, device=z0.device)
        else:
            out = np.empty((n, 4, T + 1), dt)
        mode, ptrs, keep = self._prep([z0, u, out], [(n, 4), (n, 2, T), (n, 4, T + 1)], [dt, dt, dt])
        self._check(self._cart(self._h, n, T, *ptrs, mode, self._stream_ptr(stream, mode == L.IGT_MEM_DEVICE)))
        return out


def keep_is_copy(arr, ptr):
    """True when np.ascontiguousarray had to copy an OUTPUT buffer (results would be lost)."""
    return isinstance(arr, np.ndarray) and arr.ctypes.data != ptr
