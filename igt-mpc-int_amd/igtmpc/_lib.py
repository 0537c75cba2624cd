"""ctypes binding of include/igtmpc.h (libigtmpc.so, built by csrc/Makefile).

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing a solver raises ImportError with the build command."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libigtmpc.so')

IGT_MEM_DEVICE, IGT_MEM_HOST = 0, 1
IGT_CAND_LATTICE, IGT_CAND_TABLE, IGT_CAND_RAMP_HOLD, IGT_CAND_TRACK = 0, 1, 2, 3
IGT_COST_PROGRESS, IGT_COST_VALUE_NET = 0, 1
IGT_FLAG_ABS_HEADING = 1
IGT_FLAG_WARM = 2
VIOL_BITS = dict(box_v=1, box_u=2, rate=4, ey=8, terminal=16, collision=32, nonfinite=64)


class igt_params(C.Structure):
    _fields_ = [('N', C.c_int32), ('n_rk4', C.c_int32), ('C', C.c_int32), ('n_obs', C.c_int32),
                ('cand_mode', C.c_int32), ('cost_mode', C.c_int32),
                ('dt', C.c_double), ('l_r', C.c_double), ('l_f', C.c_double),
                ('v_min', C.c_double), ('v_max', C.c_double), ('a_min', C.c_double), ('a_max', C.c_double),
                ('df_max', C.c_double), ('jerk_limit', C.c_double), ('steer_rate_limit', C.c_double),
                ('ey_lim', C.c_double), ('d_min', C.c_double), ('w_u', C.c_double), ('feas_tol', C.c_double),
                ('refine_iters', C.c_int32), ('reserved', C.c_int32),
                ('track_ke', C.c_double), ('track_span', C.c_double), ('track_beta_lim', C.c_double),
                ('track_env', C.c_double), ('track_vcap', C.c_double)]


# every symbol include/igtmpc.h declares: name -> (restype, argtypes)
_vp, _i32, _i = C.c_void_p, C.c_int32, C.c_int
_SOLVE = [_vp, _i32] + [_vp] * 12 + [_i, _vp]
_ROLL = [_vp, _i32] + [_vp] * 11 + [_i, _vp]
_SOLVE_WS = [_vp, _i32] + [_vp] * 13 + [_i, _vp]
_ROLL_WS = [_vp, _i32] + [_vp] * 12 + [_i, _vp]
_CART = [_vp, _i32, _i32, _vp, _vp, _vp, _i, _vp]
_FSTEP = [_vp, _i32, _vp, _vp, _vp, _vp, _i, _vp]
_FCAST = [_vp, _i32] + [_vp] * 9 + [_i, _vp]
SYMBOLS = {
    'igt_version': (_i, []),
    'igt_last_error': (C.c_char_p, []),
    'igt_params_default': (_i, [C.POINTER(igt_params)]),
    'igt_create': (_i, [C.POINTER(igt_params), _i, C.POINTER(_vp)]),
    'igt_destroy': (_i, [_vp]),
    'igt_get_params': (_i, [_vp, C.POINTER(igt_params)]),
    'igt_set_cinf': (_i, [_vp, _vp, _vp, _i32]),
    'igt_set_candidate_table': (_i, [_vp, _vp]),
    'igt_set_value_net': (_i, [_vp, _i32, _vp, _vp, _vp, _vp, C.c_double, C.c_double]),
    'igt_solve_batch_f32': (_i, _SOLVE),
    'igt_solve_batch_f64': (_i, _SOLVE),
    'igt_solve_batch_ws_f32': (_i, _SOLVE_WS),
    'igt_solve_batch_ws_f64': (_i, _SOLVE_WS),
    'igt_rollout_batch_f32': (_i, _ROLL),
    'igt_rollout_batch_f64': (_i, _ROLL),
    'igt_rollout_batch_ws_f32': (_i, _ROLL_WS),
    'igt_rollout_batch_ws_f64': (_i, _ROLL_WS),
    'igt_set_routes': (_i, [_vp, _i32, _vp]),
    'igt_forecast_batch_f32': (_i, _FCAST),
    'igt_forecast_batch_f64': (_i, _FCAST),
    'igt_frenet_step_f32': (_i, _FSTEP),
    'igt_frenet_step_f64': (_i, _FSTEP),
    'igt_cartesian_euler_f32': (_i, _CART),
    'igt_cartesian_euler_f64': (_i, _CART),
    'igt_comm_unique_id': (_i, [_vp]),
    'igt_comm_init': (_i, [_vp, _i32, _i32, _vp]),
    'igt_comm_destroy': (_i, [_vp]),
    'igt_allgather_controls_f32': (_i, [_vp, _i32, _vp, _vp, _vp]),
    'igt_allgather_controls_f64': (_i, [_vp, _i32, _vp, _vp, _vp]),
    'igt_set_concurrency': (_i, [_vp, _i32]),
    'igt_set_profiling': (_i, [_vp, _i]),
    'igt_get_kernel_ms': (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'igt_algorithmic_bytes_per_solve': (_i, [_vp, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_libs = {}
DEV_KERNEL_FLAGS = 32 | 1024 | 2048      # csrc/igt_device.h IGT_DEV_KERNEL_FLAGS
LIB_PATH_DEV = os.path.join(os.path.dirname(LIB_PATH), 'libigtmpc_dev.so')


def wants_dev_kernels():
    """True when IGT_DEV_FLAGS (read by the library at igt_create) selects kernels only libigtmpc_dev.so carries."""
    try:
        return bool(int(os.environ.get('IGT_DEV_FLAGS', '0') or 0) & DEV_KERNEL_FLAGS)
    except ValueError:
        return False


def load(dev=None):
    """Loads libigtmpc.so (the shipped library) once and sets every prototype.  Raises ImportError when the library is
    absent -- there is deliberately no fallback path.  dev=True, or dev=None with IGT_DEV_FLAGS asking for developer
    kernels: libigtmpc_dev.so, the same sources built with IGT_DEV_KERNELS=1 (tests and A/B tools only)."""
    if dev is None:
        dev = wants_dev_kernels()
    dev = bool(dev)
    if dev in _libs:
        return _libs[dev]
    path = LIB_PATH_DEV if dev else LIB_PATH
    if not dev and os.environ.get('IGT_LIB_PATH'):      # A/B runs against another build of the same ABI (tools/ab_lib.sh)
        path = os.environ['IGT_LIB_PATH']
    if not os.path.exists(path):
        raise ImportError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'or `make -C igt-mpc-int_amd/csrc`')
    try:
        # torch ships its own libamdhip64; it must be the first HIP runtime the process
        # maps, otherwise a later `import torch` finds "No HIP GPUs" (two runtimes).
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise ImportError(f'cannot load {path}: {e}') from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _libs[dev] = lib
    return lib


class IgtError(RuntimeError):
    pass


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load(False)).igt_last_error()
        raise IgtError(f'igtmpc error {rc}: {msg.decode() if msg else "?"}')
