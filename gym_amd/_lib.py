"""ctypes binding of libgymx.so (include/gymx.h).

The library is the product: if it cannot be loaded this module raises, there
is no Python or CPU fallback behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GYMX_LIB selects another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("GYMX_LIB") or os.path.join(_HERE, "lib", "libgymx.so")

GX_OK, GX_ERR_INVALID, GX_ERR_HIP, GX_ERR_OOM, GX_ERR_UNSUPPORTED, GX_ERR_RCCL = range(6)

(ENV_CARTPOLE, ENV_MOUNTAINCAR, ENV_MOUNTAINCAR_CONT, ENV_PENDULUM, ENV_ACROBOT, ENV_FROZENLAKE, ENV_TAXI,
 ENV_CLIFFWALKING, ENV_PONG, ENV_DUMMY, ENV_BREAKOUT, ENV_SPACEINVADERS, ENV_BLACKJACK, ENV_TABLE) = range(14)
MODE_EXACT, MODE_FAST32, MODE_FP64 = 0, 1, 2
AUTORESET, RECORD_STATS, SLIPPERY, TAXI_FULL, ACROBOT_RK4, BLACKJACK_FULL = 1, 2, 4, 8, 16, 64
POLICY_DEVICE_ACTIONS, POLICY_RANDOM = 0, 1
OUT_LAST, OUT_ALL, OUT_NONE = 0, 1, 2
F32, F64, I32, U8 = range(4)


class GxDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("env_id", C.c_int32), ("num_envs", C.c_uint64), ("seed", C.c_uint64),
        ("env_index_base", C.c_uint64), ("mode", C.c_int32), ("flags", C.c_uint32), ("time_limit", C.c_int32),
        ("reserved", C.c_int32), ("map_rows", C.c_char_p),
        # GX_ENV_TABLE only
        ("table", C.c_void_p), ("table_len", C.c_uint64), ("n_states", C.c_int32), ("n_actions", C.c_int32),
        ("n_outcomes", C.c_int32), ("start_state", C.c_int32), ("step_limit", C.c_int32), ("reserved2", C.c_int32),
    ]


class GxTransition(C.Structure):
    """gx_transition: one entry of a caller-supplied table"""
    _fields_ = [("next_state", C.c_int32), ("reward", C.c_float), ("done", C.c_int32)]


DESC_SIZE_V1 = 56  # GX_DESC_SIZE_V1: the struct before the GX_ENV_TABLE fields


class GxInfo(C.Structure):
    _fields_ = [
        ("num_envs", C.c_uint64), ("env_id", C.c_int32), ("n_real", C.c_int32), ("real_dtype", C.c_int32),
        ("n_int", C.c_int32), ("obs_dim", C.c_int32), ("obs_dtype", C.c_int32), ("action_dtype", C.c_int32),
        ("action_n", C.c_int32), ("action_low", C.c_float), ("action_high", C.c_float), ("reward_dtype", C.c_int32),
        ("obs_is_state", C.c_int32), ("state_bytes", C.c_uint64), ("obs_bytes", C.c_uint64),
    ]


class GxView(C.Structure):
    _fields_ = [
        ("num_envs", C.c_uint64), ("stride", C.c_uint64), ("state_real", C.c_void_p), ("state_int", C.c_void_p),
        ("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("truncated", C.c_void_p),
    ]


class GxStats(C.Structure):
    _fields_ = [
        ("episodes", C.c_int64), ("sum_return", C.c_double), ("sum_length", C.c_int64),
        ("sum_return_sq", C.c_double), ("min_return", C.c_double), ("max_return", C.c_double),
    ]


class GxRows(C.Structure):
    """gx_rows: caller-owned device rows for gx_rows_step, the entry point that allocates nothing"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("env_id", C.c_int32), ("num_envs", C.c_uint64), ("stride", C.c_uint64),
        ("seed", C.c_uint64), ("env_index_base", C.c_uint64), ("tick", C.c_uint64), ("mode", C.c_int32),
        ("state_bounded", C.c_int32), ("state_real", C.c_void_p), ("steps", C.c_void_p), ("ep_return", C.c_void_p),
        ("actions", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("stats", C.c_void_p),
        ("error", C.c_void_p),
    ]


ROWS_MAX_ENVS = 1 << 24


class GymxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("gymx error %d: %s" % (code, message))
        self.code = code


# Every symbol include/gymx.h declares, with its signature.
_vp, _u64, _i = C.c_void_p, C.c_uint64, C.c_int
SIGNATURES = {
    "gx_version": (C.c_uint32, []),
    "gx_last_error": (C.c_char_p, []),
    "gx_device_count": (_i, [C.POINTER(_i)]),
    "gx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "gx_ctx_destroy": (_i, [_vp]),
    "gx_ctx_synchronize": (_i, [_vp]),
    "gx_ctx_stream": (_i, [_vp, C.POINTER(_vp)]),
    "gx_vec_create": (_i, [_vp, C.POINTER(GxDesc), C.POINTER(_vp)]),
    "gx_vec_destroy": (_i, [_vp]),
    "gx_vec_info": (_i, [_vp, C.POINTER(GxInfo)]),
    "gx_vec_reset": (_i, [_vp]),
    "gx_vec_get_state": (_i, [_vp, _vp, C.c_size_t]),
    "gx_vec_set_state": (_i, [_vp, _vp, C.c_size_t]),
    "gx_vec_set_rng_draws": (_i, [_vp, _vp, C.c_size_t]),
    "gx_vec_get_obs": (_i, [_vp, _vp, C.c_size_t]),
    "gx_vec_step_async": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gx_vec_wait": (_i, [_vp]),
    "gx_vec_step_device": (_i, [_vp, _vp, _vp]),
    "gx_vec_step_random": (_i, [_vp, _vp]),
    "gx_vec_device_view": (_i, [_vp, C.POINTER(GxView)]),
    "gx_vec_rollout": (_i, [_vp, _i, _i, _vp, _vp]),
    "gx_vec_rollout_out": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "gx_vec_sample_actions": (_i, [_vp, _u64, _vp, _vp]),
    "gx_vec_tick": (_i, [_vp, C.POINTER(_u64)]),
    "gx_vec_stats": (_i, [_vp, C.POINTER(GxStats)]),
    "gx_rows_step": (_i, [_vp, C.POINTER(GxRows), _i, _vp]),
    "gx_comm_unique_id": (_i, [_vp]),
    "gx_comm_init": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "gx_comm_destroy": (_i, [_vp]),
    "gx_stats_allreduce": (_i, [_vp, _vp, C.POINTER(GxStats)]),
}

_lib = None


def load():
    """Load libgymx.so; raises ImportError with the build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -m gym_amd.build` (needs hipcc); there is no fallback path" % LIB_PATH)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64, and a process
    # that loads the system one first and torch's second ends up with two runtimes (torch then
    # sees no GPU).  Importing torch first makes libgymx.so bind to the runtime torch loaded, so
    # device pointers and streams can be shared between the two.  Hosts without torch (the C#
    # binding) simply use the system runtime.
    if not os.environ.get("GYMX_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != GX_OK:
        raise GymxError(code, load().gx_last_error().decode("utf-8", "replace"))
