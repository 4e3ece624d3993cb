"""ctypes binding of libazamd.so (include/azamd.h).

The library is the product: if it is missing, or no MI355X is visible, every compute call raises.
There is no CPU path behind this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZAMD_LIB") or os.path.join(HERE, "libazamd.so")  # AZAMD_LIB: diagnostic builds

GAME_C4, GAME_TTT, GAME_NBTTT = 0, 1, 2
EVAL_UNIFORM, EVAL_HASH, EVAL_CNN, EVAL_HOST, EVAL_ROLLOUT, EVAL_ROLLOUT_HASH = 0, 1, 2, 3, 4, 5
FLIP_NEVER, FLIP_STREAM, FLIP_RANDOM = 0, 1, 2
MOVE_FIRST_MAX, MOVE_SAMPLE = 0, 1
K3_KINDS = {None: 0, "default": 0, "wino": 1, "split": 2, "direct": 3}
GAME_IDS = {"Connect4": GAME_C4, "TTT": GAME_TTT, "NBTTT": GAME_NBTTT}


class AzError(RuntimeError):
    pass


class az_config(C.Structure):
    _fields_ = [
        ("game", C.c_int32), ("device", C.c_int32), ("n_games", C.c_int32), ("n_playout", C.c_int32),
        ("c_init", C.c_double), ("c_base", C.c_int32), ("evaluator", C.c_int32), ("eval_seed", C.c_uint64),
        ("flip_mode", C.c_int32), ("reuse_tree", C.c_int32), ("dirichlet_alpha", C.c_double),
        ("rng_seed", C.c_uint64), ("move_mode", C.c_int32), ("max_plies", C.c_int32),
        ("nodes_per_game", C.c_int64), ("game_id_base", C.c_uint64), ("n_cohorts", C.c_int32),
        ("game_id_stride", C.c_uint64), ("eval_cache_entries", C.c_int64),
    ]


class az_stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("playouts", "levels", "leaf_evals", "terminal_hits", "children",
                                         "nodes_used", "games_finished", "plies", "rollout_plies", "cache_hits")]


_P = C.POINTER
_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double

# name -> (restype, argtypes).  Mirrors include/azamd.h one to one; tests check every symbol is exported.
SIGNATURES = {
    "az_last_error": (C.c_char_p, []),
    "az_version": (C.c_int, []),
    "az_state_words": (C.c_int, [C.c_int]),
    "az_n_actions": (C.c_int, [C.c_int]),
    "az_device_count": (C.c_int, []),
    "az_env_step_batch": (C.c_int, [C.c_int, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "az_env_status_batch": (C.c_int, [C.c_int, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp]),
    "az_env_planes_batch": (C.c_int, [C.c_int, C.c_int, _i64, _vp, _vp]),
    "az_env_flip_batch": (C.c_int, [C.c_int, C.c_int, _i64, _vp, _vp, _vp]),
    "az_env_step_batch_dev": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "az_engine_create": (C.c_int, [_P(az_config), _P(_vp)]),
    "az_engine_destroy": (None, [_vp]),
    "az_engine_sync": (C.c_int, [_vp]),
    "az_get_config": (C.c_int, [_vp, _P(az_config)]),
    "az_get_stats": (C.c_int, [_vp, _P(az_stats)]),
    "az_reset_stats": (C.c_int, [_vp]),
    "az_set_profiling": (C.c_int, [_vp, C.c_int]),
    "az_get_profile": (C.c_int, [_vp, _vp, _vp]),
    "az_get_profile_busy": (C.c_int, [_vp, _vp]),
    "az_set_noise": (C.c_int, [_vp, _dbl]),
    "az_set_n_playout": (C.c_int, [_vp, _i32]),
    "az_reset_games": (C.c_int, [_vp]),
    "az_set_roots": (C.c_int, [_vp, _vp]),
    "az_get_roots": (C.c_int, [_vp, _vp, _vp, _vp]),
    "az_set_flip_stream": (C.c_int, [_vp, _vp, _i64]),
    "az_search": (C.c_int, [_vp, _i32]),
    "az_select": (C.c_int, [_vp, _vp, _vp]),
    "az_backup": (C.c_int, [_vp, _vp, _vp]),
    "az_get_root_children": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "az_choose_moves": (C.c_int, [_vp, _vp, _dbl, _vp, _vp]),
    "az_advance": (C.c_int, [_vp, _vp]),
    "az_set_game_target": (C.c_int, [_vp, _i64]),
    "az_set_active": (C.c_int, [_vp, _vp]),
    "az_selfplay": (_i64, [_vp, _i64, _dbl, _i32, _i64]),
    "az_selfplay_step": (C.c_int, [_vp, _dbl, _i32, _i32]),
    "az_selfplay_rounds": (C.c_int, [_vp, _i32, _dbl, _i32, _i32]),
    "az_selfplay_rolling": (_i64, [_vp, _i64, _dbl, _i32, _i64]),
    "az_traj_count": (C.c_int, [_vp, _P(_i64), _P(_i64)]),
    "az_traj_fetch": (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "az_traj_pack_dev": (_i64, [_vp, _i64, _vp, C.c_int]),
    "az_traj_record_bytes": (C.c_int, [C.c_int]),
    "az_traj_unpack_dev": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "az_net_load": (C.c_int, [_vp, _vp, C.c_size_t]),
    "az_net_forward": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "az_net_forward_static": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "az_net_set_kernel": (C.c_int, [_vp, C.c_int]),
    "az_net_get_kernel": (C.c_int, [_vp]),
    "az_net_forward_dev": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "az_set_root_children": (C.c_int, [_vp, _vp, _vp, _vp]),
    "az_puct_batch": (C.c_int, [C.c_int, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Loads libazamd.so; raises if it has not been built (python __graft_entry__.py / make -C alphazero_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AzError("libazamd.so is missing (%s): build it with `make -C alphazero_amd/csrc`; "
                      "there is no CPU fallback" % LIB_PATH)
    try:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; loading it FIRST makes libazamd bind
        # to that same runtime, so torch tensors' device pointers (collector.py) are valid inside our kernels.
        # (Loaded the other way round, torch's runtime finds no GPU.)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if isinstance(rc, int) and rc < 0:
        msg = lib().az_last_error()
        raise AzError("libazamd error %d: %s" % (rc, msg.decode() if msg else "?"))
    return rc


def ptr(a):
    """numpy array (or None, or a raw device address as int) -> void*"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().az_device_count()


def require_gpu():
    n = device_count()
    if n <= 0:
        raise AzError("no HIP device visible: alphazero_amd runs on MI355X only (no CPU path)")
    return n


def as_u64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if shape is not None:
        a = a.reshape(shape)
    return a
