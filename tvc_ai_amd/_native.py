"""ctypes loader for libtvc_hip.so (the C ABI declared in include/tvc_native.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if it cannot be
loaded, or no GPU is visible when a handle is created, the product path raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loaded first so that libamdhip64.so.7 resolves to the copy torch ships)

from . import build as _build

_lib = None


class TvcError(RuntimeError):
    pass


class EnvCfg(C.Structure):
    """Mirror of `struct tvc_env_cfg` (include/tvc_native.h)."""
    _fields_ = [
        ("mass", C.c_double), ("inertia_xx", C.c_double), ("inertia_zz", C.c_double),
        ("thrust", C.c_double), ("half_len", C.c_double), ("radius", C.c_double),
        ("lin_damp", C.c_double), ("ang_damp", C.c_double), ("gravity", C.c_double),
        ("dt_sub", C.c_double),
        ("n_sub", C.c_int32), ("max_episode_steps", C.c_int32), ("distinct_window", C.c_int32),
        ("contact", C.c_int32), ("auto_reset", C.c_int32),
        ("mu", C.c_double), ("erp", C.c_double), ("cop_s0", C.c_double),
        ("init_pos", C.c_double * 3), ("init_quat", C.c_double * 4),
        ("dr_enabled", C.c_int32), ("_pad0", C.c_int32),
        ("dr_mass_var", C.c_double), ("dr_thrust_std", C.c_double), ("dr_cg_max", C.c_double),
        ("dr_wind_std", C.c_double), ("dr_init_tilt_max", C.c_double), ("dr_obs_noise_std", C.c_double),
        ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
    ]


_VP = C.c_void_p

# name -> (restype, argtypes); the export list tests check against include/tvc_native.h
SIGNATURES = {
    "tvc_last_error": (C.c_char_p, []),
    "tvc_abi_version": (C.c_int, []),
    "tvc_env_default_cfg": (None, [C.POINTER(EnvCfg)]),
    "tvc_env_create": (C.c_int, [C.POINTER(EnvCfg), C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "tvc_env_destroy": (None, [_VP]),
    "tvc_env_num_envs": (C.c_int32, [_VP]),
    "tvc_env_set_dr": (C.c_int, [_VP, C.POINTER(EnvCfg)]),
    "tvc_env_set_dr_async": (C.c_int, [_VP, C.POINTER(EnvCfg), _VP]),
    "tvc_env_reset": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP]),
    "tvc_env_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "tvc_env_step_many": (C.c_int, [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "tvc_env_export_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "tvc_env_import_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "tvc_env_info": (C.c_int, [_VP, _VP, _VP]),
    "tvc_env_set_components_out": (C.c_int, [_VP, _VP]),
    "tvc_env_set_episode_stats": (C.c_int, [_VP, _VP, _VP]),
    "tvc_debug_hwid": (C.c_int, [_VP, C.c_int32, _VP]),
    "tvc_debug_philox": (C.c_int, [_VP, _VP, C.c_int32, _VP]),
    "tvc_env_fuel_thresholds": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}


class SacCfg(C.Structure):
    """Mirror of `struct tvc_sac_cfg` (include/tvc_native.h)."""
    _fields_ = [
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("family", C.c_int32),
        ("d_model", C.c_int32), ("n_layers", C.c_int32), ("ff_dim", C.c_int32), ("head1", C.c_int32), ("head2", C.c_int32),
        ("mlp1", C.c_int32), ("mlp2", C.c_int32), ("critic1", C.c_int32), ("critic2", C.c_int32),
        ("batch_size", C.c_int32), ("max_act_rows", C.c_int32), ("pe_rows", C.c_int32),
        ("gamma", C.c_float), ("alpha", C.c_float), ("tau", C.c_float), ("lr", C.c_float),
        ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float), ("use_se", C.c_int32),
        ("dropout_p", C.c_float), ("nhead", C.c_int32), ("dropout_seed", C.c_uint32),
    ]


SIGNATURES.update({
    "tvc_sac_default_cfg": (None, [C.POINTER(SacCfg), C.c_int32]),
    "tvc_sac_param_count": (C.c_int64, [C.POINTER(SacCfg)]),
    "tvc_sac_trainable_count": (C.c_int64, [C.POINTER(SacCfg)]),
    "tvc_sac_num_tensors": (C.c_int32, [C.POINTER(SacCfg)]),
    "tvc_sac_tensor_info": (C.c_int, [C.POINTER(SacCfg), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tvc_sac_create": (C.c_int, [C.POINTER(SacCfg), C.c_int32, _VP, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "tvc_sac_destroy": (None, [_VP]),
    "tvc_sac_get_adam_steps": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "tvc_sac_set_adam_steps": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "tvc_sac_get_act_counter": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "tvc_sac_set_act_counter": (C.c_int, [_VP, C.c_int32]),
    "tvc_sac_sync_derived": (C.c_int, [_VP, _VP]),
    "tvc_sac_snapshot_policy": (C.c_int, [_VP, _VP]),
    "tvc_sac_enable_x3": (C.c_int, [_VP, _VP]),
    "tvc_sac_act": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, C.c_int32, _VP]),
    "tvc_debug_rows_clock": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.POINTER(C.c_double), _VP]),
    "tvc_debug_rows_stamps": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), _VP]),
    "tvc_mlp_param_count": (C.c_int64, [C.POINTER(C.c_int32), C.c_int32]),
    "tvc_mlp_tensor_offset": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tvc_mlp_layout": (C.c_int64, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tvc_goal_sample": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP]),
    "tvc_mlp_create": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, C.POINTER(_VP)]),
    "tvc_mlp_destroy": (None, [_VP]),
    "tvc_mlp_forward": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, _VP, C.c_int32, C.c_int32, _VP, _VP]),
    "tvc_curiosity_add": (C.c_int, [_VP, _VP, C.c_int32, _VP, C.c_int32, _VP, _VP, _VP, C.c_int32, _VP]),
    "tvc_safety_apply": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, C.c_int32, C.c_float, C.c_float, C.c_float, _VP]),
    "tvc_sac_critic_grads": (C.c_int, [_VP] + [_VP] * 8),
    "tvc_sac_critic_apply": (C.c_int, [_VP, C.c_float, _VP]),
    "tvc_sac_actor_grads": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "tvc_sac_actor_apply": (C.c_int, [_VP, C.c_float, _VP]),
    "tvc_sac_update": (C.c_int, [_VP] + [_VP] * 9),
    "tvc_sac_q_values": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int32, _VP, _VP]),
    "tvc_nn_linear_ln_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "tvc_nn_linear_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "tvc_replay_create": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "tvc_replay_destroy": (None, [_VP]),
    "tvc_replay_export": (C.c_int, [_VP, _VP, C.POINTER(C.c_int64)]),
    "tvc_replay_import": (C.c_int, [_VP, _VP, C.POINTER(C.c_int64)]),
    "tvc_replay_size": (C.c_int64, [_VP]),
    "tvc_replay_insert": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32, _VP]),
    "tvc_replay_sample": (C.c_int, [_VP, C.c_int32, C.c_uint64, C.c_uint64, _VP, _VP, _VP, _VP, _VP, _VP]),
})


def lib_path():
    return _build.LIB


def load():
    """Load (building first if needed) libtvc_hip.so and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build()
    try:
        L = C.CDLL(path)
    except OSError as e:  # fail loudly: the HIP library IS the product
        raise TvcError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise TvcError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().tvc_last_error()
        raise TvcError(f"libtvc_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream
