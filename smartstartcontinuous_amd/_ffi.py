"""ctypes binding of libssc.so (include/ssc.h).

The HIP library is the product; there is NO CPU fallback.  Importing this module without a
built ``libssc.so`` raises immediately (build it with ``python -m smartstartcontinuous_amd.build``
or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSC_LIB_PATH: development override (A/B of variant builds under tools/, e.g. tools/gpu_variant_ab.sh) -- the
# package itself always ships and loads the in-tree libssc.so
LIB_PATH = os.environ.get("SSC_LIB_PATH") or os.path.join(_HERE, "libssc.so")

SSC_OK, SSC_EINVAL, SSC_EUNSUPPORTED, SSC_EHIP = 0, -1, -2, -3
SSC_ENV_MOUNTAINCAR, SSC_ENV_PENDULUM = 0, 1
SSC_POLICY_RANDOM, SSC_POLICY_ACTOR = 0, 1
SSC_PREC_F32, SSC_PREC_BF16_MFMA, SSC_PREC_BF16_MFMA_PREPARED = 0, 1, 2
SSC_MAX_OBS, SSC_MAX_LAYERS, SSC_MAX_STATE, SSC_MAX_ACT = 3, 4, 8, 4


class SscError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libssc error {code}: {msg}")
        self.code = code


class EnvParams(Structure):
    _fields_ = [("kind", c_int32), ("max_episode_steps", c_int32),
                ("min_action", c_float), ("max_action", c_float),
                ("min_position", c_float), ("max_position", c_float),
                ("max_speed", c_float), ("goal_position", c_float), ("power", c_float),
                ("reset_low", c_float), ("reset_high", c_float),
                ("max_torque", c_float), ("pend_max_speed", c_float), ("dt", c_float),
                ("g", c_float), ("m", c_float), ("l", c_float), ("pend_v1_order", c_int32)]


class ActorDesc(Structure):
    _fields_ = [("obs_dim", c_int32), ("h1", c_int32), ("h2", c_int32), ("act_dim", c_int32),
                ("W1", c_void_p), ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p),
                ("W3", c_void_p), ("b3", c_void_p),
                ("last_layer_tanh", c_int32), ("precision", c_int32), ("obs_clip", c_float),
                ("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p), ("ln2_b", c_void_p)]


class OuDesc(Structure):
    _fields_ = [("mu", c_float), ("sigma", c_float), ("theta", c_float), ("dt", c_float), ("epsilon", c_float),
                ("d_epsilon", c_void_p)]


class PolicyDesc(Structure):
    _fields_ = [("kind", c_int32), ("act_low", c_float), ("act_high", c_float),
                ("actor", ActorDesc), ("ou", OuDesc)]


class RolloutState(Structure):
    _fields_ = [("s0", c_void_p), ("s1", c_void_p), ("steps", c_void_p), ("ep_ret", c_void_p), ("ou_x", c_void_p)]


class TransitionLog(Structure):
    _fields_ = [("obs", c_void_p * SSC_MAX_OBS), ("act", c_void_p), ("rew", c_void_p), ("done", c_void_p),
                ("obs2", c_void_p * SSC_MAX_OBS), ("row_stride", c_int64), ("done_row_stride", c_int64)]


class EpisodeRing(Structure):
    _fields_ = [("env_id", c_void_p), ("length", c_void_p), ("ret", c_void_p), ("cursor", c_void_p),
                ("capacity", c_int32)]


class MlpDesc(Structure):
    _fields_ = [("n_layers", c_int32), ("dims", c_int32 * (SSC_MAX_LAYERS + 1)),
                ("W", c_void_p * SSC_MAX_LAYERS), ("b", c_void_p * SSC_MAX_LAYERS)]


class Norm(Structure):
    _fields_ = [("mean_x", c_float * SSC_MAX_STATE), ("std_x", c_float * SSC_MAX_STATE),
                ("mean_y", c_float * SSC_MAX_ACT), ("std_y", c_float * SSC_MAX_ACT),
                ("mean_z", c_float * SSC_MAX_STATE), ("std_z", c_float * SSC_MAX_STATE)]


class MpcNavState(Structure):
    _fields_ = [("cur_idx", c_void_p), ("start_idx", c_void_p), ("actions_done", c_void_p), ("at_goal", c_void_p),
                ("give_up_after", c_int32), ("final_steps", c_int32)]


class MpcProblems(Structure):
    _fields_ = [("n_problems", c_int32), ("n_samples", c_int32), ("horizon", c_int32), ("state_dim", c_int32),
                ("wp", c_void_p), ("left", c_void_p), ("wp_off", c_void_p), ("cur_idx", c_void_p),
                ("radii", c_void_p), ("theta", c_float), ("gamma", c_float),
                ("horizontal_penalty_factor", c_float), ("per_row_projection", c_int32),
                ("plan_of", c_void_p), ("wp_len", c_void_p), ("active", c_void_p),
                ("live_list", c_void_p), ("n_live", c_void_p)]


class SmartStartStep(Structure):
    _fields_ = [("mode", c_void_p), ("plan_of", c_void_p), ("d_actor_out", c_void_p), ("d_eta", c_void_p),
                ("d_ou_epsilon", c_void_p), ("d_pool", c_void_p), ("ou", OuDesc), ("act_low", c_float), ("act_high", c_float),
                ("d_mode_log", c_void_p), ("mode_log_stride", c_int64), ("d_n_live", c_void_p)]


class MpcSampling(Structure):
    _fields_ = [("n_samples", c_int32), ("low", c_float * SSC_MAX_ACT), ("high", c_float * SSC_MAX_ACT),
                ("seed", c_uint64), ("problem_id0", c_uint64), ("t", c_uint64), ("d_t_base", c_void_p),
                ("d_problem_active", c_void_p), ("d_live_list", c_void_p), ("d_n_live", c_void_p)]


class CriticDesc(Structure):
    _fields_ = [("obs_dim", c_int32), ("act_dim", c_int32), ("h1", c_int32), ("h2", c_int32),
                ("W1", c_void_p), ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p),
                ("W3", c_void_p), ("b3", c_void_p), ("last_layer_tanh", c_int32), ("obs_clip", c_float),
                ("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p), ("ln2_b", c_void_p)]


class MlpTrainDesc(Structure):
    _fields_ = [("n_layers", c_int32), ("dims", c_int32 * (SSC_MAX_LAYERS + 1)),
                ("W", c_void_p * SSC_MAX_LAYERS), ("b", c_void_p * SSC_MAX_LAYERS),
                ("mW", c_void_p * SSC_MAX_LAYERS), ("vW", c_void_p * SSC_MAX_LAYERS),
                ("mb", c_void_p * SSC_MAX_LAYERS), ("vb", c_void_p * SSC_MAX_LAYERS),
                ("adam_t", c_void_p), ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("epsilon", c_float)]


class DdpgDesc(Structure):
    _fields_ = [("obs_dim", c_int32), ("act_dim", c_int32), ("actor_h1", c_int32), ("actor_h2", c_int32),
                ("critic_h1", c_int32), ("critic_h2", c_int32), ("last_layer_tanh", c_int32), ("batch_size", c_int32),
                ("actor", c_void_p), ("critic", c_void_p), ("target_actor", c_void_p), ("target_critic", c_void_p),
                ("adam_m_actor", c_void_p), ("adam_v_actor", c_void_p), ("adam_m_critic", c_void_p),
                ("adam_v_critic", c_void_p), ("adam_t", c_void_p),
                ("gamma", c_float), ("tau", c_float), ("actor_lr", c_float), ("critic_lr", c_float),
                ("beta1", c_float), ("beta2", c_float), ("epsilon", c_float), ("obs_clip", c_float), ("layer_norm", c_int32),
                ("critic_l2_reg", c_float), ("clip_norm", c_float)]


class ReplayView(Structure):
    _fields_ = [("s", c_void_p), ("a", c_void_p), ("r", c_void_p), ("t", c_void_p), ("s2", c_void_p),
                ("capacity", c_int64)]


class ReplayRing(Structure):
    _fields_ = [("s", c_void_p), ("a", c_void_p), ("r", c_void_p), ("t", c_void_p), ("s2", c_void_p),
                ("capacity", c_int64), ("obs_dim", c_int32), ("act_dim", c_int32),
                ("ep_steps", c_void_p), ("ep_run", c_void_p)]


# symbol -> (restype, argtypes); every function include/ssc.h declares must be listed here
# (tests/test_abi.py cross-checks the header against this table and the built library).
_SIGNATURES = {
    "ssc_version": (c_int, []),
    "ssc_last_error": (c_char_p, []),
    "ssc_env_params_default": (c_int, [c_int, c_float, c_int32, POINTER(EnvParams)]),
    "ssc_mc_step": (c_int, [POINTER(EnvParams), c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "ssc_pend_step": (c_int, [POINTER(EnvParams), c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "ssc_env_reset": (c_int, [POINTER(EnvParams), c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_uint64, c_uint64, c_uint64, c_void_p]),
    "ssc_env_observe": (c_int, [POINTER(EnvParams), c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssc_rollout": (c_int, [POINTER(EnvParams), POINTER(PolicyDesc), c_int64, c_int32, POINTER(RolloutState),
                            POINTER(TransitionLog), POINTER(EpisodeRing), c_void_p, c_uint64, c_uint64,
                            c_uint64, c_void_p]),
    "ssc_actor_forward": (c_int, [POINTER(ActorDesc), c_int64, c_void_p, c_void_p, c_void_p]),
    "ssc_pack_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "ssc_pack_transitions": (c_int, [POINTER(TransitionLog), c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p,
                                     c_void_p]),
    "ssc_critic_forward": (c_int, [POINTER(CriticDesc), c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssc_kde_evaluate": (c_int, [c_int32, c_int64, c_void_p, c_int64, c_void_p, POINTER(c_float), c_double, c_void_p,
                                 c_void_p]),
    "ssc_ucb_argmax": (c_int, [c_int64, c_void_p, c_void_p, c_float, c_float, c_double, c_double, c_void_p, c_void_p,
                               c_void_p]),
    "ssc_replay_append": (c_int, [POINTER(ReplayRing), POINTER(TransitionLog), c_int32, c_int64, c_int64, c_float,
                                  c_void_p]),
    "ssc_replay_append_shard": (c_int, [POINTER(ReplayRing), POINTER(TransitionLog), c_int32, c_int64, c_int64, c_int64, c_int64,
                                        c_float, c_void_p]),
    "ssc_decay_schedule": (c_int, [c_void_p, c_double, c_double, c_int32, POINTER(c_double), POINTER(c_double), c_void_p, POINTER(c_void_p),
                                   c_void_p]),
    "ssc_replay_sample": (c_int, [c_uint64, c_uint64, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ssc_replay_smart_start_workspace_bytes": (c_size_t, [c_int32]),
    "ssc_replay_smart_start_indices": (c_int, [POINTER(ReplayRing), c_int64, c_int64, c_int32, c_uint64, c_uint64,
                                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ssc_replay_episode_path": (c_int, [POINTER(ReplayRing), c_int64, c_int64, c_void_p, c_int32, c_void_p, c_void_p,
                                        c_void_p]),
    "ssc_ddpg_train": (c_int, [POINTER(DdpgDesc), POINTER(ReplayView), c_void_p, c_int32, c_void_p, c_void_p]),
    "ssc_ddpg_train_workspace_bytes": (c_size_t, [POINTER(DdpgDesc)]),
    "ssc_ddpg_train_ws": (c_int, [POINTER(DdpgDesc), POINTER(ReplayView), c_void_p, c_int32, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "ssc_dataset_scan_workspace_bytes": (c_size_t, [c_int64]),
    "ssc_dataset_scan": (c_int, [POINTER(TransitionLog), c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "ssc_dataset_build": (c_int, [POINTER(TransitionLog), c_int32, c_int32, c_int64, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssc_column_stats_workspace_bytes": (c_size_t, [c_int32]),
    "ssc_column_stats": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ssc_zscore": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ssc_path_shortcut": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_double, c_void_p, c_void_p]),
    "ssc_zscore_concat": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p,
                                  c_void_p]),
    "ssc_add_noise": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_double, c_uint64, c_uint64, c_void_p]),
    "ssc_mlp_train_workspace_bytes": (c_size_t, [POINTER(MlpTrainDesc), c_int32]),
    "ssc_mlp_train_steps": (c_int, [POINTER(MlpTrainDesc), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "ssc_mlp_train_step": (c_int, [POINTER(MlpTrainDesc), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "ssc_mlp_workspace_bytes": (c_size_t, [POINTER(MlpDesc), c_int64, c_int]),
    "ssc_mlp_forward": (c_int, [POINTER(MlpDesc), c_int64, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "ssc_mse_batches": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "ssc_dyn_workspace_bytes": (c_size_t, [POINTER(MlpDesc), c_int64, c_int]),
    "ssc_dyn_prepare": (c_int, [POINTER(MlpDesc), POINTER(Norm), c_void_p, c_size_t, c_void_p]),
    "ssc_dyn_forward_sim": (c_int, [POINTER(MlpDesc), POINTER(Norm), c_int64, c_int32, c_int32, c_int32, c_void_p,
                                    c_int64, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "ssc_mpc_forward_sim": (c_int, [POINTER(MlpDesc), POINTER(Norm), POINTER(MpcSampling), c_int64, c_int32, c_int32, c_int32,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "ssc_mpc_score_select": (c_int, [POINTER(MpcProblems), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     POINTER(MpcSampling), c_int32, c_float, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "ssc_mpc_sample_actions": (c_int, [c_int32, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(c_float),
                                       c_uint64, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "ssc_mpc_rollout_step": (c_int, [POINTER(EnvParams), POINTER(MpcProblems), POINTER(MpcNavState), c_void_p, c_void_p,
                                     c_float, c_uint64, c_uint64, POINTER(RolloutState), POINTER(TransitionLog),
                                     POINTER(EpisodeRing), c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "ssc_smartstart_rollout_step": (c_int, [POINTER(EnvParams), POINTER(MpcProblems), POINTER(MpcNavState), POINTER(SmartStartStep),
                                            c_void_p, c_void_p, c_float, c_uint64, c_uint64, POINTER(RolloutState),
                                            POINTER(TransitionLog), POINTER(EpisodeRing), c_void_p, c_uint64, c_uint64, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssc_nav_compact": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssc_mpc_score_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "ssc_mpc_score": (c_int, [POINTER(MpcProblems), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                              c_void_p]),
    "ssc_mpc_observe": (c_int, [POINTER(MpcProblems), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "ssc_mpc_select_action": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                      c_float, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def lib():
    """Load libssc.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP library is the product and there is no CPU fallback. "
                "Build it with `python -m smartstartcontinuous_amd.build` (needs hipcc).")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != SSC_OK:
        raise SscError(rc, lib().ssc_last_error().decode("utf-8", "replace"))


def default_params(kind, power_scalar=1.0, max_episode_steps=0):
    p = EnvParams()
    check(lib().ssc_env_params_default(kind, power_scalar, max_episode_steps, ctypes.byref(p)))
    return p


def ptr(t):
    """data_ptr() of a tensor (or None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())
