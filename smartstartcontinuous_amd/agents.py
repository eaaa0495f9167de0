"""Agent-side drop-in surface: the ``RLAgent`` family of
smartstart/reinforcementLearningCore/agents_abstract_classes.py and GPU-backed counterparts of
``DDPG_Baselines_agent`` (action path) and ``NND_MB_agent`` (the SmartStart navigator).

Constructor keyword names follow the reference classes so existing call sites keep working; the
TensorFlow session argument ``sess`` is accepted and ignored.  Both learners run on the GPU: the DDPG step
(``ssc_ddpg_train``) and the dynamics-model training of the navigator
(``ssc_mlp_train_steps``, rank 3; ``NND_MB_agent.train_dynamics_model``).  Weights stay plain torch tensors
(TensorFlow layout) that a caller may also overwrite through ``set_weights``.
"""
from __future__ import annotations

import abc
import ctypes
import os

import numpy as np
import torch

from . import _ffi, navigator
from .numerical import (distances_left, elliptical_euclidean_distance_function_generator,
                        get_start_waypoints_final_states_steps, path_deltas_stds_and_means_per_dim,
                        path_shortcutter, radii_calc)
from .replay_buffer import ReplayBuffer
from .vec_env import ActorPolicy


# --------------------------------------------------------------------------------- ABCs --
class RLAgent(metaclass=abc.ABCMeta):
    """agents_abstract_classes.py:6-53"""

    @abc.abstractmethod
    def get_action(self, state):
        """state -> action appropriate for the environment"""

    @abc.abstractmethod
    def observe(self, state, action, reward, new_state, done):
        """called after every env.step"""

    @abc.abstractmethod
    def render(self, env, **kwargs):
        """render the environment the agent is in"""

    def start_new_episode(self, state):
        pass

    def end_episode(self):
        pass

    def get_param_dict(self):
        raise NotImplementedError("Agent hasn't overridden get_param_dict, if you don't wish to implement it "
                                  "just return None")


class NavigationRLAgent(RLAgent):
    """agents_abstract_classes.py:55-65"""

    def start_new_episode_plan(self, state, path_to_follow):
        raise NotImplementedError


class ValueFuncRLAgent(RLAgent):
    """agents_abstract_classes.py:68-80"""

    @abc.abstractmethod
    def get_state_value(self, state):
        """value of ``state`` (max over actions)"""


class ReplayBufferRLAgent(RLAgent):
    """agents_abstract_classes.py:82-91"""

    def __init__(self):
        self.replay_buffer = None

    def set_replay_buffer_main_agent(self, new_main_agent):
        self.replay_buffer.set_main_agent(new_main_agent)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------------- DDPG --
class DecayingOrnsteinUhlenbeckActionNoise:
    """DDPG_Baselines_agent.py:52-78 over baselines' OrnsteinUhlenbeckActionNoise [third-party]:
    x <- x + theta (mu - x) dt + sigma sqrt(dt) N(0,1); output x * max(epsilon, 0)."""

    def __init__(self, epsilon, min_epsilon, epsilon_decay_factor, mu, sigma, theta=.15, dt=1e-2, x0=None):
        self.mu, self.sigma, self.theta, self.dt, self.x0 = np.asarray(mu, np.float64), sigma, theta, dt, x0
        self.epsilon, self.min_epsilon, self.epsilon_decay_factor = epsilon, min_epsilon, epsilon_decay_factor
        self.reset()

    def __call__(self):
        x = self.x_prev + self.theta * (self.mu - self.x_prev) * self.dt + \
            self.sigma * np.sqrt(self.dt) * np.random.normal(size=self.mu.shape)
        self.x_prev = x
        return x * max(self.epsilon, 0)

    def reset(self):
        self.x_prev = self.x0 if self.x0 is not None else np.zeros_like(self.mu)

    def reduce_epsilon(self):
        self.epsilon = max(self.epsilon * self.epsilon_decay_factor, self.min_epsilon)


def init_critic_weights(obs_dim, h1, h2, nb_actions, generator=None):
    """Critic_Editted (models_editted.py:78-100): glorot-uniform kernels, zero biases, W2 is
    [h1 + nb_actions, h2] (the action joins after the first ReLU), last layer U(-3e-3, 3e-3)."""
    g = generator

    def glorot(i, o):
        lim = float(np.sqrt(6.0 / (i + o)))
        return (torch.rand((i, o), generator=g) * 2 - 1) * lim
    return dict(W1=glorot(obs_dim, h1), b1=torch.zeros(h1), W2=glorot(h1 + nb_actions, h2), b2=torch.zeros(h2),
                W3=(torch.rand((h2, 1), generator=g) * 2 - 1) * 3e-3, b3=torch.zeros(1))


def init_actor_weights(obs_dim, h1, h2, nb_actions, generator=None):
    """tf.layers.dense defaults in Actor_Editted (models_editted.py:44-59): glorot-uniform kernels,
    zero biases; last layer U(-3e-3, 3e-3)."""
    g = generator

    def glorot(i, o):
        lim = float(np.sqrt(6.0 / (i + o)))
        return (torch.rand((i, o), generator=g) * 2 - 1) * lim
    return dict(W1=glorot(obs_dim, h1), b1=torch.zeros(h1), W2=glorot(h1, h2), b2=torch.zeros(h2),
                W3=(torch.rand((h2, nb_actions), generator=g) * 2 - 1) * 3e-3, b3=torch.zeros(nb_actions))


PARAM_KEYS = ("W1", "b1", "W2", "b2", "W3", "b3")
# layer_norm=True (models_editted.py:45-46, 50-51, 85-86, 91-92): tc.layers.layer_norm creates beta, then gamma
LN_PARAM_KEYS = ("W1", "b1", "ln1_b", "ln1_g", "W2", "b2", "ln2_b", "ln2_g", "W3", "b3")


def param_keys(weights):
    return LN_PARAM_KEYS if "ln1_g" in weights else PARAM_KEYS


def with_layer_norm(weights):
    """The same network with tc.layers.layer_norm's initial parameters (gamma = 1, beta = 0) behind both hidden layers."""
    h1, h2 = weights["W1"].shape[1], weights["W2"].shape[1]
    return dict(weights, ln1_b=torch.zeros(h1), ln1_g=torch.ones(h1), ln2_b=torch.zeros(h2), ln2_g=torch.ones(h2))


def flatten_params(weights, device, extra=0):
    """One flat fp32 tensor in TensorFlow trainable_vars order [W1|b1|W2|b2|W3|b3] (with LayerNorm:
    [W1|b1|beta1|gamma1|W2|b2|beta2|gamma2|W3|b3]) plus a dict of VIEWS into it, so that kernels reading the dict see what
    the learner kernel wrote into the flat array.
    ``extra`` zero-initialised floats follow the parameters in the same allocation (flat[-extra:])."""
    keys = param_keys(weights)
    parts = [torch.as_tensor(weights[k], dtype=torch.float32).reshape(-1) for k in keys]
    if extra:
        parts.append(torch.zeros(int(extra), dtype=torch.float32))
    flat = torch.cat(parts).to(device).contiguous()
    views, o = {}, 0
    for k in keys:
        shape = tuple(torch.as_tensor(weights[k]).shape)
        n = int(np.prod(shape))
        views[k] = flat[o:o + n].view(shape)
        o += n
    return flat, views


class DDPG_Baselines_agent(ValueFuncRLAgent, ReplayBufferRLAgent):
    """Action path of smartstart/RLAgents/DDPG_Baselines_agent.py (:86-273): actor forward on the GPU
    (``ssc_actor_forward``), decaying OU noise, clip, double ``scale``; transitions go to the shared
    ReplayBuffer.  ``as_policy()`` hands the same actor to the fused rollout kernel."""

    def __init__(self, env, sess=None, replay_buffer=None, buffer_size=10000, batch_size=64, num_train_iterations=50,
                 num_steps_before_train=40, ou_epsilon=1.0, ou_min_epsilon=0.01, ou_epsilon_decay_factor=.99,
                 ou_mu=0.4, ou_sigma=0.2, ou_theta=.15, actor_lr=1e-4, actor_h1=64, actor_h2=64, critic_lr=1e-3,
                 critic_h1=64, critic_h2=64, gamma=0.99, tau=0.001, layer_norm=False, normalize_observations=False,
                 normalize_returns=False, critic_l2_reg=0, enable_popart=False, clip_norm=None, reward_scale=1.,
                 lastLayerTanh=False, finalizeGraph=True, device="cuda", precision="f32", seed=None, training=True):
        args = dict(locals())
        self.param_dict = {k: (v if isinstance(v, (int, float, bool, str, type(None))) else "Not serializable")
                           for k, v in args.items() if k not in ("self", "__class__")}   # :135-137
        if normalize_observations or normalize_returns or enable_popart:
            raise NotImplementedError("observation & return normalisation / popart are not on the "
                                      "accelerated path (every shipped run uses False)")
        self.layer_norm = bool(layer_norm)      # models_editted.py:45-46, 50-51, 85-86, 91-92 (fp32 kernels)
        if self.layer_norm:
            precision = "f32"
        self.env = env
        self.device = torch.device(device)
        self.lib = _ffi.lib()
        self.batch_size = batch_size
        self.num_train_iterations = num_train_iterations
        self.num_steps_before_train = num_steps_before_train
        self.remaining_steps_before_train = num_steps_before_train
        self.reward_scale = reward_scale
        self.gamma, self.tau, self.actor_lr, self.critic_lr = gamma, tau, actor_lr, critic_lr
        # ddpg_editted.py:183-191 / :175, 197 (unused by the shipped runs; the multi-workgroup learner carries both)
        if critic_l2_reg < 0 or (clip_norm is not None and not clip_norm > 0):
            raise ValueError("critic_l2_reg must be >= 0 and clip_norm None or > 0")
        self.critic_l2_reg, self.clip_norm = float(critic_l2_reg), clip_norm
        self.lastLayerTanh = bool(lastLayerTanh)
        # DDPG_editted clips what it feeds its networks to observation_range (ddpg_editted.py:66,106-109); the reference
        # agent never overrides the default (-5, 5), and it applies with normalize_observations=False too
        self.observation_range = (-5.0, 5.0)
        self.precision = precision
        self.replay_buffer = replay_buffer if replay_buffer is not None else ReplayBuffer(self, buffer_size)
        self.training_enabled = training
        nb_actions = env.action_space.shape[-1]
        obs_dim = env.observation_space.shape[-1]
        gen = torch.Generator().manual_seed(int(seed)) if seed is not None else None
        aw, cw = init_actor_weights(obs_dim, actor_h1, actor_h2, nb_actions, gen), init_critic_weights(obs_dim, critic_h1, critic_h2, nb_actions, gen)
        if self.layer_norm:
            aw, cw = with_layer_norm(aw), with_layer_norm(cw)
        self.set_weights(aw)
        self.set_critic_weights(cw)
        self.decaying_ou_action_noise = DecayingOrnsteinUhlenbeckActionNoise(
            ou_epsilon, ou_min_epsilon, ou_epsilon_decay_factor, mu=ou_mu * np.ones(nb_actions),
            sigma=float(ou_sigma) * np.ones(nb_actions), theta=ou_theta)   # :152-157
        self.ou = dict(mu=ou_mu, sigma=ou_sigma, theta=ou_theta)
        self.d_epsilon.fill_(float(max(ou_epsilon, 0.0)))

    # ---- weights ---------------------------------------------------------------------------
    def set_weights(self, weights):
        """weights: dict W1[obs,h1] b1 W2[h1,h2] b2 W3[h2,act] b3 (TensorFlow layout)."""
        # [parameters | epsilon]: what an actor needs from the learner is ONE contiguous array (one broadcast in the
        # sharded loop); the trailing float is the device copy of the OU epsilon (as_policy(device_epsilon=True))
        self.actor_sync, self.weights = flatten_params(weights, self.device, extra=1)
        self.actor_flat, self.d_epsilon = self.actor_sync[:-1], self.actor_sync[-1:]
        if hasattr(self, "decaying_ou_action_noise"):
            self.d_epsilon.fill_(float(max(self.decaying_ou_action_noise.epsilon, 0.0)))
        self.target_actor_flat = self.actor_flat.clone()        # target_init_updates (ddpg_editted.py:331-336)
        self._adam_actor = (torch.zeros_like(self.actor_flat), torch.zeros_like(self.actor_flat))
        if hasattr(self, "_adam_t"):
            self._adam_t[0] = 0                                  # fresh moments start their bias correction over
        w = self.weights
        self.obs_dim, self.h1 = w["W1"].shape
        self.h2, self.act_dim = w["W3"].shape
        d = _ffi.ActorDesc()
        d.obs_dim, d.h1, d.h2, d.act_dim = self.obs_dim, self.h1, self.h2, self.act_dim
        d.W1, d.b1, d.W2, d.b2, d.W3, d.b3 = (w[k].data_ptr() for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
        d.last_layer_tanh = int(self.lastLayerTanh)
        d.precision = _ffi.SSC_PREC_F32 if self.precision == "f32" else _ffi.SSC_PREC_BF16_MFMA
        d.obs_clip = float(self.observation_range[1])
        if "ln1_g" in w:
            d.ln1_g, d.ln1_b, d.ln2_g, d.ln2_b = (w[k].data_ptr() for k in ("ln1_g", "ln1_b", "ln2_g", "ln2_b"))
        self._desc = d

    def set_critic_weights(self, weights):
        """weights: dict W1[obs,h1] b1 W2[h1+act,h2] b2 W3[h2,1] b3 (Critic_Editted, models_editted.py:78-100)."""
        self.critic_flat, self.critic_weights = flatten_params(weights, self.device)
        self.target_critic_flat = self.critic_flat.clone()
        self._adam_critic = (torch.zeros_like(self.critic_flat), torch.zeros_like(self.critic_flat))
        self._adam_t = torch.zeros(2, dtype=torch.int32, device=self.device)
        w = self.critic_weights
        c = _ffi.CriticDesc()
        c.obs_dim, c.h1 = w["W1"].shape
        c.h2 = w["W2"].shape[1]
        c.act_dim = w["W2"].shape[0] - c.h1
        c.W1, c.b1, c.W2, c.b2, c.W3, c.b3 = (w[k].data_ptr() for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
        c.last_layer_tanh = int(self.lastLayerTanh)
        c.obs_clip = float(self.observation_range[1])
        if "ln1_g" in w:
            c.ln1_g, c.ln1_b, c.ln2_g, c.ln2_b = (w[k].data_ptr() for k in ("ln1_g", "ln1_b", "ln2_g", "ln2_b"))
        self._critic_desc = c

    def critic(self, obs, act):
        """Critic_Editted forward: Q(obs [m, obs_dim], act [m, act_dim]) -> [m]."""
        o = torch.as_tensor(obs, dtype=torch.float32, device=self.device).reshape(-1, self.obs_dim).contiguous()
        a = torch.as_tensor(act, dtype=torch.float32, device=self.device).reshape(o.shape[0], -1).contiguous()
        q = torch.empty(o.shape[0], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.ssc_critic_forward(ctypes.byref(self._critic_desc), o.shape[0], _ffi.ptr(o),
                                                   _ffi.ptr(a), _ffi.ptr(q), _stream()))
        return q

    def actor(self, obs):
        """Actor_Editted forward (models_editted.py:38-61) on a batch: obs [m, obs_dim] -> [m, act_dim]."""
        o = torch.as_tensor(obs, dtype=torch.float32, device=self.device).reshape(-1, self.obs_dim).contiguous()
        out = torch.empty((o.shape[0], self.act_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.ssc_actor_forward(ctypes.byref(self._desc), o.shape[0], _ffi.ptr(o), _ffi.ptr(out),
                                                  _stream()))
        return out

    # ---- RLAgent ---------------------------------------------------------------------------
    def scale(self, actions):
        """:236-240"""
        actions = np.clip(actions, -1, 1)
        low, high = self.env.action_space.low, self.env.action_space.high
        return (((actions + 1) / 2) * (high - low)) + low

    def get_action(self, state):
        """:206-234 -> DDPG_editted.pi (ddpg_editted.py:255-272)"""
        action = self.actor(np.asarray(state, np.float32)[None, :])[0].cpu().numpy()   # fp32, like sess.run
        noise = self.decaying_ou_action_noise()
        action = action + noise.astype(np.float32)             # in-place add keeps fp32 (:266-270)
        action = np.clip(action, -1.0, 1.0)                    # :271
        return self.scale(self.scale(action))

    def as_policy(self, precision=None, device_epsilon=False):
        """The same action path as a fused-rollout policy (current epsilon).  ``device_epsilon``: the kernel reads
        epsilon from ``self.d_epsilon`` (kept current by a :class:`rl_train.DecaySchedule`) instead of the host value."""
        n = self.decaying_ou_action_noise
        if "ln1_g" in self.weights:
            precision = "f32"                    # LayerNorm networks run on the fp32 kernels
        return ActorPolicy(self.weights, last_layer_tanh=self.lastLayerTanh, precision=precision or "bf16_mfma",
                           ou_mu=float(self.ou["mu"]), ou_sigma=float(self.ou["sigma"]), ou_theta=float(self.ou["theta"]),
                           ou_dt=n.dt, ou_epsilon=float(max(n.epsilon, 0)), obs_clip=float(self.observation_range[1]),
                           d_ou_epsilon=self.d_epsilon if device_epsilon else None)

    def get_state_value(self, state):
        """:197-204 -> DDPG_editted.get_q_value (ddpg_editted.py:274-279): Q(s, pi(s)) without noise.
        Returns [n, 1] for a batch, a length-1 array for a single state (like ``sess.run(...)[0]``)."""
        single = np.ndim(state) == 1 if not torch.is_tensor(state) else state.dim() == 1
        q = self.state_value_device(state).reshape(-1, 1).double().cpu().numpy()
        return q[0] if single else q

    def state_value_device(self, state):
        """Q(s, pi(s)) for a batch of states as a DEVICE tensor [m] (no host round trip: the SmartStart selection of the
        vectorised loop feeds candidate states gathered from the device replay ring)."""
        o = torch.as_tensor(state, dtype=torch.float32, device=self.device).reshape(-1, self.obs_dim)
        return self.critic(o, self.actor(o)).reshape(-1)

    def observe(self, state, action, reward, new_state, done):
        """:242-247 (store_transition, ddpg_editted.py:281-285)"""
        self.replay_buffer.add(self, state, action, reward * self.reward_scale, done, new_state)
        self.remaining_steps_before_train -= 1
        if self.training_enabled and self.remaining_steps_before_train <= 0:
            self.train()

    def start_new_episode(self, state):
        """:249-250 -- ``pass`` in the reference: a bare DDPG agent never marks episode starts in its buffer (only the
        SmartStart wrapper, the buffer's main agent then, does: smartexplorationcontinuous.py:369)."""

    def render(self, env, **kwargs):
        return env.render()

    def end_episode(self):
        """:255-258"""
        self.decaying_ou_action_noise.reset()
        self.decaying_ou_action_noise.reduce_epsilon()
        if self.training_enabled:
            self.train()

    def get_param_dict(self):
        return self.param_dict

    # ---- learner -------------------------------------------------------------
    def ddpg_desc(self):
        d = _ffi.DdpgDesc()
        d.obs_dim, d.act_dim = self.obs_dim, self.act_dim
        d.actor_h1, d.actor_h2 = self.h1, self.h2
        d.critic_h1, d.critic_h2 = self._critic_desc.h1, self._critic_desc.h2
        d.last_layer_tanh, d.batch_size = int(self.lastLayerTanh), self.batch_size
        d.actor, d.critic = self.actor_flat.data_ptr(), self.critic_flat.data_ptr()
        d.target_actor, d.target_critic = self.target_actor_flat.data_ptr(), self.target_critic_flat.data_ptr()
        d.adam_m_actor, d.adam_v_actor = (t.data_ptr() for t in self._adam_actor)
        d.adam_m_critic, d.adam_v_critic = (t.data_ptr() for t in self._adam_critic)
        d.adam_t = self._adam_t.data_ptr()
        d.gamma, d.tau, d.actor_lr, d.critic_lr = self.gamma, self.tau, self.actor_lr, self.critic_lr
        d.beta1, d.beta2, d.epsilon = 0.9, 0.999, 1e-8          # ddpg_editted.py:176,198
        d.obs_clip = float(self.observation_range[1])
        d.layer_norm = int("ln1_g" in self.weights)
        d.critic_l2_reg, d.clip_norm = self.critic_l2_reg, 0.0 if self.clip_norm is None else float(self.clip_norm)
        return d

    def train_on(self, s, a, r, t, s2, batch_idx, n_iters):
        """``n_iters`` x (DDPG_editted.train + update_target_net) on device replay arrays; ``batch_idx``
        int32 [n_iters, batch_size].  Returns the (critic_loss, actor_loss) tensor [n_iters, 2]."""
        rv = _ffi.ReplayView(s.data_ptr(), a.data_ptr(), r.data_ptr(), t.data_ptr(), s2.data_ptr(), s.shape[0])
        losses = torch.empty((n_iters, 2), dtype=torch.float32, device=self.device)
        d = self.ddpg_desc()
        if tuple(batch_idx.shape) != (n_iters, self.batch_size):
            raise ValueError(f"batch_idx must be [{n_iters}, {self.batch_size}], got {tuple(batch_idx.shape)}")
        with torch.cuda.device(self.device):
            # gradient partials of the multi-workgroup path (layers wider than 64 / batch != 64); kept with the agent
            need = self.lib.ssc_ddpg_train_workspace_bytes(ctypes.byref(d))
            ws = getattr(self, "_train_ws", None)
            if ws is None or ws.numel() < need:
                ws = self._train_ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            _ffi.check(self.lib.ssc_ddpg_train_ws(ctypes.byref(d), ctypes.byref(rv), _ffi.ptr(batch_idx), n_iters,
                                                  _ffi.ptr(losses), _ffi.ptr(ws), ws.numel(), _stream()))
        return losses

    def train_from(self, device_replay, n_iters=None):
        """``train()`` on a :class:`DeviceReplayBuffer`: batch indices drawn on the device, the records read
        where the rollout left them -- nothing crosses PCIe.  Returns the loss tensor [n_iters, 2] or None."""
        if len(device_replay) < self.batch_size:
            return None                                          # DDPG_Baselines_agent.py:265-266
        n = self.num_train_iterations if n_iters is None else int(n_iters)
        idx = device_replay.sample_indices(n, self.batch_size)
        r = device_replay
        return self.train_on(r.s, r.a, r.r, r.t, r.s2, idx, n)

    def train(self):
        """DDPG_Baselines_agent.train (:264-273): ``num_train_iterations`` x (train + update_target_net),
        each on a fresh uniform batch -- all iterations in one kernel launch."""
        if len(self.replay_buffer) < self.batch_size:
            return None
        self.remaining_steps_before_train = self.num_steps_before_train
        n = self.num_train_iterations
        B = self.batch_size
        cols = [[] for _ in range(5)]
        for _ in range(n):                                       # sample_batch per iteration (:287-289)
            for c, x in zip(cols, self.replay_buffer.sample_batch(B)):
                c.append(x)
        s, a, r, t, s2 = (np.concatenate(c) for c in cols)
        dev = self.device
        f = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x), dtype=dt).to(dev)
        idx = torch.arange(n * B, dtype=torch.int32, device=dev).view(n, B)
        losses = self.train_on(f(s, torch.float32), f(a, torch.float32), f(r, torch.float32), f(t, torch.uint8),
                               f(s2, torch.float32), idx, n)
        return losses


# ---------------------------------------------------------------------------- navigator --
def add_noise(data_inp, noiseToSignal, rng=None):
    """NN_Dynamics_Model/helper_funcs.py:10-17: per column, Gaussian noise of std ``mean(column) * noiseToSignal``
    -- applied only where that product is > 0, so columns with a negative or zero mean get none (the
    reference's behaviour, kept)."""
    rng = rng if rng is not None else np.random
    data = np.array(data_inp, dtype=np.float64, copy=True)
    if data.size == 0:
        return data
    std_of_noise = np.mean(data, axis=0) * noiseToSignal
    for j in range(std_of_noise.shape[0]):
        if std_of_noise[j] > 0:
            data[:, j] = data[:, j] + rng.normal(0, np.absolute(std_of_noise[j]), (data.shape[0],))
    return data


def init_dynamics_weights(in_dim, out_dim, num_fc_layers, depth_fc_layers, generator=None):
    """xavier-NORMAL weights and biases (feedforward_network.py:8, 14-23)."""
    dims = [in_dim] + [depth_fc_layers] * num_fc_layers + [out_dim]
    Ws, bs = [], []
    for i in range(len(dims) - 1):
        Ws.append(torch.randn((dims[i], dims[i + 1]), generator=generator) * float(np.sqrt(2.0 / (dims[i] + dims[i + 1]))))
        bs.append(torch.randn(dims[i + 1], generator=generator) * float(np.sqrt(2.0 / (1 + dims[i + 1]))))
    return Ws, bs


class NND_MB_agent(NavigationRLAgent):
    """The SmartStart navigator (smartstart/RLAgents/NND_MB_agent.py): MPC over a learned dynamics
    model -- sample ``num_control_samples`` action sequences of length ``horizon``, forward-simulate
    them (``ssc_dyn_forward_sim``), score waypoint progress (``ssc_mpc_score``), execute the first action
    of the best sequence plus small Gaussian noise (``ssc_mpc_select_action``)."""

    noiseToSignal = 0.01
    actions_ag = 'nc'

    def __init__(self, env, sess=None, replay_buffer=None, BUFFER_SIZE=10000,
                 final_steps=10, steps_per_waypoint=1, mean_per_stepsize=1, std_per_stepsize=1,
                 stepsizes_in_waypoint_radii=1,
                 gamma=.75, horizontal_penalty_factor=.5, horizon=20, num_control_samples=5000, path_shortcutting=True,
                 steps_before_giving_up_on_waypoint=5,
                 num_fc_layers=1, depth_fc_layers=500,
                 training_data=None, weights=None, biases=None, norm=None,
                 make_aggregated_dataset_noisy=True, nEpochs=30, fraction_use_new=0.9,
                 num_episodes_for_aggregation=3,
                 save_dir_name="save_untitled", load_dir_name="untitled_load", model_root=None,
                 save_training_data=False, load_existing_training_data=False,
                 save_resulting_dynamics_model=False, load_existing_dynamics_model=False,
                 make_training_dataset_noisy=True, num_rollouts_train=25, num_rollouts_val=20,
                 steps_per_rollout_train=333, steps_per_rollout_val=333,
                 device="cuda", precision="bf16_mfma", seed=1234, per_row_projection=False, noise_stream=0, **unused):
        self.env = env
        self.noise_stream = int(noise_stream)   # which pair of Philox streams add_noise draws the data set's noise from
        self.device = torch.device(device)
        self.replay_buffer = replay_buffer if replay_buffer is not None else ReplayBuffer(self, BUFFER_SIZE)
        self.final_steps, self.steps_per_waypoint = final_steps, steps_per_waypoint
        self.mean_per_stepsize, self.std_per_stepsize = mean_per_stepsize, std_per_stepsize
        self.stepsizes_in_waypoint_radii = stepsizes_in_waypoint_radii
        self.gamma, self.horizontal_penalty_factor = gamma, horizontal_penalty_factor
        self.horizon, self.N = horizon, num_control_samples
        self.path_shortcutting = path_shortcutting
        self.steps_before_giving_up_on_waypoint = steps_before_giving_up_on_waypoint
        self.make_aggregated_dataset_noisy = make_aggregated_dataset_noisy   # NND_MB_agent.py:66
        self.nEpochs, self.fraction_use_new = nEpochs, fraction_use_new      # :64, :68
        self.num_episodes_for_aggregation = num_episodes_for_aggregation     # :65
        self.num_episodes_finished = 0
        self.theta = 1                      # NND_MB_agent.py:143
        self.noise_amount = 0.005           # :188-191
        self.per_row_projection = per_row_projection
        self.seed, self._t = int(seed), 0
        state_dim = env.observation_space.shape[0]
        act_dim = env.action_space.shape[0]
        self._train_inputs = self._train_outputs = None
        self.states_val = self.controls_val = None
        # the reference keeps its data under <models>/NND_MB_agent/<dir_name>/training_data (:35-44, :181-185)
        root = model_root if model_root is not None else os.path.join(os.getcwd(), "models", "NND_MB_agent")
        self.load_dir, self.save_dir = os.path.join(root, load_dir_name), os.path.join(root, save_dir_name)
        if load_existing_training_data and training_data is None:
            # :203-213 -- e.g. the reference's own models/NND_MB_agent/default/training_data/*.npy
            d = os.path.join(self.load_dir, "training_data")
            training_data = {k: np.load(os.path.join(d, k + ".npy")) for k in ("dataX", "dataY", "dataZ")}
            self.states_val = np.load(os.path.join(d, "states_val.npy"), allow_pickle=True)
            self.controls_val = np.load(os.path.join(d, "controls_val.npy"), allow_pickle=True)
        if norm is None and training_data is None:
            norm = self._collect_training_data(num_rollouts_train, steps_per_rollout_train, num_rollouts_val,
                                               steps_per_rollout_val, make_training_dataset_noisy)
        elif norm is None:
            norm = self.normalisation_from_data(training_data["dataX"], training_data["dataY"], training_data["dataZ"])
        if training_data is not None:
            # z-scored (x, y) -> z training set (NND_MB_agent.py:302-319)
            with np.errstate(divide="ignore", invalid="ignore"):
                nz = lambda v, m, s: np.nan_to_num((np.asarray(v, np.float64) - m) / s)
                self._train_inputs = np.concatenate([nz(training_data["dataX"], norm["mean_x"], norm["std_x"]),
                                                     nz(training_data["dataY"], norm["mean_y"], norm["std_y"])], axis=1)
                self._train_outputs = nz(training_data["dataZ"], norm["mean_z"], norm["std_z"])
        if save_training_data:                                                  # :289-296
            d = os.path.join(self.save_dir, "training_data")
            os.makedirs(d, exist_ok=True)
            src = training_data if training_data is not None else \
                {k: getattr(self, k).double().cpu().numpy() for k in ("dataX", "dataY", "dataZ")}
            for k in ("dataX", "dataY", "dataZ"):
                np.save(os.path.join(d, k + ".npy"), np.asarray(src[k], np.float64))
            if self.states_val is not None:
                def stacked(rollouts):       # equal-length rollouts stack; ragged ones (a rollout hit a terminal state)
                    same = len({len(r) for r in rollouts}) <= 1          # become an object array, as numpy 1.15 made them
                    if same:
                        return np.asarray(rollouts)
                    out = np.empty(len(rollouts), dtype=object)
                    for i, r in enumerate(rollouts):
                        out[i] = np.asarray(r)
                    return out
                np.save(os.path.join(d, "states_val.npy"), stacked(self.states_val), allow_pickle=True)
                np.save(os.path.join(d, "controls_val.npy"), stacked(self.controls_val), allow_pickle=True)
        if weights is None:
            gen = torch.Generator().manual_seed(self.seed)
            weights, biases = init_dynamics_weights(state_dim + act_dim, state_dim, num_fc_layers, depth_fc_layers, gen)
        self.dyn_model = navigator.DynamicsModel(weights, biases, norm, state_dim, act_dim, device=device,
                                                 precision=precision)
        self.state_dim, self.act_dim = state_dim, act_dim
        self.save_resulting_dynamics_model = save_resulting_dynamics_model   # :179
        self.use_existing_dynamics_model = load_existing_dynamics_model      # :158
        self.desired_states = None
        self.radii = None                   # NND_MB_agent.py:168
        self.param_dict = None

    def _collect_training_data(self, num_rollouts_train, steps_per_rollout_train, num_rollouts_val,
                               steps_per_rollout_val, make_training_dataset_noisy):
        """The constructor's data-collection branch (NND_MB_agent.py:215-319) with everything resident in HBM:
        random-policy rollouts (one fused launch), (s, a, s' - s) formatting, optional ``add_noise`` on states and
        deltas (:263-266), column statistics and z-scoring.  Leaves ``dataX / dataY / dataZ`` (raw, device),
        the z-scored training matrices, ``states_val / controls_val`` (host lists like the reference keeps) and
        returns the ``norm`` dict."""
        from . import collect_samples as cs
        collector = cs.CollectSamples(self.env, cs.Policy_Random(self.env), seed=self.seed)
        train = collector.collect_dataset(num_rollouts_train, steps_per_rollout_train)
        if len(train) == 0:
            raise ValueError("the random rollouts produced no training rows")
        self.states_val, self.controls_val, _, _ = collector.collect_samples(num_rollouts_val, steps_per_rollout_val)
        if make_training_dataset_noisy:
            cs.add_noise_device(train.dataX, self.noiseToSignal, self.seed, stream_id=2 * self.noise_stream)
            cs.add_noise_device(train.dataZ, self.noiseToSignal, self.seed, stream_id=2 * self.noise_stream + 1)
        self.dataX, self.dataY, self.dataZ = train.dataX, train.dataY, train.dataZ
        (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (train.dataX, train.dataY, train.dataZ))
        host = lambda t: t.cpu().numpy()
        norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
        self._train_inputs, self._train_outputs = train.normalised(norm)  # :302-319, np.concatenate((dataX, dataY), axis=1)
        return norm

    @staticmethod
    def normalisation_from_data(dataX, dataY, dataZ):
        """NND_MB_agent.py:302-315 (mean / std per column)."""
        X, Y, Z = (np.asarray(v, np.float64) for v in (dataX, dataY, dataZ))
        return dict(mean_x=X.mean(0), std_x=(X - X.mean(0)).std(0), mean_y=Y.mean(0), std_y=(Y - Y.mean(0)).std(0),
                    mean_z=Z.mean(0), std_z=(Z - Z.mean(0)).std(0))

    # ---- planning (host, once per episode) -----------------------------------------------------
    def start_new_episode_plan(self, starting_state, path_to_follow):
        """NND_MB_agent.py:375-423, including the retraining of the dynamics model on the aggregated replay-buffer
        transitions every ``num_episodes_for_aggregation`` planned episodes (:420-423; the very first plan trains
        too).  An agent built from ``norm`` statistics alone has no initial data set and skips the training."""
        self.current_desired_state_index = 0
        self.actions_done_for_current_waypoint = 0
        stds, means = path_deltas_stds_and_means_per_dim(path_to_follow)
        self.stds = stds
        self.radii = radii_calc(means, stds, self.mean_per_stepsize, self.std_per_stepsize,
                                self.stepsizes_in_waypoint_radii)
        self.distance_function = elliptical_euclidean_distance_function_generator(self.radii)
        if self.path_shortcutting:
            self.path_to_follow = path_shortcutter(path_to_follow, self.distance_function, self.theta)
        else:
            self.path_to_follow = np.asarray(path_to_follow, np.float64)
        self.desired_states = np.asarray(get_start_waypoints_final_states_steps(self.path_to_follow,
                                                                                self.steps_per_waypoint))
        self.distances_left = distances_left(self.desired_states, self.distance_function)
        self._problems = None
        can_train = self._train_inputs is not None or self.use_existing_dynamics_model
        if can_train and self.num_episodes_finished % self.num_episodes_for_aggregation == 0:
            self.train_dynamics_model()                                       # :421-422
        self.num_episodes_finished += 1

    @property
    def current_desired_state(self):
        return self.desired_states[self.current_desired_state_index]

    @property
    def next_desired_state(self):
        return self.desired_states[min(self.current_desired_state_index + 1, len(self.desired_states) - 1)]

    def move_to_next(self, pt, desired_state_index, distance_to_curr, distance_to_next):
        """:491-496"""
        return np.logical_and(np.logical_or(distance_to_curr <= self.theta, distance_to_next <= distance_to_curr),
                              desired_state_index != len(self.desired_states) - 1)

    def close_enough_to_goal(self, current_state):
        """:425-432"""
        if self.distance_function(current_state, self.desired_states[-1]) <= self.theta:
            return True
        return self.current_desired_state_index == len(self.desired_states) - 1 and \
            self.final_steps <= self.actions_done_for_current_waypoint

    # ---- acting (GPU, every step) --------------------------------------------------------------
    def get_action(self, state):
        return self.get_action_with_predicted_states(state)[0]

    def get_best_sim_actions(self, curr_nn_state):
        """:498-520 -> (best_action, best_sim_number, best_sequence, best_path) and the device tensors."""
        low, high = self.env.action_space.low, self.env.action_space.high
        A = navigator.mpc_sample_actions(1, self.N, self.horizon, low, high, self.seed, 0, self._t, self.device)
        S = self.dyn_model.do_forward_sim(np.asarray(curr_nn_state, np.float32), A)
        wp = self.desired_states
        if len(wp) < 2:   # the reference would raise IndexError at desired_states[b + 1]
            wp = np.concatenate([wp, wp], axis=0)
            left = np.asarray([0.0, 0.0])
        else:
            left = self.distances_left
        ps = navigator.MpcProblemSet([wp], [left], [self.radii], [self.current_desired_state_index],
                                     device=self.device, theta=self.theta, gamma=self.gamma,
                                     horizontal_penalty_factor=self.horizontal_penalty_factor,
                                     per_row_projection=self.per_row_projection)
        scores, best, _ = navigator.mpc_score(ps, S)
        return A, S, scores, best

    def get_action_with_predicted_states(self, state):
        """:339-358"""
        self.actions_done_for_current_waypoint += 1
        A, S, scores, best = self.get_best_sim_actions(state)
        noise = self.noise_amount if self.actions_ag in ('nn', 'nc') else 0.0
        action, path = navigator.mpc_select_action(A, S, best, 1, noise, self.seed, 0, self._t)
        self._t += 1
        return action[0].double().cpu().numpy(), path[0].double().cpu().numpy()

    def observe(self, state, action, reward, new_state, done):
        """:360-373"""
        self.replay_buffer.add(self, state, action, reward, done, new_state)
        distance_to_current = self.distance_function(new_state, self.current_desired_state)
        distance_to_next = self.distance_function(new_state, self.next_desired_state)
        if self.move_to_next(new_state, self.current_desired_state_index, distance_to_current, distance_to_next) or \
                (self.actions_done_for_current_waypoint > self.steps_before_giving_up_on_waypoint and
                 self.current_desired_state_index != len(self.desired_states) - 1):
            self.current_desired_state_index += 1
            self.actions_done_for_current_waypoint = 0

    def render(self, env, **kwargs):
        env.render()

    def get_param_dict(self):
        return self.param_dict

    def aggregated_dataset(self, rng=None):
        """NND_MB_agent.train_dynamics_model :442-460: the replay buffer's (s, a, s2 - s) rows, optionally with
        ``add_noise`` on states and deltas (:451-453), z-scored with the statistics of the initial data set."""
        if len(self.replay_buffer) == 0:
            s = np.zeros((0, self.state_dim)); a = np.zeros((0, self.act_dim)); s2 = np.zeros((0, self.state_dim))
        else:
            s, a, _, _, s2 = self.replay_buffer.all_batch()
            s, a, s2 = (np.asarray(v, np.float64).reshape(len(s), -1) for v in (s, a, s2))
        new_x, new_z = s, s2 - s
        if self.make_aggregated_dataset_noisy:
            new_x = add_noise(new_x, self.noiseToSignal, rng)
            new_z = add_noise(new_z, self.noiseToSignal, rng)
        nm = self.dyn_model.norm
        col = lambda name, n: np.asarray([getattr(nm, name)[i] for i in range(n)], np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            nz = lambda v, m, sd: np.nan_to_num((v - m) / sd)
            x = nz(new_x, col("mean_x", self.state_dim), col("std_x", self.state_dim))
            y = nz(a, col("mean_y", self.act_dim), col("std_y", self.act_dim))
            z = nz(new_z, col("mean_z", self.state_dim), col("std_z", self.state_dim))
        return np.concatenate([x, y], axis=1), z

    def train_dynamics_model(self, dataX_new=None, dataZ_new=None, nEpoch=None, fraction_use_new=None, batchsize=512,
                             lr=0.001, rng=None):
        """NND_MB_agent.train_dynamics_model (:437-480; no TF saver): trains the model on the stored
        (normalised) initial data set mixed with the aggregated rows through ``Dyn_Model.train``'s batching
        (DynamicsModel.train -> ssc_mlp_train_step).  Without arguments the aggregated rows come from the
        replay buffer exactly as in the reference (``aggregated_dataset``); ``dataX_new`` / ``dataZ_new`` pass
        already normalised (x||y, z) rows instead."""
        if self.use_existing_dynamics_model:                     # :463-468: restore instead of training
            self.dyn_model.load(os.path.join(self.load_dir, "models", "finalModel.npz"))
            return 0
        if self._train_inputs is None:
            raise RuntimeError("NND_MB_agent was built without training_data")
        if dataX_new is None and dataZ_new is None:
            xn, zn = self.aggregated_dataset(rng)
        else:
            in_dim, out_dim = self.dyn_model.in_dim, self.dyn_model.out_dim
            xn = np.zeros((0, in_dim)) if dataX_new is None else np.asarray(dataX_new)
            zn = np.zeros((0, out_dim)) if dataZ_new is None else np.asarray(dataZ_new)
        nEpoch = self.nEpochs if nEpoch is None else nEpoch
        fraction_use_new = self.fraction_use_new if fraction_use_new is None else fraction_use_new
        loss = self.dyn_model.train(self._train_inputs, self._train_outputs, xn, zn, nEpoch, fraction_use_new,
                                    batchsize=batchsize, lr=lr, rng=rng)
        if self.save_resulting_dynamics_model:                   # :475-480 (an .npz instead of a TF checkpoint)
            d = os.path.join(self.save_dir, "models")
            os.makedirs(d, exist_ok=True)
            n_train = 1 + self.num_episodes_finished // self.num_episodes_for_aggregation
            self.dyn_model.save(os.path.join(d, "model_numTrain%d.npz" % n_train))
            self.dyn_model.save(os.path.join(d, "finalModel.npz"))
        return loss
