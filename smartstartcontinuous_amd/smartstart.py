"""``SmartStartContinuous`` (smartstart/smartexploration/smartexplorationcontinuous.py): with
probability ``eta`` an episode starts by navigating (NND_MB MPC) back to a "smart start" state
chosen by UCB1 over the critic value and a kernel-density visitation count, then hands over to the
base agent.  Same constructor keywords and RLAgent methods as the reference; selection runs on the
GPU (``ssc_critic_forward`` / ``ssc_kde_evaluate`` / ``ssc_ucb_argmax``)."""
from __future__ import annotations

import ctypes
import time

import numpy as np
import torch

from . import _ffi
from .agents import NND_MB_agent, ReplayBufferRLAgent, RLAgent
from .numerical import volume_of_n_dimensional_hyperellipsoid
from .replay_buffer import DeviceReplayBuffer, ReplayBuffer


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def kde_scott_bandwidth(data):
    """``scipy.stats.gaussian_kde(data.T, bw_method='scott')`` bandwidth [third-party; :260]:
    covariance = cov(data, ddof=1) * n^(-2/(d+4)).  The O(|D|) moments are reduced on the device in
    fp64; the d x d algebra is host-side.  Returns (whitening [d,d] fp32, norm)."""
    x = data.double()
    n, d = x.shape
    # two-pass covariance with elementwise kernels and column sums (torch.cov goes through an f64 GEMM that takes 5.5 ms
    # for a [2, 100 000] matrix on this stack -- 85 % of a whole smart-start selection; this is ~0.1 ms)
    xc = x - x.mean(dim=0, keepdim=True)
    cov = ((xc.unsqueeze(2) * xc.unsqueeze(1)).sum(dim=0) / (n - 1)).cpu().numpy() * (n ** (-1.0 / (d + 4))) ** 2
    inv = np.linalg.inv(cov)
    wh = np.linalg.cholesky(inv).T
    norm = 1.0 / (n * np.sqrt(np.linalg.det(2 * np.pi * cov)))
    return np.ascontiguousarray(wh, np.float32), float(norm)


def kde_evaluate(data, points, whitening, norm):
    """pdf of ``points`` [m, d] under the Gaussian KDE of ``data`` [n, d] (device tensors, fp32)."""
    data = data.float().contiguous()
    points = points.float().contiguous()
    n, d = data.shape
    m = points.shape[0]
    pdf = torch.empty(m, dtype=torch.float32, device=data.device)
    wh = (ctypes.c_float * (d * d))(*np.asarray(whitening, np.float32).reshape(-1).tolist())
    with torch.cuda.device(data.device):
        _ffi.check(_ffi.lib().ssc_kde_evaluate(d, n, _ffi.ptr(data), m, _ffi.ptr(points), wh, float(norm),
                                               _ffi.ptr(pdf), _stream()))
    return pdf


def ucb_argmax(values, pdf, buffer_len, volume, exploitation_param, exploration_param):
    """:275-280 -> (ucb [m], best index tensor [1])"""
    m = values.numel()
    ucb = torch.empty(m, dtype=torch.float32, device=values.device)
    best = torch.empty(1, dtype=torch.int32, device=values.device)
    with torch.cuda.device(values.device):
        _ffi.check(_ffi.lib().ssc_ucb_argmax(m, _ffi.ptr(values.float().contiguous()), _ffi.ptr(pdf),
                                             float(exploitation_param), float(exploration_param), float(buffer_len),
                                             float(volume), _ffi.ptr(ucb), _ffi.ptr(best), _stream()))
    return ucb, best


def device_smart_start_path(replay, agent, radii, n_ss, exploitation_param=1., exploration_param=2.):
    """``get_smart_start_path`` (smartexplorationcontinuous.py:223-305) with everything up to the chosen path on the
    device: candidates from the device ring's episode index (``DeviceReplayBuffer.get_possible_smart_start_indices``),
    V = Q(s, pi(s)) (``agent.state_value_device``), the Gaussian KDE over every state in the ring, UCB1 argmax, and the
    episodic path of the winner gathered out of the ring -- the replay contents never visit the host (the reference
    re-reads its whole buffer per episode, :258-260).  -> (path [L + 1, obs_dim] device tensor, buffer index) or None."""
    if not isinstance(replay, DeviceReplayBuffer) or not replay.track_episodes:
        raise TypeError("device_smart_start_path needs a DeviceReplayBuffer(track_episodes=True)")
    if len(replay) == 0:
        return None
    idx = replay.get_possible_smart_start_indices(n_ss)                                       # :243-246
    if idx is None:
        return None
    all_states = replay.get_all_states()                                                      # :258
    wh, norm = kde_scott_bandwidth(all_states)                                                # :260
    volume = volume_of_n_dimensional_hyperellipsoid(radii) if radii is not None else 1         # :262-268
    cand = replay.s2[replay.physical(idx)]                                                     # :272-273
    values = agent.state_value_device(cand)                                                    # :274
    pdf = kde_evaluate(all_states, cand, wh, norm)                                             # :275
    _, best = ucb_argmax(values, pdf, len(replay), volume, exploitation_param, exploration_param)   # :276-280
    chosen = idx[best.long()]                                                                  # stays on the device
    return replay.get_episodic_path_to_buffer_index(chosen), chosen


class SmartStartContinuous(RLAgent):
    def __init__(self, agent, env, sess=None, buffer_size=500000, exploitation_param=1., exploration_param=2.,
                 eta=0.5, eta_decay_factor=1., n_ss=1000, print_ss_stuff=True, device="cuda", **nnd_mb_kwargs):
        """``nnd_mb_*`` keywords are forwarded to :class:`NND_MB_agent` with the prefix stripped
        (smartexplorationcontinuous.py:55-198); navigator-only extras (``nnd_mb_training_data``,
        ``nnd_mb_weights`` ...) are accepted the same way."""
        self.param_dict = dict(buffer_size=buffer_size, exploitation_param=exploitation_param,
                               exploration_param=exploration_param, eta=eta, eta_decay_factor=eta_decay_factor,
                               n_ss=n_ss, agent=agent.get_param_dict())
        self.exploitation_param, self.exploration_param = exploitation_param, exploration_param
        self.eta, self.eta_decay_factor = eta, eta_decay_factor
        self.agent, self.env = agent, env
        self.device = torch.device(device)
        if isinstance(agent, ReplayBufferRLAgent):              # :140-144
            self.replay_buffer = agent.replay_buffer
            agent.set_replay_buffer_main_agent(self)
        else:
            self.replay_buffer = ReplayBuffer(self, buffer_size)
        self.n_ss, self.print_ss_stuff = n_ss, print_ss_stuff
        self.smart_start_pathing = False
        self.smart_start_path = None
        nav_kwargs = {k[len("nnd_mb_"):]: v for k, v in nnd_mb_kwargs.items() if k.startswith("nnd_mb_")}
        unknown = [k for k in nnd_mb_kwargs if not k.startswith("nnd_mb_")]
        if unknown:
            raise TypeError("unexpected keyword arguments: %s" % unknown)
        self.nnd_mb_agent = NND_MB_agent(env, sess, replay_buffer=self.replay_buffer, device=device, **nav_kwargs)
        self.times_for_smart_start = []

    def get_param_dict(self):
        return self.param_dict

    def get_summary_name(self):
        base = self.agent.get_summary_name() if hasattr(self.agent, 'get_summary_name') else self.agent.__class__.__name__
        return "SmartStartC_" + base

    @property
    def normal_agent_pathing(self):
        return not self.smart_start_pathing

    def reduce_eta(self):
        self.eta = self.eta * self.eta_decay_factor

    # ------------------------------------------------------------------------- selection --
    def smart_start_scores(self, possible_start_indices):
        """The device part of get_smart_start_path (:256-280) -> (ucb tensor, best position)."""
        all_states = torch.as_tensor(self.replay_buffer.get_all_states(), dtype=torch.float32, device=self.device)
        wh, norm = kde_scott_bandwidth(all_states)                                            # :260
        radii = self.nnd_mb_agent.radii
        volume = volume_of_n_dimensional_hyperellipsoid(radii) if radii is not None else 1      # :262-268
        _, _, _, _, s2 = self.replay_buffer._gather(np.asarray(possible_start_indices))
        cand = torch.as_tensor(s2, dtype=torch.float32, device=self.device)                    # :272-273
        values = torch.as_tensor(self.agent.get_state_value(cand), device=self.device).reshape(-1)   # :274
        pdf = kde_evaluate(all_states, cand, wh, norm)                                         # :275
        ucb, best = ucb_argmax(values, pdf, len(self.replay_buffer), volume, self.exploitation_param,
                               self.exploration_param)                                         # :276-280
        return ucb, int(best.item())

    def get_smart_start_path(self):
        """:223-305"""
        if isinstance(self.replay_buffer, DeviceReplayBuffer):
            got = device_smart_start_path(self.replay_buffer, self.agent, self.nnd_mb_agent.radii, self.n_ss,
                                          self.exploitation_param, self.exploration_param)
            return None if got is None else [row for row in got[0].double().cpu().numpy()]
        if len(self.replay_buffer) == 0:
            return None
        possible_start_indices = self.replay_buffer.get_possible_smart_start_indices(self.n_ss)
        if possible_start_indices is None:
            return None
        _, best = self.smart_start_scores(possible_start_indices)
        return self.replay_buffer.get_episodic_path_to_buffer_index(int(possible_start_indices[best]))

    # --------------------------------------------------------------------------- RLAgent --
    def get_action(self, state):
        """:307-317"""
        if self.smart_start_pathing:
            return self.nnd_mb_agent.get_action(state)
        return self.agent.get_action(state)

    def observe(self, state, action, reward, new_state, done):
        """:319-339"""
        self.replay_buffer.add(self, state, action, reward, done, new_state)
        self.agent.observe(state, action, reward, new_state, done)
        if self.smart_start_pathing:
            self.nnd_mb_agent.observe(state, action, reward, new_state, done)
            if self.nnd_mb_agent.close_enough_to_goal(new_state):
                self.smart_start_pathing = False

    def start_new_episode(self, state):
        """:341-370"""
        self.smart_start_pathing = False
        self.smart_start_path = None
        if np.random.rand() <= self.eta:
            t0 = time.time()
            self.smart_start_path = self.get_smart_start_path()
            self.times_for_smart_start.append(time.time() - t0)
            if self.smart_start_path:
                self.nnd_mb_agent.start_new_episode_plan(state, self.smart_start_path)
                if not self.nnd_mb_agent.close_enough_to_goal(state):
                    self.smart_start_pathing = True
        self.agent.start_new_episode(state)
        self.replay_buffer.start_new_episode(self)

    def end_episode(self):
        """:372-376"""
        self.reduce_eta()
        self.agent.end_episode()
        self.smart_start_pathing = False
        self.smart_start_path = None

    def render(self, env, **kwargs):
        return env.render()
