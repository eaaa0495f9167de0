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


class VecSmartStart:
    """``SmartStartContinuous`` for the N envs of a :class:`VecEnv` at once (smartexplorationcontinuous.py:307-376,
    vectorised): every env is either navigating to a smart-start state (per-env mode 1: the NND_MB MPC picks its
    action) or handed over to the base DDPG agent (mode 0: actor + OU noise); an env that finishes an episode starts
    the next one -- inside the step kernel, without the host -- with a smart start with probability ``eta``.

    What one env does is what the scalar agent does: ``get_action`` by mode (:307-317), waypoint bookkeeping and the
    hand-over test after every navigated step (:319-339), ``start_new_episode`` on the reset state (:341-370).  What is
    batched is the smart-start SELECTION (:223-305): instead of one selection per finished episode, ``refresh_plans``
    runs it once per rollout chunk on the device replay ring (candidates, Q(s, pi(s)), Gaussian KDE, UCB1), turns the
    ``n_plans`` best candidates' episodic paths into plans (radii, optional shortcutting, waypoints, distances_left --
    NND_MB_agent.start_new_episode_plan's host geometry) and publishes them in a :class:`navigator.PlanPool`; the envs
    finishing during the chunk draw from those.  ``n_plans = 1`` is the reference's argmax for every episode that starts
    before the next refresh.  One step = five launches -- compaction of the navigating envs (``ssc_nav_compact``), actor
    forward, forward simulation drawing its own candidates, scoring (one launch for N <= 64 candidates, two above), and
    ``ssc_smartstart_rollout_step`` -- captured once as a HIP graph and replayed K times per chunk.  (Evaluating the actor
    inside the step kernel was built and measured in round 4: 42.5 us against 21.3 + 10.8 us for the two launches -- the
    step kernel is a one-wave-per-SIMD latency chain and the actor's weight fetch lengthens it; NOTEBOOK section 10.)
    """

    def __init__(self, env, agent, dyn_model, eta=0.5, eta_decay_factor=1.0, n_ss=1000, exploitation_param=1.,
                 exploration_param=2., n_plans=1, n_slots=None, w_max=None, num_control_samples=64, horizon=4,
                 noise_amount=0.005, steps_before_giving_up_on_waypoint=5, final_steps=10, theta=1.0, gamma=0.75,
                 horizontal_penalty_factor=0.5, path_shortcutting=True, mean_per_stepsize=1, std_per_stepsize=1,
                 stepsizes_in_waypoint_radii=1, steps_per_waypoint=1, chunk_steps=64, seed=1234, log_modes=False,
                 kde_max_states=500000):
        from . import navigator as nav
        self.env, self.agent, self.model = env, agent, dyn_model
        self.eta, self.eta_decay_factor = float(eta), float(eta_decay_factor)
        self.n_ss, self.n_plans = int(n_ss), int(n_plans)
        self.exploitation_param, self.exploration_param = exploitation_param, exploration_param
        self.path_shortcutting, self.steps_per_waypoint = path_shortcutting, steps_per_waypoint
        self.mean_per_stepsize, self.std_per_stepsize = mean_per_stepsize, std_per_stepsize
        self.stepsizes_in_waypoint_radii = stepsizes_in_waypoint_radii
        dev = env.device
        max_steps = env.spec.max_episode_steps or 1000
        w_max = int(w_max) if w_max is not None else max_steps + 1
        if n_slots is None:        # a published plan must outlive every episode that started on it
            n_slots = self.n_plans * (-(-max_steps // max(int(chunk_steps), 1)) + 2)
        self.pool = nav.PlanPool(env.n, n_slots, w_max, env.obs_dim, dev, theta=theta, gamma=gamma,
                                 horizontal_penalty_factor=horizontal_penalty_factor)
        self.nav = nav.NavigatorBatch(dyn_model, self.pool, num_control_samples=num_control_samples, horizon=horizon,
                                      action_low=env.action_space.low, action_high=env.action_space.high,
                                      noise_amount=noise_amount,
                                      steps_before_giving_up_on_waypoint=steps_before_giving_up_on_waypoint,
                                      final_steps=final_steps, seed=seed, problem_id0=env.env_id0)
        self.nav.start_idx.zero_()
        self.mode = torch.zeros(env.n, dtype=torch.uint8, device=dev)
        self.pool.active = self.mode     # the MPC launches skip the envs that are not navigating (where the kernels can)
        # ... and spend their threads on the navigating envs only: every step starts by compacting ``mode`` into a list
        # (``ssc_nav_compact``); simulation (fused small-network kernel) and scoring (one-launch scorer) then cover
        # list[0 .. count) with the launch geometry of all envs (HIP-graph safe: blocks past the count exit at once).
        # The step kernel -- the last launch of a step -- leaves the counter at zero for the next step.
        self.live_list = torch.zeros(env.n, dtype=torch.int32, device=dev)
        self.n_live = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pool.live_list, self.pool.n_live = self.live_list, self.n_live
        self.actor_out = torch.zeros((env.n, 1), dtype=torch.float32, device=dev)
        self.d_eta = torch.tensor([self.eta], dtype=torch.float32, device=dev)
        self.d_eps = torch.tensor([0.0], dtype=torch.float32, device=dev)
        self.log_modes = bool(log_modes)
        # The reference estimates the visitation density from EVERY state in its buffer (<= 100 000 there, :258-260); a
        # device ring sized for 65 536 envs holds tens of millions, and the n_ss x |D| kernel then is most of a selection.
        # kde_max_states bounds |D| by an evenly strided subsample of the ring (a density estimate does not depend on the
        # sample count).  Default 500 000 = the reference's own replay capacity (SmartStartContinuous(buffer_size=500000),
        # :55), i.e. the most states its KDE can ever see; None = every state in the ring.
        self.kde_max_states = kde_max_states
        self.mode_log = None
        self.last_radii = None
        self._graphs = {}
        self.selections = 0

    # ------------------------------------------------------------------------- selection --
    def plan_from_path(self, path):
        """NND_MB_agent.start_new_episode_plan's geometry (NND_MB_agent.py:385-418) -> (waypoints, distances_left, radii)."""
        from .numerical import (distances_left, elliptical_euclidean_distance_function_generator,
                                get_start_waypoints_final_states_steps, path_deltas_stds_and_means_per_dim,
                                path_shortcutter, radii_calc)
        path = np.asarray(path, np.float64)
        stds, means = path_deltas_stds_and_means_per_dim(path)
        radii = radii_calc(means, stds, self.mean_per_stepsize, self.std_per_stepsize, self.stepsizes_in_waypoint_radii)
        dist = elliptical_euclidean_distance_function_generator(radii)
        if self.path_shortcutting:
            path = path_shortcutter(path, dist, self.pool.theta)
        wp = np.asarray(get_start_waypoints_final_states_steps(path, self.steps_per_waypoint))
        return wp, distances_left(wp, dist), radii

    def select_plans(self, replay):
        """One smart-start selection on the device ring (get_smart_start_path, :223-305) -> up to ``n_plans`` plans
        (waypoints, distances_left, radii) as host arrays, or None when the ring holds no complete episode start yet.
        The device part runs on the CURRENT stream (rl_train_vec_smartstart(overlap_selection=True) makes that a side
        stream) and the host reads its results; nothing here writes what a rollout in flight reads."""
        if len(replay) == 0:
            return None
        idx = replay.get_possible_smart_start_indices(self.n_ss)                                  # :243-246
        if idx is None:
            return None
        all_states = replay.get_all_states()                                                      # :258
        if self.kde_max_states is not None and all_states.shape[0] > self.kde_max_states:
            stride = -(-all_states.shape[0] // int(self.kde_max_states))
            all_states = all_states[::stride].contiguous()
        wh, norm = kde_scott_bandwidth(all_states)                                                # :260
        volume = volume_of_n_dimensional_hyperellipsoid(self.last_radii) if self.last_radii is not None else 1   # :262-268
        cand = replay.s2[replay.physical(idx)]                                                    # :272-273
        values = self.agent.state_value_device(cand)                                              # :274
        pdf = kde_evaluate(all_states, cand, wh, norm)                                            # :275
        ucb, best = ucb_argmax(values, pdf, len(replay), volume, self.exploitation_param, self.exploration_param)
        if self.n_plans == 1:
            chosen = idx[best.long()]
        else:
            chosen = idx[torch.topk(ucb, min(self.n_plans, ucb.numel())).indices]
        self.last_chosen = chosen
        plans = []
        for c in chosen.reshape(-1):
            path = replay.get_episodic_path_to_buffer_index(c.reshape(1))
            if path is None or path.shape[0] < 2:
                continue
            plans.append(self.plan_from_path(path.double().cpu().numpy()))
        return plans

    def publish_plans(self, plans):
        """Put ``plans`` on offer (writes the pool on the current stream)."""
        if plans:
            self.pool.publish(plans, now=self.env.t, min_age=(self.env.spec.max_episode_steps or 1000))
            self.last_radii = plans[0][2]
            self.selections += 1

    def refresh_plans(self, replay):
        """``select_plans`` + ``publish_plans``.  Returns the chosen buffer indices (device tensor) or None when the ring
        holds no complete episode start yet."""
        self.last_chosen = None
        plans = self.select_plans(replay)
        if plans is None:
            return None
        self.publish_plans(plans)
        return self.last_chosen

    # ------------------------------------------------------------------------------ steps --
    def _step_struct(self, chunk_k):
        e, a = self.env, self.agent
        ss = _ffi.SmartStartStep()
        ss.mode, ss.plan_of = self.mode.data_ptr(), self.pool.plan_of.data_ptr()
        ss.d_actor_out, ss.d_eta, ss.d_ou_epsilon = self.actor_out.data_ptr(), self.d_eta.data_ptr(), self.d_eps.data_ptr()
        ss.d_pool = self.pool.pool.data_ptr()
        n = a.decaying_ou_action_noise
        ss.ou.mu, ss.ou.sigma, ss.ou.theta, ss.ou.dt = float(a.ou["mu"]), float(a.ou["sigma"]), float(a.ou["theta"]), float(n.dt)
        ss.act_low, ss.act_high = float(e.action_space.low[0]), float(e.action_space.high[0])
        ss.d_n_live = self.n_live.data_ptr()
        if self.log_modes:
            if self.mode_log is None or self.mode_log.shape != (chunk_k, e.n):
                self.mode_log = torch.zeros((chunk_k, e.n), dtype=torch.uint8, device=e.device)
            ss.d_mode_log, ss.mode_log_stride = self.mode_log.data_ptr(), e.n
        return ss

    def fused_step(self, chunk, ring):
        """Enqueue ONE step for every env (five launches); step index and log row are device counters."""
        from . import navigator as nav
        env, b, lib = self.env, self.nav, _ffi.lib()
        fb = b._fused_buffers(env.device)
        with torch.cuda.device(env.device):
            _ffi.check(lib.ssc_nav_compact(env.n, _ffi.ptr(self.mode), _ffi.ptr(self.live_list), _ffi.ptr(self.n_live), _stream()))
            _ffi.check(lib.ssc_actor_forward(ctypes.byref(self.agent._desc), env.n, _ffi.ptr(fb["plan"]),
                                             _ffi.ptr(self.actor_out), _stream()))
        sp = nav.mpc_sampling(b.N, b.low, b.high, b.seed, b.problem_id0, 0, t_base=fb["t"], active=self.mode,
                              live_list=self.live_list, n_live=self.n_live)
        S = self.model.do_forward_sim_sampled(fb["plan"], sp, b.P * b.N, b.H, out=b._S, A_out=fb["A"])
        st = self.pool.as_struct(b.N, b.H)
        navs = _ffi.MpcNavState(self.pool.cur_idx.data_ptr(), b.start_idx.data_ptr(), b.actions_done.data_ptr(),
                                b.at_goal.data_ptr(), b.give_up, b.final_steps)
        rs = _ffi.RolloutState(env.s0.data_ptr(), env.s1.data_ptr(), env.steps.data_ptr(), env.ep_ret.data_ptr(),
                               env.ou_x.data_ptr())
        log_s = chunk.as_struct() if chunk is not None else None
        ring_s = ring.as_struct() if ring is not None else None
        ss = self._step_struct(chunk.K if chunk is not None else 1)
        with torch.cuda.device(env.device):
            _ffi.check(lib.ssc_mpc_score(ctypes.byref(st), _ffi.ptr(S), _ffi.ptr(fb["scores"]), _ffi.ptr(fb["best"]),
                                         _ffi.ptr(fb["best_score"]), _ffi.ptr(fb["ws"]), fb["ws"].numel(), _stream()))
            _ffi.check(lib.ssc_smartstart_rollout_step(
                ctypes.byref(env.params), ctypes.byref(st), ctypes.byref(navs), ctypes.byref(ss), _ffi.ptr(fb["A"]),
                _ffi.ptr(fb["best"]), float(b.noise_amount), b.seed, b.problem_id0, ctypes.byref(rs),
                ctypes.byref(log_s) if log_s is not None else None,
                ctypes.byref(ring_s) if ring_s is not None else None,
                _ffi.ptr(env.stats), env._seed, env.env_id0, _ffi.ptr(fb["t"]), _ffi.ptr(fb["k"]), _ffi.ptr(fb["ticket"]),
                _ffi.ptr(fb["plan"]), _stream()))

    def _state_tensors(self):
        e, b = self.env, self.nav
        fb = b._fused_buffers(e.device)
        return [e.s0, e.s1, e.steps, e.ep_ret, e.ou_x, e.stats, self.pool.cur_idx, self.pool.plan_of, b.actions_done,
                b.at_goal, self.mode, fb["plan"], fb["t"], fb["k"], self.n_live]

    def rollout(self, K, out, ring=None, graph=True):
        """K steps of every env into the TransitionChunk ``out`` (one graph replay per step when ``graph``)."""
        env, b = self.env, self.nav
        if env._needs_reset:
            env.reset()
        if (out.K, out.N, out.obs_dim) != (K, env.n, env.obs_dim):
            raise ValueError("out chunk has the wrong shape")
        out.step0, out.env_id0 = env.t, env.env_id0
        fb = b._fused_buffers(env.device)
        fb["t"].fill_(env.t)
        fb["k"].zero_()
        fb["plan"].copy_(env.observe())
        self.d_eps.fill_(float(max(self.agent.decaying_ou_action_noise.epsilon, 0.0)))
        self.d_eta.fill_(self.eta)
        self.model.refresh_prepared_image()
        if not graph:
            for _ in range(K):
                self.fused_step(out, ring)
            env.t += K
            return out
        key = (K, out.obs.data_ptr(), None if ring is None else ring.cursor.data_ptr(), self.agent.actor_flat.data_ptr(),
               tuple(w.data_ptr() for w in self.model.W), None if self.model._image is None else self.model._image.data_ptr(),
               self.log_modes)
        if key not in self._graphs:
            snap = [x.clone() for x in self._state_tensors()]
            ring_snap = None if ring is None else ring.cursor.clone()
            side = torch.cuda.Stream(env.device)
            side.wait_stream(torch.cuda.current_stream(env.device))
            with torch.cuda.stream(side):
                self.fused_step(out, ring)           # lazy allocations, weight image, LDS opt-ins outside the capture
            torch.cuda.current_stream(env.device).wait_stream(side)
            for dst, src in zip(self._state_tensors(), snap):
                dst.copy_(src)
            if ring is not None:
                ring.cursor.copy_(ring_snap)
            key = key[:5] + (None if self.model._image is None else self.model._image.data_ptr(),) + key[6:]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.fused_step(out, ring)
            for k_old in list(self._graphs)[:max(0, len(self._graphs) - 3)]:
                del self._graphs[k_old]
            self._graphs[key] = g
        g = self._graphs[key]
        for _ in range(K):
            g.replay()
        env.t += K
        return out

    # ------------------------------------------------------------- dynamics-model aggregation --
    def train_dynamics_model(self, replay, max_rows=65536, n_epoch=1, batchsize=512, lr=0.001, noise_to_signal=0.0):
        """NND_MB_agent.train_dynamics_model (NND_MB_agent.py:437-480) for the vectorised loop: the navigator's model is
        retrained on transitions the envs themselves produced -- an evenly strided sample of up to ``max_rows`` records of
        the device replay ring as (s, a, s2 - s) rows (:442-449), optionally with ``add_noise`` (:451-453), z-scored with
        the model's EXISTING statistics (:455-461) -- ``n_epoch`` passes of Adam steps on the device
        (``DynamicsModel.train_steps``).  Nothing leaves HBM; the packed MFMA weight image is refreshed in place before
        the next rollout, so the captured step graph keeps reading the right bytes.  Returns the mean loss of the last
        epoch (a device -> host read) or None when the ring is still smaller than one batch."""
        from .collect_samples import TrainingSet, add_noise_device
        size = len(replay)
        if size < batchsize:
            return None
        stride = max(1, -(-size // int(max_rows)))
        sel = torch.arange(0, size, stride, device=replay.s.device)
        s, a, s2 = replay.s[sel].contiguous(), replay.a[sel].contiguous(), replay.s2[sel].contiguous()
        dz = (s2 - s).contiguous()
        if noise_to_signal:
            self._agg_calls = getattr(self, "_agg_calls", 0) + 1
            add_noise_device(s, noise_to_signal, self.nav.seed, 2 * self._agg_calls)
            add_noise_device(dz, noise_to_signal, self.nav.seed, 2 * self._agg_calls + 1)
        nm, m = self.model.norm, self.model
        norm = {k: [getattr(nm, k)[i] for i in range(n)] for k, n in (("mean_x", m.state_dim), ("std_x", m.state_dim),
                                                                       ("mean_y", m.act_dim), ("std_y", m.act_dim),
                                                                       ("mean_z", m.state_dim), ("std_z", m.state_dim))}
        X, Z = TrainingSet(s, a, dz, None, None).normalised(norm)
        rows = X.shape[0]
        n_batches = rows // batchsize
        gen = torch.Generator(device=X.device)
        gen.manual_seed(int(self.nav.seed) + 7919 * getattr(self, "_trainings", 0))
        losses = None
        for _ in range(int(n_epoch)):
            perm = torch.randperm(rows, generator=gen, device=X.device)[: n_batches * batchsize]
            losses = self.model.train_steps(X, Z, perm.view(n_batches, batchsize).to(torch.int32), lr=lr)
        self._trainings = getattr(self, "_trainings", 0) + 1
        return None if losses is None else float(losses.mean().item())

    def end_of_generation(self):
        """SmartStartContinuous.end_episode (:372-376) once per episode PER ENV: eta decays."""
        self.eta *= self.eta_decay_factor
