"""The training-loop driver: ``rlTrain`` with the reference's signature and control flow
(smartstart/reinforcementLearningCore/rlTrain.py:13-133) for any ``gym.Env``-shaped env and any
``RLAgent``, plus ``rl_train_vec`` -- the same loop over N device envs in fused chunks.
``Episode`` / ``Summary`` keep the per-episode quantities of smartstart/utilities/datacontainers.py
(:41-49, :61-80, :173-193) and read/write the reference's Summary JSON files (:257-374)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np


class Episode:
    """datacontainers.py:16-80"""

    def __init__(self):
        self.obs, self.action, self.reward, self.obs_tp1, self.done = [], [], [], [], []

    def append(self, obs, action, reward, obs_tp1, done):
        self.obs.append(np.ravel(obs).tolist())
        self.action.append(np.ravel(action).tolist())
        self.reward.append(float(reward))
        self.obs_tp1.append(np.ravel(obs_tp1).tolist())
        self.done.append(bool(done))

    def total_reward(self):
        return sum(self.reward)

    def average_reward(self):
        return sum(self.reward) / max(len(self.reward), 1)

    def path(self):
        return self.obs + [self.obs_tp1[-1]] if self.obs else []

    def __len__(self):
        return len(self.reward)


class Summary:
    """smartstart/utilities/datacontainers.py:130-374: per-episode (steps, total_reward), best path, the
    last ``last_x`` paths, SmartStart episode indices and the agent's hyper-parameters -- with the
    reference's attribute names, so that ``to_json`` / ``save`` / ``load`` read and write the same JSON
    files as the reference (``data/**/*.json``); the GCS upload is out of scope."""

    def __init__(self, name=None, last_x=5):
        self.name = name
        self._pending = []      # (lengths, returns) numpy chunks from the fused rollouts, not yet turned into tuples
        self.episodes = []
        self.best_path = None
        self.best_reward = -sys.maxsize                      # :163
        self.last_x = max(last_x, 1)
        self.last_paths = [[None]] * last_x                  # :167 (oldest first)
        self.last_rewards = [None] * last_x
        self.smart_start_episodes = []
        self.name_of_agent = ""
        self.param_dict = {}

    def set_agent(self, agent):
        """:173-176"""
        self.name_of_agent = str(agent.__class__.__name__)
        try:
            pd = agent.get_param_dict()
        except NotImplementedError:
            pd = None
        if pd is not None:
            self.param_dict = {**pd, **self.param_dict}

    def add_params_to_param_dict(self, **kwargs):
        self.param_dict = {**kwargs, **self.param_dict}

    def start_smart_start_episode(self):
        self.smart_start_episodes.append(len(self.episodes))

    def append(self, episode):
        """:180-203"""
        total = episode.total_reward()
        self.episodes.append((len(episode), total))
        if total > self.best_reward:
            self.best_path, self.best_reward = episode.path(), total
        self.last_paths = self.last_paths[1:] + [episode.path()]
        self.last_rewards = self.last_rewards[1:] + [total]

    def append_record(self, length, total_reward):
        """A finished episode reported by the fused rollout kernel's episode ring (no path)."""
        self.episodes.append((int(length), float(total_reward)))
        if total_reward > self.best_reward:
            self.best_reward = float(total_reward)

    def extend_records(self, lengths, total_rewards):
        """Many finished episodes at once (numpy arrays from the episode ring).  A 65 536-env chunk can finish
        65 536 of them; building that many Python tuples per chunk would dominate the loop, so the arrays are
        kept and turned into the reference's list of ``(steps, total_reward)`` tuples when ``episodes`` is read."""
        if len(lengths) == 0:
            return
        self._pending.append((np.asarray(lengths).copy(), np.asarray(total_rewards).copy()))
        best = float(total_rewards.max())
        if best > self.best_reward:
            self.best_reward = best

    @property
    def episodes(self):
        if self._pending:
            pend, self._pending = self._pending, []
            for lengths, rets in pend:
                self._episodes.extend(zip(lengths.tolist(), rets.tolist()))
        return self._episodes

    @episodes.setter
    def episodes(self, value):
        self._pending = []
        self._episodes = value

    # ---- accessors (:205-255) ---------------------------------------------------------------------
    def total_episode_reward(self):
        return [reward for _, reward in self.episodes]

    def total_reward(self):
        return sum(self.total_episode_reward())

    def average_reward(self):
        return self.total_reward() / len(self)

    def average_episode_reward(self):
        return [reward / steps for steps, reward in self.episodes]

    def steps_episode(self):
        return [steps for steps, _ in self.episodes]

    def get_best_path_and_reward(self):
        return self.best_path, self.best_reward

    def get_last_path(self, x):
        return self.last_paths[-(x + 1)]

    def get_last_reward(self, x):
        return self.last_rewards[-(x + 1)]

    def __len__(self):
        return len(self._episodes) + sum(len(l) for l, _ in self._pending)

    # ---- persistence (:257-374) ---------------------------------------------------------------------
    def to_json(self):
        def plain(o):
            return o.tolist() if isinstance(o, np.ndarray) else float(o) if isinstance(o, np.floating) else \
                int(o) if isinstance(o, np.integer) else str(o)
        d = {k: v for k, v in self.__dict__.items() if k not in ("_episodes", "_pending")}
        d["episodes"] = self.episodes
        return json.dumps(d, default=plain)

    @classmethod
    def from_json(cls, data):
        summary = cls()
        fields = json.loads(data)
        summary.episodes = fields.pop("episodes", [])
        summary.__dict__.update(fields)
        return summary

    def save(self, directory=".", post_fix=0, extra_name_append="", last_name_section=False):
        """:288-326: <name><extra>_<post_fix>.json, post_fix auto-incremented past existing files."""
        def make_name(pf):
            name = self.name.split("_")[-1] if last_name_section else self.name
            return os.path.join(directory, name + extra_name_append + "_" + str(pf) + ".json")
        fp = make_name(post_fix)
        while os.path.exists(fp):
            post_fix += 1
            fp = make_name(post_fix)
        with open(fp, "x") as f:
            f.write(self.to_json())
        return fp

    @classmethod
    def load(cls, fp):
        with open(fp, "r") as f:
            return cls.from_json(f.read())


def rlTrain(agent, env, render=False, render_episode=False, print_results=True, print_steps=True,
            num_episodes=500, max_steps=1000, progress_bar=False, id=None, num_ticks=100, print_time=False):
    """rlTrain.py:13-133, minus the console plotting."""
    name = agent.get_summary_name() if hasattr(agent, 'get_summary_name') else agent.__class__.__name__
    spec_id = env.spec.id if getattr(env, "spec", None) is not None else type(env).__name__
    summary = Summary(name + "_" + spec_id)                      # :50-55
    summary.set_agent(agent)
    times = [time.time()]
    for i_episode in range(num_episodes):                        # :63
        episode = Episode()
        observation = env.reset()                                # :68
        agent.start_new_episode(observation)                     # :71
        if getattr(agent, "smart_start_pathing", False):
            summary.start_smart_start_episode()
        for step in range(max_steps):                            # :75
            if render:
                render = agent.render(env)
            action = agent.get_action(observation)               # :81
            new_observation, reward, done, _ = env.step(action)  # :84
            if print_steps:
                print("        Step: {}, State: {}, Action: {}, New_State: {}, Reward: {}".format(
                    step, observation, action, new_observation, reward))
            agent.observe(observation, action, reward, new_observation, done)   # :91
            episode.append(observation, action, reward, new_observation, done)  # :94
            if done:
                break
            observation = new_observation
        agent.end_episode()                                      # :101
        if print_results:
            print("Episode: %d, steps: %d, reward: %.2f" % (i_episode, len(episode), episode.average_reward()))
        summary.append(episode)                                  # :114
        if print_time:
            times.append(time.time())
            print("took: " + str(times[-1] - times[-2]))
    return summary


def rl_train_vec(env, policy, num_chunks, chunk_steps=1024, ring_capacity=1 << 20, on_chunk=None):
    """The rlTrain loop for N parallel device envs: ``num_chunks`` fused rollouts of ``chunk_steps``
    steps each (auto-reset on done); every finished episode lands in the Summary as (len, return).
    ``on_chunk(chunk, env)`` -- if given -- sees each TransitionChunk (e.g. to feed a replay buffer)."""
    from .vec_env import EpisodeRing, TransitionChunk
    summary = Summary("vec_" + env.spec.id)
    ring = EpisodeRing(ring_capacity, env.device)
    chunk = TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device) if on_chunk is not None else None
    pd = env.policy_desc(policy)
    dropped = 0
    for _ in range(num_chunks):
        out = env.rollout(chunk_steps, out=chunk, ring=ring, log=on_chunk is not None, policy_desc=pd)
        if on_chunk is not None:
            on_chunk(out, env)
        (ids, lens, rets), d = ring.drain()
        dropped += d
        summary.extend_records(lens, rets)
    summary.dropped_episode_records = dropped
    return summary


class DecaySchedule:
    """The per-episode decays of the reference -- ``reduce_epsilon`` (DDPG_Baselines_agent.py:77-78, from end_episode
    :255-258) and ``reduce_eta`` (smartexplorationcontinuous.py:372-376) -- for a loop over n parallel envs, kept on the
    DEVICE: one decay per GENERATION = ``per_generation`` finished episodes (once per episode per env), computed by
    ``ssc_decay_schedule`` from the finished-episode counter the rollout kernel accumulates (``env.stats[3]``).  The
    arithmetic is the host's (fp64 ``value = max(value * factor, floor)``, one application per generation), so the
    values equal ``reduce_epsilon()`` called ``generations`` times -- but no chunk boundary needs the host any more.

    ``entries``: list of (start value, factor, floor, out) with ``out`` a 1-element fp32 device tensor (or None) that
    receives the current value -- e.g. ``agent.d_epsilon``, which ``agent.as_policy(device_epsilon=True)`` hands to the
    rollout kernel."""

    def __init__(self, device, per_generation, entries):
        import ctypes
        import torch
        from . import _ffi
        self._ffi, self._ct, self._torch = _ffi, ctypes, torch
        self.device = torch.device(device)
        self.per_generation = float(per_generation)
        self.n = len(entries)
        if not 1 <= self.n <= 4:
            raise ValueError("1..4 schedules")
        self._factor = (ctypes.c_double * self.n)(*[float(e[1]) for e in entries])
        self._floor = (ctypes.c_double * self.n)(*[float(e[2]) for e in entries])
        self.state = torch.tensor([0.0] + [float(e[0]) for e in entries], dtype=torch.float64, device=self.device)
        self.outs = [e[3] for e in entries]
        self._out_ptrs = (ctypes.c_void_p * self.n)(*[None if o is None else o.data_ptr() for o in self.outs])
        for e in entries:
            if e[3] is not None:
                e[3].fill_(float(e[0]))

    def update(self, finished, finished0=0.0):
        """Enqueue the decay on the current stream; ``finished``: device fp64 view of the finished-episode counter,
        ``finished0``: what it read when the loop started."""
        ffi, ct, torch = self._ffi, self._ct, self._torch
        with torch.cuda.device(self.device):
            ffi.check(ffi.lib().ssc_decay_schedule(ffi.ptr(finished), float(finished0), self.per_generation, self.n, self._factor, self._floor,
                                                   ffi.ptr(self.state), self._out_ptrs,
                                                   ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def read(self):
        """(generations applied, [current values]) -- a device -> host read (synchronises the current stream)."""
        st = self.state.cpu().tolist()
        return int(st[0]), st[1:]


def epsilon_schedule(agent, per_generation, device=None):
    """DecaySchedule of one agent's OU epsilon, starting from its current host value and writing ``agent.d_epsilon``."""
    n = agent.decaying_ou_action_noise
    return DecaySchedule(device if device is not None else agent.device, per_generation,
                         [(float(n.epsilon), n.epsilon_decay_factor, n.min_epsilon, agent.d_epsilon)])


def default_drain_every(ring_capacity, n_envs, chunk_steps, shortest_episode=64):
    """How many chunks an episode ring of ``ring_capacity`` records holds when no episode is shorter than
    ``shortest_episode`` steps (clamped to 1..16)."""
    per_chunk = n_envs * -(-int(chunk_steps) // shortest_episode)
    return int(max(1, min(16, ring_capacity // max(per_chunk, 1))))


def rl_train_vec_ddpg(env, agent, num_chunks, chunk_steps=256, replay_capacity=1 << 20, train_iters=None,
                      replay_last_steps=None, seed=0, ring_capacity=1 << 20, track_episodes=False, overlap=False,
                      drain_every=None, on_chunk=None):
    """Actor-learner loop entirely in HBM: every chunk is a fused rollout of ``chunk_steps`` steps of all
    ``env.n`` envs under the agent's current actor (+ OU noise), appended to a device replay ring, followed
    by ``train_iters`` DDPG iterations (default ``agent.num_train_iterations``) on batches drawn from it.
    The vectorised counterpart of rlTrain + DDPG_Baselines_agent.observe/train
    (rlTrain.py:75-100, DDPG_Baselines_agent.py:238-273).  ``track_episodes`` keeps the episode index in the device
    ring as well, so that ``smartstart.device_smart_start_path(replay, agent, radii, n_ss)`` can pick a smart-start
    state and recover the path to it without the replay contents leaving HBM.

    Nothing at a chunk boundary needs the host: epsilon decays once per generation of finished episodes ON THE DEVICE
    (:class:`DecaySchedule` behind each rollout, the kernel reads ``agent.d_epsilon``), and the finished-episode records
    are read back every ``drain_every`` chunks only (default: as many chunks as the episode ring holds,
    :func:`default_drain_every`; a ring that overflows drops records and counts them in
    ``summary.dropped_episode_records``).  The host therefore runs ahead of the GPU and the chunk costs what its
    kernels cost.  The host-side ``agent.decaying_ou_action_noise.epsilon`` is brought up to date when the loop ends.

    ``overlap=True`` rolls chunk i+1 on a second stream WHILE the learner runs its iterations on chunk i: the rollout
    then acts with a snapshot of the actor taken before those iterations (one chunk stale -- the single-GPU form of
    ``rl_train_sharded_ddpg(pipelined=True)``); everything else (replay contents, epsilon decay per finished
    generation, sampling) is ordered as in the synchronous loop, and the result is deterministic.  The default follows
    the reference's order: act with the weights of the last completed train().
    ``on_chunk(i, chunk, env)`` -- if given -- is called once chunk i's rollout, append and learner iterations are enqueued.
    Returns (Summary, losses per chunk, replay)."""
    import torch
    from .replay_buffer import DeviceReplayBuffer
    from .vec_env import EpisodeRing, TransitionChunk
    summary = Summary("vec_ddpg_" + env.spec.id)
    ring = EpisodeRing(ring_capacity, env.device)
    chunk = TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device)
    replay = DeviceReplayBuffer(replay_capacity, env.obs_dim, 1, env.device, seed=seed, track_episodes=track_episodes,
                                n_envs=env.n, max_path_len=(env.spec.max_episode_steps or 1000) + 1)
    losses = []
    if drain_every is None:
        drain_every = default_drain_every(ring_capacity, env.n, chunk_steps)
    schedule = epsilon_schedule(agent, env.n, env.device)   # once per episode PER ENV (DDPG_Baselines_agent.py:255-258)
    finished, finished0 = env.stats[3:4], float(env.stats[3].item())   # the env may have run before: count from here
    dropped = 0

    def after_rollout():
        schedule.update(finished, finished0)

    def drain():
        nonlocal dropped
        (ids, lens, rets), d = ring.drain()
        dropped += d
        summary.extend_records(lens, rets)

    def finish():
        drain()
        summary.dropped_episode_records = dropped
        _g, (eps,) = schedule.read()
        agent.decaying_ou_action_noise.epsilon = eps

    if overlap:
        from .vec_env import ActorPolicy
        cur = torch.cuda.current_stream(env.device)
        with torch.cuda.device(env.device):              # env.device need not be the process's current device
            act = torch.cuda.Stream(env.device)          # the rollouts' stream
        chunks = [chunk, TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device)]
        snap = agent.actor_flat.clone()                  # the weights the NEXT rollout acts with
        views, o = {}, 0
        for k, v in agent.weights.items():               # same order and shapes as the flat array (flatten_params)
            views[k] = snap[o:o + v.numel()].view(v.shape)
            o += v.numel()
        rolled2 = [torch.cuda.Event(), torch.cuda.Event()]
        appended = [torch.cuda.Event(), torch.cuda.Event()]
        snap_ready = torch.cuda.Event()
        live = agent.as_policy(device_epsilon=True)
        pd = env.policy_desc(ActorPolicy(views, last_layer_tanh=live.last_layer_tanh, precision=live.precision, ou_mu=live.ou_mu,
                                         ou_sigma=live.ou_sigma, ou_theta=live.ou_theta, ou_dt=live.ou_dt,
                                         obs_clip=live.obs_clip, d_ou_epsilon=agent.d_epsilon))
        act.wait_stream(cur)

        drained = torch.cuda.Event()

        def launch_rollout(b, wait_for):
            for e in wait_for:
                act.wait_event(e)
            with torch.cuda.stream(act):
                env.rollout(chunk_steps, out=chunks[b], ring=ring, policy_desc=pd)
                after_rollout()
                rolled2[b].record(act)

        snap_ready.record(cur)
        launch_rollout(0, [snap_ready])
        for i in range(num_chunks):
            b = i & 1
            cur.wait_event(rolled2[b])
            replay.append_chunk(chunks[b], reward_scale=agent.reward_scale, last_steps=replay_last_steps)
            appended[b].record(cur)
            snap.copy_(agent.actor_flat)                 # behind train i-1 on this stream, in front of train i
            snap_ready.record(cur)
            l = agent.train_from(replay, train_iters)
            if l is not None:
                losses.append(l)
            if on_chunk is not None:
                on_chunk(i, chunks[b], env)
            extra = []
            if (i + 1) % drain_every == 0:               # the ring holds the records of chunks <= i; rollout i+1 is not queued yet
                drain()
                drained.record(cur)                      # ... and must not start before the cursor reset
                extra = [drained]
            if i + 1 < num_chunks:                       # chunk buffer b^1 was last read by append i-1
                launch_rollout(b ^ 1, [snap_ready] + extra + ([appended[b ^ 1]] if i >= 1 else []))
        cur.wait_stream(act)
        finish()
        return summary, losses, replay

    pd = env.policy_desc(agent.as_policy(device_epsilon=True))   # weights are views into the flat parameter arrays
    for i in range(num_chunks):
        out = env.rollout(chunk_steps, out=chunk, ring=ring, policy_desc=pd)
        after_rollout()
        replay.append_chunk(out, reward_scale=agent.reward_scale, last_steps=replay_last_steps)
        l = agent.train_from(replay, train_iters)
        if l is not None:
            losses.append(l)
        if on_chunk is not None:
            on_chunk(i, out, env)
        if (i + 1) % drain_every == 0:
            drain()
    finish()
    return summary, losses, replay


def rl_train_vec_smartstart(env, smart, num_chunks, chunk_steps=64, replay_capacity=1 << 20, train_iters=None,
                            replay_last_steps=None, seed=0, ring_capacity=1 << 20, refresh_every=1, graph=True,
                            on_chunk=None, aggregate_every=0, aggregate_kwargs=None, overlap_selection=False):
    """The vectorised SmartStart loop: rlTrain (rlTrain.py:63-114) with ``SmartStartContinuous(DDPG_Baselines_agent)``
    (smartexplorationcontinuous.py:307-376) for all envs of ``env`` at once, everything in HBM.  Per chunk:
    smart-start selection on the device replay ring -> plans on offer (``smart.refresh_plans``, every ``refresh_every``
    chunks), ``chunk_steps`` steps of every env in its own mode (``smart.rollout``), the chunk appended to the ring
    (episode index kept on the device), ``train_iters`` DDPG iterations, and the finished episodes read back for the
    epsilon / eta decay (once per episode per env, like DDPG_Baselines_agent.end_episode :255-258 and
    SmartStartContinuous.end_episode :372-376).  ``aggregate_every`` > 0 retrains the navigator's dynamics model on the
    ring every that many chunks (``smart.train_dynamics_model``: the reference does it every
    ``num_episodes_for_aggregation`` planned episodes, NND_MB_agent.py:420-423).  ``smart``: :class:`smartstart.VecSmartStart`.

    ``overlap_selection=True``: the selection of chunk c runs WHILE chunk c rolls -- its kernels (candidates, Q(s, pi(s)),
    KDE, UCB) on a side stream that fills the gaps between the rollout's launches, its host part (episodic paths, path
    shortcutting, waypoints) while the GPU works through the queued steps -- and the plans it produces go on offer for chunk
    c + 1 (published on the rollout's stream behind chunk c: deterministic).  The selection then sees the ring as of chunk
    c - 1 and its plans are used one chunk later than in the sequential loop; the chunk costs max(rollout, selection)
    instead of their sum.
    Returns (Summary, losses per chunk, replay)."""
    import torch
    from .replay_buffer import DeviceReplayBuffer
    from .vec_env import EpisodeRing, TransitionChunk
    agent = smart.agent
    summary = Summary("vec_smartstart_" + env.spec.id)
    ring = EpisodeRing(ring_capacity, env.device)
    chunk = TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device)
    replay = DeviceReplayBuffer(replay_capacity, env.obs_dim, 1, env.device, seed=seed, track_episodes=True,
                                n_envs=env.n, max_path_len=(env.spec.max_episode_steps or 1000) + 1)
    losses, generations = [], 0.0
    if overlap_selection:
        main = torch.cuda.current_stream(env.device)
        with torch.cuda.device(env.device):
            side = torch.cuda.Stream(env.device)
        learned = torch.cuda.Event()
    for c in range(num_chunks):
        if aggregate_every and c > 0 and c % aggregate_every == 0:
            smart.train_dynamics_model(replay, **(aggregate_kwargs or {}))
        plans = None
        if overlap_selection:
            out = smart.rollout(chunk_steps, chunk, ring=ring, graph=graph)         # chunk c is queued; the host is free
            if c % refresh_every == 0 and c > 0:
                side.wait_event(learned)                                            # the ring and the networks as of chunk c - 1
                with torch.cuda.stream(side):
                    plans = smart.select_plans(replay)                              # kernels on the side stream, host geometry meanwhile
                main.wait_stream(side)                                              # (already complete: the host read its results)
            if plans:
                smart.publish_plans(plans)                                          # main stream, behind chunk c: on offer from chunk c + 1
        else:
            if c % refresh_every == 0:
                smart.refresh_plans(replay)
            out = smart.rollout(chunk_steps, chunk, ring=ring, graph=graph)
        replay.append_chunk(out, reward_scale=agent.reward_scale, last_steps=replay_last_steps)
        l = agent.train_from(replay, train_iters)
        if overlap_selection:
            learned.record(main)
        if l is not None:
            losses.append(l)
        if on_chunk is not None:
            on_chunk(c, out, smart)
        (ids, lens, rets), _d = ring.drain()
        summary.extend_records(lens, rets)
        generations += len(lens) / float(env.n)
        while generations >= 1.0:
            agent.decaying_ou_action_noise.reduce_epsilon()
            smart.end_of_generation()
            generations -= 1.0
    return summary, losses, replay
