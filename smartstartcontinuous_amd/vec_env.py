"""Vectorised environments over the HIP kernels, keeping the ``gym.Env`` surface.

``VecEnv`` is the batched counterpart of the reference environments
(smartstart/environments/continuous_mountain_car_editted.py and ``gym.make(...)``):
``reset() / step() / seed() / close()`` plus ``observation_space / action_space / spec.id``,
with [N, ...] torch tensors on the GPU instead of scalars.  ``SingleEnvView`` presents an
N=1 ``VecEnv`` with *exactly* the scalar tuple types of gym 0.10.5 so that an unchanged
``rlTrain``-style loop and any ``RLAgent`` can drive it (BASELINE config 1).

State lives in caller-visible torch tensors (structure of arrays, fp32); every call goes
through the C ABI in ``include/ssc.h`` on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes
import re
from dataclasses import dataclass

import numpy as np
import torch

from . import _ffi
from .spaces import Box, EnvSpec

RESET_T0 = (1 << 56) - 1  # RNG step index of the very first reset (oracle/ssc_oracle.py:RESET_T0)

_MC_EDITTED = re.compile(r"^MountainCarContinuousActionX([0-9.]+)-v(\d+)$")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def parse_env_id(env_id, power_scalar=None, max_episode_steps=None):
    """-> (kind, power_scalar, max_episode_steps, spec_id).

    'MountainCarContinuous-v0'  : stock gym env, TimeLimit 999
    'MountainCarContinuousActionX<ps>-v0' : Continuous_MountainCarEnv_Editted.get_name()
        (continuous_mountain_car_editted.py:151-152); the examples wrap it in TimeLimit(1000)
    'Pendulum-v0' : gym 0.10.5, TimeLimit 200 [third-party]; theta integrates the UNCLIPPED new velocity
    'Pendulum-v1' : the modern update order (velocity clipped before theta integrates) -- the id BASELINE
        configs[3] names; VecEnv sets ``params.pend_v1_order`` from the id
    """
    if env_id == "MountainCarContinuous-v0":
        ps = 1.0 if power_scalar is None else power_scalar
        return _ffi.SSC_ENV_MOUNTAINCAR, ps, 999 if max_episode_steps is None else max_episode_steps, env_id
    m = _MC_EDITTED.match(env_id)
    if m:
        ps = float(m.group(1)) if power_scalar is None else power_scalar
        return _ffi.SSC_ENV_MOUNTAINCAR, ps, 1000 if max_episode_steps is None else max_episode_steps, env_id
    if env_id in ("Pendulum-v0", "Pendulum-v1"):
        return _ffi.SSC_ENV_PENDULUM, 1.0, 200 if max_episode_steps is None else max_episode_steps, env_id
    raise ValueError(f"unknown environment id {env_id!r}")


@dataclass
class RandomPolicy:
    """Policy_Random (NN_Dynamics_Model/policy_random.py:3-15): a ~ U(action_space.low, high)."""


@dataclass
class ActorPolicy:
    """DDPG_Baselines_agent.get_action (smartstart/RLAgents/DDPG_Baselines_agent.py:206-234):
    actor forward + epsilon * OU noise + clip + scale.  ``weights`` = dict of fp32 CUDA
    tensors W1[obs,h1] b1 W2[h1,h2] b2 W3[h2,1] b3 in TensorFlow layout."""
    weights: dict
    last_layer_tanh: bool = True
    precision: str = "bf16_mfma"       # "f32" | "bf16_mfma"
    ou_mu: float = 0.4
    ou_sigma: float = 0.6
    ou_theta: float = 0.15
    ou_dt: float = 1e-2
    ou_epsilon: float = 1.0
    obs_clip: float = 0.0              # > 0: the actor sees clip(obs, -obs_clip, obs_clip) (ddpg_editted.py:106-109: 5.0)
    d_ou_epsilon: object = None        # a 1-element fp32 device tensor: the kernel reads epsilon from it (ou_epsilon is
                                       # ignored) -- rl_train.DecaySchedule keeps it current without the host


@dataclass
class MpcPolicy:
    """The SmartStart navigator as a rollout policy (NND_MB_agent.get_action, NND_MB_agent.py:339-358):
    ``navigators`` is a :class:`smartstartcontinuous_amd.navigator.NavigatorBatch` with one problem per env.
    Unlike RANDOM / ACTOR this policy is a chain of launches per step -- forward sim (drawing its own candidates), score (2), and one
    fused launch for action / env step / log / statistics / waypoint advance / auto-reset -- whose step index and
    log row are device counters: the chain is captured once as a HIP graph and replayed K times per chunk
    (``graph=False``: the step-by-step reference path through the single-purpose entry points)."""
    navigators: object
    graph: bool = True


class TransitionChunk:
    """Transition log of one rollout chunk -- the (s, a, r, t, s2) records of
    ReplayBuffer.add (smartstart/RLAgents/replay_buffer.py:49-74) as SoA columns.

    obs, obs2: [obs_dim, K, N] fp32; act, rew: [K, N] fp32; done: [K, N] uint8."""

    def __init__(self, obs_dim, K, N, device, packed=True):
        """``packed=True`` (default) stores the 2*obs_dim+2 fp32 columns of one step side by side
        ([K][cols][N] -- one contiguous region per step, fewer concurrent HBM write streams); the
        column attributes are then strided views.  ``packed=False`` gives dense [K][N] columns."""
        ncol = 2 * obs_dim + 2
        self.packed = bool(packed)
        if self.packed:
            self._buf = torch.empty((K, ncol, N), dtype=torch.float32, device=device)
            cols = self._buf.permute(1, 0, 2)                      # [ncol, K, N] view, row stride ncol*N
            self.obs, self.act, self.rew = cols[:obs_dim], cols[obs_dim], cols[obs_dim + 1]
            self.obs2 = cols[obs_dim + 2:]
        else:
            self.obs = torch.empty((obs_dim, K, N), dtype=torch.float32, device=device)
            self.obs2 = torch.empty((obs_dim, K, N), dtype=torch.float32, device=device)
            self.act = torch.empty((K, N), dtype=torch.float32, device=device)
            self.rew = torch.empty((K, N), dtype=torch.float32, device=device)
        self.done = torch.empty((K, N), dtype=torch.uint8, device=device)
        self.K, self.N, self.obs_dim = K, N, obs_dim
        self.step0 = 0
        self.env_id0 = 0

    BYTES_PER_STEP = {2: 25, 3: 33}

    @classmethod
    def from_columns(cls, obs, act, rew, obs2, done):
        """A chunk over EXISTING column tensors (obs/obs2 [obs_dim, K, N], act/rew/done [K, N]; rows of one column
        equally strided) -- e.g. the records a learner rank received from another rank (TransitionGather.unpack)."""
        self = cls.__new__(cls)
        self.packed = False
        self.obs, self.act, self.rew, self.obs2, self.done = obs, act, rew, obs2, done
        self.obs_dim, self.K, self.N = obs.shape[0], act.shape[0], act.shape[1]
        self.step0 = self.env_id0 = 0
        return self

    def nbytes(self):
        return self.K * self.N * (8 * self.obs_dim + 9)

    def columns(self):
        """All columns as one list (used by the multi-GPU gather)."""
        return [self.obs, self.act, self.rew, self.done, self.obs2]

    def as_struct(self):
        log = _ffi.TransitionLog()
        for c in range(self.obs_dim):
            log.obs[c] = self.obs[c].data_ptr()
            log.obs2[c] = self.obs2[c].data_ptr()
        log.act = self.act.data_ptr()
        log.rew = self.rew.data_ptr()
        log.done = self.done.data_ptr()
        log.row_stride = self.act.stride(0)
        log.done_row_stride = self.done.stride(0)
        return log

    def records(self):
        """(s[KN,obs], a[KN,1], r[KN], t[KN] bool, s2[KN,obs]) in step-major order."""
        s = self.obs.reshape(self.obs_dim, -1).t()
        s2 = self.obs2.reshape(self.obs_dim, -1).t()
        return s, self.act.reshape(-1, 1), self.rew.reshape(-1), self.done.reshape(-1).bool(), s2


class EpisodeRing:
    """Completed-episode records (len, total_reward) -- Summary.append
    (smartstart/utilities/datacontainers.py:173-193)."""

    def __init__(self, capacity, device):
        self.env_id = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.length = torch.zeros(capacity, dtype=torch.int32, device=device)
        self.ret = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        self.capacity = capacity

    def as_struct(self):
        r = _ffi.EpisodeRing()
        r.env_id, r.length, r.ret = self.env_id.data_ptr(), self.length.data_ptr(), self.ret.data_ptr()
        r.cursor, r.capacity = self.cursor.data_ptr(), self.capacity
        return r

    def drain(self):
        """-> (env_id, length, ret) numpy arrays of the recorded episodes, then clears.
        Returns also the number of dropped records."""
        n = int(self.cursor.item()) & 0xFFFFFFFF
        kept = min(n, self.capacity)
        if kept == 0:      # (a device -> host copy call costs ~50 us of host time even for zero bytes)
            return (np.empty(0, np.int64), np.empty(0, np.int32), np.empty(0, np.float32)), n
        out = (self.env_id[:kept].cpu().numpy(), self.length[:kept].cpu().numpy(), self.ret[:kept].cpu().numpy())
        self.cursor.zero_()
        return out, n - kept

    def drain_overlapped(self, side, ready):
        """drain() that waits for ``ready`` only (an event recorded right behind the rollout that filled the ring)
        instead of for everything the caller has enqueued since: the readback and the cursor reset run on the stream
        ``side``, so the learner iterations queued behind the rollout keep the GPU busy while the host reads the episode
        records.  -> (records, dropped, done_event); the stream that launches the next rollout must wait for
        ``done_event`` (the ring is written again by that rollout)."""
        if getattr(self, "_drained", None) is None:
            self._drained = torch.cuda.Event()     # reused: creating a HIP event per chunk costs tens of microseconds
        with torch.cuda.stream(side):
            side.wait_event(ready)
            out, dropped = self.drain()
            self._drained.record(side)
        return out, dropped, self._drained


class VecEnv:
    """N independent environments on one GPU.

    Parameters mirror the reference: ``env_id`` is what ``env.spec.id`` reports;
    ``power_scalar`` / ``max_episode_steps`` as in
    ``Continuous_MountainCarEnv_Editted.make_timed_env`` (continuous_mountain_car_editted.py:154-159).
    ``env_id0`` is the global id of env 0 (multi-GPU sharding keeps one id space).
    """

    def __init__(self, env_id="MountainCarContinuous-v0", n_envs=1, device="cuda", power_scalar=None,
                 max_episode_steps=None, seed=1234, env_id0=0, auto_reset=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("VecEnv runs on the GPU only (device='cuda'); there is no CPU fallback")
        self.lib = _ffi.lib()
        kind, ps, mes, spec_id = parse_env_id(env_id, power_scalar, max_episode_steps)
        self.kind, self.power_scalar, self.n = kind, ps, int(n_envs)
        self.params = _ffi.default_params(kind, ps, mes or 0)
        if kind == _ffi.SSC_ENV_PENDULUM:
            self.params.pend_v1_order = 1 if spec_id == "Pendulum-v1" else 0
        self.spec = EnvSpec(spec_id, mes)
        self.env_id0 = int(env_id0)
        self.auto_reset = auto_reset
        if kind == _ffi.SSC_ENV_MOUNTAINCAR:
            self.obs_dim = 2
            # continuous_mountain_car_editted.py:45-51
            self.action_space = Box(low=-1.0, high=1.0, shape=(1,))
            self.observation_space = Box(low=np.array([-1.2, -0.07]), high=np.array([0.6, 0.07]))
        else:
            self.obs_dim = 3
            self.action_space = Box(low=-2.0, high=2.0, shape=(1,))
            self.observation_space = Box(low=np.array([-1.0, -1.0, -8.0]), high=np.array([1.0, 1.0, 8.0]))
        dev, n = self.device, self.n
        self.s0 = torch.zeros(n, dtype=torch.float32, device=dev)
        self.s1 = torch.zeros(n, dtype=torch.float32, device=dev)
        self.steps = torch.zeros(n, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ou_x = torch.zeros(n, dtype=torch.float32, device=dev)
        self._obs = torch.empty((self.obs_dim, n), dtype=torch.float32, device=dev)
        self._rew = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self._seed = int(seed)
        self.t = 0            # global step index (RNG counter)
        self._needs_reset = True

    # ------------------------------------------------------------------ gym.Env surface --
    def seed(self, seed=None):
        """gym ``Env.seed`` (continuous_mountain_car_editted.py:56-58): returns [seed]."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy & ((1 << 63) - 1))
        self._seed = int(seed)
        return [self._seed]

    def reset(self, mask=None):
        """Reset all envs (``mask=None``) or those with ``mask[i] != 0``; returns obs [N, obs_dim]."""
        with torch.cuda.device(self.device):
            t = RESET_T0 if (mask is None and self._needs_reset) else max(self.t - 1, 0)
            if mask is not None:
                mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            _ffi.check(self.lib.ssc_env_reset(ctypes.byref(self.params), self.n, _ffi.ptr(mask), _ffi.ptr(self.s0),
                                              _ffi.ptr(self.s1), _ffi.ptr(self.steps), _ffi.ptr(self.ep_ret),
                                              _ffi.ptr(self.ou_x), self._seed, self.env_id0, t, _stream()))
            if mask is None:
                self._needs_reset = False
            return self.observe()

    def observe(self):
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.ssc_env_observe(ctypes.byref(self.params), self.n, _ffi.ptr(self.s0),
                                                _ffi.ptr(self.s1), _ffi.ptr(self._obs), _stream()))
        return self._obs.t()

    def step(self, actions):
        """actions [N, act_dim] (or [N]) -> (obs [N, obs_dim], rew [N], done [N] bool, {})."""
        self._step_raw(actions)
        obs, rew, done = self._obs.t().clone(), self._rew.clone(), self._done.bool()
        if self.auto_reset:
            self.reset(mask=self._done)
        return obs, rew, done, {}

    def _step_raw(self, actions):
        """The launches of ``step``; the results stay in ``self._obs`` [obs_dim, N], ``self._rew``, ``self._done``."""
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(-1).contiguous()
        if a.numel() != self.n:
            raise ValueError(f"expected {self.n} actions, got {a.numel()}")
        with torch.cuda.device(self.device):
            if self.kind == _ffi.SSC_ENV_MOUNTAINCAR:
                _ffi.check(self.lib.ssc_mc_step(ctypes.byref(self.params), self.n, _ffi.ptr(self.s0),
                                                _ffi.ptr(self.s1), _ffi.ptr(a), _ffi.ptr(self._rew),
                                                _ffi.ptr(self._done), _ffi.ptr(self.steps), _stream()))
                _ffi.check(self.lib.ssc_env_observe(ctypes.byref(self.params), self.n, _ffi.ptr(self.s0),
                                                    _ffi.ptr(self.s1), _ffi.ptr(self._obs), _stream()))
            else:
                _ffi.check(self.lib.ssc_pend_step(ctypes.byref(self.params), self.n, _ffi.ptr(self.s0),
                                                  _ffi.ptr(self.s1), _ffi.ptr(a), _ffi.ptr(self._obs),
                                                  _ffi.ptr(self._rew), _ffi.ptr(self._done), _ffi.ptr(self.steps),
                                                  _stream()))
        self.t += 1

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope (SURVEY.md section 8: visualisation)")

    def close(self):
        pass

    # --------------------------------------------------------------------- fused rollout --
    def policy_desc(self, policy):
        pd = _ffi.PolicyDesc()
        pd.act_low = float(self.action_space.low[0])
        pd.act_high = float(self.action_space.high[0])
        if isinstance(policy, RandomPolicy) or policy in ("random", None):
            pd.kind = _ffi.SSC_POLICY_RANDOM
            return pd, None
        if not isinstance(policy, ActorPolicy):
            raise TypeError("policy must be RandomPolicy() or ActorPolicy(...)")
        w = {k: v.to(device=self.device, dtype=torch.float32).contiguous() for k, v in policy.weights.items()}
        obs_dim, h1 = w["W1"].shape
        h2 = w["W2"].shape[1]
        if obs_dim != self.obs_dim or w["W2"].shape[0] != h1 or w["W3"].shape != (h2, 1):
            raise ValueError("actor weight shapes do not match the environment")
        pd.kind = _ffi.SSC_POLICY_ACTOR
        a = pd.actor
        a.obs_dim, a.h1, a.h2, a.act_dim = obs_dim, h1, h2, 1
        a.W1, a.b1, a.W2, a.b2, a.W3, a.b3 = (w[k].data_ptr() for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
        if "ln1_g" in w:                        # LayerNorm actor (models_editted.py:45-46, 50-51)
            a.ln1_g, a.ln1_b, a.ln2_g, a.ln2_b = (w[k].data_ptr() for k in ("ln1_g", "ln1_b", "ln2_g", "ln2_b"))
        a.last_layer_tanh = int(policy.last_layer_tanh)
        a.obs_clip = float(policy.obs_clip)
        a.precision = {"f32": _ffi.SSC_PREC_F32, "bf16_mfma": _ffi.SSC_PREC_BF16_MFMA}[policy.precision]
        pd.ou.mu, pd.ou.sigma, pd.ou.theta = policy.ou_mu, policy.ou_sigma, policy.ou_theta
        pd.ou.dt, pd.ou.epsilon = policy.ou_dt, policy.ou_epsilon
        if policy.d_ou_epsilon is not None:
            d = policy.d_ou_epsilon
            if d.dtype != torch.float32 or d.numel() != 1 or not d.is_cuda or d.device.index != self.s0.device.index:
                raise ValueError("d_ou_epsilon must be ONE fp32 value on the env's device")
            pd.ou.d_epsilon = d.data_ptr()
            w["_d_eps"] = d
        return pd, w  # keep the tensors alive for the duration of the launch

    def rollout(self, K, policy=None, out=None, ring=None, log=True, policy_desc=None):
        """K fused steps for every env (ssc_rollout).  Returns the TransitionChunk (or None if
        ``log=False``).  ``self.stats`` (f64[4]: sum reward, goal terminations, env-steps,
        finished episodes) is accumulated on the device."""
        if self._needs_reset:
            self.reset()
        if isinstance(policy, MpcPolicy):
            if policy.graph:
                return self._rollout_mpc_graph(K, policy.navigators, out, log, ring)
            return self._rollout_mpc(K, policy.navigators, out, log)
        if policy_desc is None:
            policy_desc = self.policy_desc(policy)
        pd, keep = policy_desc
        chunk = None
        if log:
            chunk = out if out is not None else TransitionChunk(self.obs_dim, K, self.n, self.device)
            if (chunk.K, chunk.N, chunk.obs_dim) != (K, self.n, self.obs_dim):
                raise ValueError("out chunk has the wrong shape")
            chunk.step0, chunk.env_id0 = self.t, self.env_id0
        st = _ffi.RolloutState(self.s0.data_ptr(), self.s1.data_ptr(), self.steps.data_ptr(),
                               self.ep_ret.data_ptr(), self.ou_x.data_ptr())
        log_s = chunk.as_struct() if chunk is not None else None
        ring_s = ring.as_struct() if ring is not None else None
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.ssc_rollout(
                ctypes.byref(self.params), ctypes.byref(pd), self.n, K, ctypes.byref(st),
                ctypes.byref(log_s) if log_s is not None else None,
                ctypes.byref(ring_s) if ring_s is not None else None,
                _ffi.ptr(self.stats), self._seed, self.env_id0, self.t, _stream()))
        self.t += K
        del keep
        return chunk


    def _rollout_mpc_graph(self, K, nav, out, log, ring):
        """K steps with the MPC navigator choosing every action, each step ONE replay of a captured HIP graph
        (NavigatorBatch.fused_step): nothing but the graph launch runs on the host, there is no device -> host
        read inside the chunk."""
        if nav.P != self.n:
            raise ValueError("MpcPolicy needs one navigation problem per env")
        if nav.problems.d != self.obs_dim:
            raise ValueError("the navigator plans in observation space")
        chunk = None
        # The captured launches bake in the navigator's geometry (N, H) and the address of every buffer they touch.
        # Graphs therefore LIVE ON THE NAVIGATOR (a NavigatorBatch owns most of those buffers and its N / H never
        # change): a graph dies with the object whose memory it references, and a new NavigatorBatch -- e.g. the same
        # problems with another num_control_samples -- starts with an empty cache instead of inheriting a graph through
        # a recycled id().  What can still change under a live navigator is in the key.
        cache = nav.__dict__.setdefault("_graphs", {})
        default_chunks = self.__dict__.setdefault("_mpc_chunks", {})
        if log:
            # the graph bakes the log pointers in: without `out` the steps are logged into a chunk kept with the
            # env and the caller gets a copy (a fresh chunk per call would mean a fresh capture per call)
            chunk = out if out is not None else default_chunks.setdefault(K, TransitionChunk(self.obs_dim, K, self.n, self.device))
            if (chunk.K, chunk.N, chunk.obs_dim) != (K, self.n, self.obs_dim):
                raise ValueError("out chunk has the wrong shape")
            chunk.step0, chunk.env_id0 = self.t, self.env_id0
        fb = nav._fused_buffers(self.device)
        fb["t"].fill_(self.t)
        fb["k"].zero_()
        fb["plan"].copy_(self.observe())
        model, pr = nav.model, nav.problems
        model.refresh_prepared_image()
        ptr = lambda t: None if t is None else t.data_ptr()

        def make_key():
            return (nav.P, nav.N, nav.H, self.n, self.kind, bytes(self.params), self._seed, self.env_id0,
                    tuple(ptr(x) for x in (self.s0, self.s1, self.steps, self.ep_ret, self.ou_x, self.stats)),
                    None if chunk is None else tuple(ptr(c) for c in (chunk.obs, chunk.act, chunk.rew, chunk.done, chunk.obs2)),
                    None if ring is None else (ptr(ring.cursor), ptr(ring.env_id), ptr(ring.length), ptr(ring.ret), ring.capacity),
                    model.precision, tuple(ptr(w) for w in model.W), tuple(ptr(b) for b in model.b),
                    bytes(model.norm),   # the fp32 path takes the statistics by value: baked into the capture
                    ptr(model._image), ptr(model._ws),
                    tuple(ptr(x) for x in (pr.wp, pr.left, pr.wp_off, pr.radii, pr.cur_idx)), pr.theta, pr.gamma, pr.hpf, pr.per_row,
                    tuple(ptr(x) for x in (nav._S, nav.actions_done, nav.at_goal, nav.start_idx)),
                    tuple(ptr(fb[k]) for k in sorted(fb)), nav.noise_amount, nav.seed, nav.problem_id0, nav.give_up,
                    nav.final_steps, tuple(nav.low.tolist()), tuple(nav.high.tolist()))
        key = make_key()
        if key not in cache:
            # warm-up outside the capture (lazy allocations, the weight image, LDS opt-ins) on a side stream, then
            # restore the state it advanced
            snap = [x.clone() for x in (self.s0, self.s1, self.steps, self.ep_ret, self.stats, nav.problems.cur_idx,
                                        nav.actions_done, nav.at_goal, fb["plan"], fb["t"], fb["k"])]
            ring_snap = None if ring is None else ring.cursor.clone()
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                nav.fused_step(self, chunk, ring)
            torch.cuda.current_stream(self.device).wait_stream(side)
            for dst, src in zip((self.s0, self.s1, self.steps, self.ep_ret, self.stats, nav.problems.cur_idx,
                                 nav.actions_done, nav.at_goal, fb["plan"], fb["t"], fb["k"]), snap):
                dst.copy_(src)
            if ring is not None:
                ring.cursor.copy_(ring_snap)
            key = make_key()          # the warm-up may have allocated the model's weight image / workspace
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                nav.fused_step(self, chunk, ring)
            for k in list(cache)[:max(0, len(cache) - 7)]:     # weights / chunks come and go: keep the newest graphs
                del cache[k]
            cache[key] = g
        g = cache[key]
        for _ in range(K):
            g.replay()
        self.t += K
        if chunk is not None and out is None:
            copy = TransitionChunk(self.obs_dim, K, self.n, self.device, packed=chunk.packed)
            for dst, src in zip(copy.columns(), chunk.columns()):
                dst.copy_(src)
            copy.step0, copy.env_id0 = chunk.step0, chunk.env_id0
            return copy
        return chunk

    def _rollout_mpc(self, K, nav, out, log):
        """K steps with the MPC navigator choosing every action (one problem per env); same transition log
        and statistics as the fused kernels, auto-reset on done.  Step-by-step reference path."""
        if nav.P != self.n:
            raise ValueError("MpcPolicy needs one navigation problem per env")
        chunk = None
        if log:
            chunk = out if out is not None else TransitionChunk(self.obs_dim, K, self.n, self.device)
            chunk.step0, chunk.env_id0 = self.t, self.env_id0
        for k in range(K):
            obs = self.observe().contiguous()                       # [N, obs_dim] (a copy: _obs is reused)
            action, _ = nav.get_action(obs, self.t)
            obs2, rew, done, _ = self.step(action)
            if chunk is not None:
                chunk.obs[:, k, :] = obs.t()
                chunk.act[k] = action[:, 0]
                chunk.rew[k] = rew
                chunk.done[k] = done.to(torch.uint8)
                chunk.obs2[:, k, :] = obs2.t()
            nav.observe(obs2)
            self.ep_ret += rew
            goals = (done & (obs2[:, 0] >= self.params.goal_position)) if self.kind == _ffi.SSC_ENV_MOUNTAINCAR \
                else torch.zeros_like(done)
            self.stats += torch.stack([rew.double().sum(), goals.double().sum(),
                                       torch.tensor(float(self.n), dtype=torch.float64, device=self.device),
                                       done.double().sum()])
            if bool(done.any()):
                self.ep_ret[done] = 0.0
                self.reset(mask=done)
                nav.restart(done)
        return chunk


class SingleEnvView:
    """An N=1 ``VecEnv`` behind the exact scalar ``gym.Env`` protocol of gym 0.10.5:
    ``reset() -> ndarray(obs_dim,)``, ``step(ndarray(act_dim,)) -> (ndarray, float, bool, {})``
    (what rlTrain.py:68,84 and every RLAgent consume)."""

    def __init__(self, vec_env):
        if vec_env.n != 1:
            raise ValueError("SingleEnvView needs a VecEnv with n_envs=1")
        self.vec = vec_env
        self.observation_space = vec_env.observation_space
        self.action_space = vec_env.action_space
        self.spec = vec_env.spec
        self._first = True

    def seed(self, seed=None):
        return self.vec.seed(seed)

    def reset(self):
        if self._first:
            obs = self.vec.reset()
            self._first = False
        else:
            obs = self.vec.reset(mask=torch.ones(1, dtype=torch.uint8))
        return obs[0].double().cpu().numpy()

    def step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        v = self.vec
        if v.auto_reset:
            obs, rew, done, info = v.step(torch.as_tensor(a, dtype=torch.float32))
            return obs[0].double().cpu().numpy(), float(rew[0].item()), bool(done[0].item()), info
        # one device -> host copy per step instead of three (each is a stream synchronisation)
        v._step_raw(torch.as_tensor(a, dtype=torch.float32))
        d = v.obs_dim
        packed = torch.cat((v._obs.reshape(-1), v._rew, v._done.to(torch.float32))).cpu().numpy()
        return packed[:d].astype(np.float64), float(packed[d]), bool(packed[d + 1] != 0.0), {}

    def render(self, mode="human"):
        return self.vec.render(mode)

    def close(self):
        self.vec.close()


class Continuous_MountainCarEnv_Editted(SingleEnvView):
    """Drop-in for smartstart/environments/continuous_mountain_car_editted.py:27 (N=1)."""

    version = 0

    def __init__(self, power_scalar, max_episode_steps=None, device="cuda", seed=1234):
        self.power_scalar = power_scalar
        name = "MountainCarContinuousActionX" + str(power_scalar) + "-v" + str(self.version)
        super().__init__(VecEnv(name, 1, device=device, power_scalar=power_scalar,
                                max_episode_steps=max_episode_steps or 0, seed=seed))
        if max_episode_steps is None:
            self.vec.params.max_episode_steps = 0
            self.spec = None  # the bare reference env has no spec until make_timed_env sets one (:157)

    def get_name(self):
        return "MountainCarContinuousActionX" + str(self.power_scalar) + "-v" + str(self.version)

    @classmethod
    def make_timed_env(cls, power_scalar, max_episode_steps=None, max_episode_seconds=None, device="cuda", seed=1234):
        """continuous_mountain_car_editted.py:154-159"""
        if max_episode_seconds is not None:
            raise NotImplementedError("max_episode_seconds is not supported")
        env = cls(power_scalar, max_episode_steps=max_episode_steps or 0, device=device, seed=seed)
        env.spec = EnvSpec(env.get_name(), max_episode_steps)
        return env


def make(env_id, device="cuda", seed=1234):
    """``gym.make`` for the ids the reference uses (DDPG_Baselines_agent.py:293-294 ...)."""
    return SingleEnvView(VecEnv(env_id, 1, device=device, seed=seed))
