"""Host-side experience replay with episode bookkeeping -- same public interface and semantics as
smartstart/RLAgents/replay_buffer.py (FIFO of ``(s, a, r, t, s2)`` records, episode-start markers so
that the path to any stored state can be recovered, single-writer guard ``main_agent``), plus
``add_chunk`` to ingest the SoA transition log produced by the fused rollout kernel.

Storage is a numpy ring (not a deque of tuples): ``sample_batch`` is O(batch) instead of the
reference's O(|buffer|) list copy per step (replay_buffer.py:79-83).
"""
from __future__ import annotations

import random
from collections import deque

import numpy as np


class ReplayBuffer:
    def __init__(self, main_agent, max_buffer_size):
        self.main_agent = main_agent
        self.max_buffer_size = int(max_buffer_size)
        # episode numbering exactly as in the reference (replay_buffer.py:33-44)
        self.next_episode_number = 0
        self.episode_starting_indices = deque()
        self._s = self._a = self._r = self._t = self._s2 = None
        self._head = 0     # ring index of the oldest record
        self._len = 0

    # ------------------------------------------------------------------ storage helpers --
    def _alloc(self, s, a):
        s = np.asarray(s, dtype=np.float64).reshape(-1)
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        n = self.max_buffer_size
        self._s = np.zeros((n, s.size))
        self._s2 = np.zeros((n, s.size))
        self._a = np.zeros((n, a.size))
        self._r = np.zeros(n)
        self._t = np.zeros(n, dtype=bool)

    def _phys(self, i):
        return (self._head + np.asarray(i)) % self.max_buffer_size

    @property
    def buffer(self):
        """Sequence view with ``len()`` and indexing yielding ``(s, a, r, t, s2)`` tuples."""
        return _BufferView(self)

    def set_main_agent(self, new_main_agent):
        self.main_agent = new_main_agent

    # ---------------------------------------------------------------------------- add --
    def add(self, observing_agent, s, a, r, t, s2):
        """replay_buffer.py:49-74"""
        if observing_agent is not self.main_agent:
            return
        if self._s is None:
            self._alloc(s, a)
        if self._len < self.max_buffer_size:
            p = (self._head + self._len) % self.max_buffer_size
            self._len += 1
        else:
            if len(self.episode_starting_indices) > 0 and \
                    self.episode_starting_indices[0] == self.next_episode_number - self.max_buffer_size:
                self.episode_starting_indices.popleft()
                if len(self.episode_starting_indices) > 0:
                    first = self.episode_starting_indices[0]
                    for i in range(len(self.episode_starting_indices)):
                        self.episode_starting_indices[i] -= first
                    self.next_episode_number -= first
                else:
                    self.next_episode_number = 0
            p = self._head
            self._head = (self._head + 1) % self.max_buffer_size
        self._s[p] = np.asarray(s, dtype=np.float64).reshape(-1)
        self._a[p] = np.asarray(a, dtype=np.float64).reshape(-1)
        self._r[p] = float(r)
        self._t[p] = bool(t)
        self._s2[p] = np.asarray(s2, dtype=np.float64).reshape(-1)
        self.next_episode_number += 1

    def add_chunk(self, observing_agent, chunk, env_index=0, last_steps=None):
        """Ingest one env's column of a TransitionChunk (device SoA log) in step order, opening a new
        episode after every ``done`` -- what ``rlTrain`` + ``ReplayBuffer.add`` would have produced."""
        if observing_agent is not self.main_agent:
            return
        K = chunk.K if last_steps is None else min(int(last_steps), chunk.K)
        sl = slice(chunk.K - K, chunk.K)
        s = chunk.obs[:, sl, env_index].t().double().cpu().numpy()
        s2 = chunk.obs2[:, sl, env_index].t().double().cpu().numpy()
        a = chunk.act[sl, env_index].double().cpu().numpy()
        r = chunk.rew[sl, env_index].double().cpu().numpy()
        t = chunk.done[sl, env_index].bool().cpu().numpy()
        for k in range(K):
            self.add(observing_agent, s[k], [a[k]], r[k], t[k], s2[k])
            if t[k] and k + 1 < K:
                self.start_new_episode(observing_agent)

    def size(self):
        return self._len

    def __len__(self):
        return self._len

    # ------------------------------------------------------------------------ sampling --
    def _gather(self, idx):
        p = self._phys(idx)
        return self._s[p], self._a[p], self._r[p], self._t[p], self._s2[p]

    def sample_batch(self, batch_size):
        """replay_buffer.py:79-91 (uniform without replacement)."""
        n = min(batch_size, self._len)
        return self._gather(np.asarray(random.sample(range(self._len), n), dtype=np.int64))

    def all_batch(self):
        return self._gather(np.arange(self._len))

    def get_all_states(self):
        """replay_buffer.py:102-103: every s plus the last s2."""
        s = self._s[self._phys(np.arange(self._len))]
        return np.concatenate([s, self._s2[self._phys(self._len - 1)][None, :]], axis=0)

    def clear(self):
        self._head = self._len = 0
        self.next_episode_number = 0

    # ------------------------------------------------------------------------ episodes --
    def start_new_episode(self, observing_agent):
        """replay_buffer.py:109-115"""
        if observing_agent is not self.main_agent:
            return
        if len(self.episode_starting_indices) > 0 and self.episode_starting_indices[-1] == self.next_episode_number:
            return  # the reference prints a warning and ignores the duplicate
        self.episode_starting_indices.append(self.next_episode_number)

    def episode_number_to_buffer_index(self, episode_number):
        return self._len - (self.next_episode_number - episode_number)

    def buffer_index_to_episode_number(self, buffer_index):
        return buffer_index - self._len + self.next_episode_number

    def get_possible_smart_start_indices(self, n_ss):
        """replay_buffer.py:136-152"""
        if len(self.episode_starting_indices) == 0:
            return None
        first = self.episode_number_to_buffer_index(self.episode_starting_indices[0])
        number_of_states = min(n_ss, self._len - first)
        return np.array(random.sample(range(first, self._len), number_of_states))

    def get_episodic_path_to_buffer_index(self, buffer_index):
        """replay_buffer.py:154-176: states of the episode containing ``buffer_index`` up to and
        including its s2."""
        if len(self.episode_starting_indices) == 0:
            raise ValueError(": (   -   no episodes have been recorded")
        episode_index = self.buffer_index_to_episode_number(buffer_index)
        start = None
        starts = self.episode_starting_indices
        for i in range(len(starts) - 1):
            if starts[i] <= episode_index < starts[i + 1]:
                start = starts[i]
                break
        if start is None and starts[-1] <= episode_index:
            start = starts[-1]
        b0 = self.episode_number_to_buffer_index(start)
        idx = np.arange(b0, buffer_index + 1)
        s = self._s[self._phys(idx)]
        return [row for row in s] + [self._s2[self._phys(buffer_index)]]

    def step_to_s(self, step):
        return np.array(step[0])

    def step_to_s2(self, step):
        return np.array(step[4])


class _BufferView:
    def __init__(self, rb):
        self.rb = rb

    def __len__(self):
        return self.rb._len

    def __getitem__(self, i):
        if i < 0:
            i += self.rb._len
        if not 0 <= i < self.rb._len:
            raise IndexError(i)
        p = int(self.rb._phys(i))
        rb = self.rb
        return (rb._s[p], rb._a[p], rb._r[p], bool(rb._t[p]), rb._s2[p])


class DeviceReplayBuffer:
    """Device-resident replay ring for the vectorised loop: the storage of :class:`ReplayBuffer`
    (replay_buffer.py:49-74 of the reference -- FIFO of (s, a, r, t, s2) records, oldest overwritten) kept in
    HBM and fed straight from rollout chunks (``ssc_replay_append``); minibatch indices are drawn on the
    device too (``ssc_replay_sample``, uniform without replacement inside a batch like ``random.sample``,
    replay_buffer.py:79-83), so rollout -> replay -> ``ssc_ddpg_train`` never goes through the host.

    The SmartStart index queries (episode starts, smart-start paths) stay with the host
    :class:`ReplayBuffer`; this ring serves the learner's data path."""

    def __init__(self, capacity, obs_dim, act_dim=1, device="cuda", seed=0):
        import ctypes
        import torch
        from . import _ffi
        self._ctypes, self._torch, self._ffi = ctypes, torch, _ffi
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceReplayBuffer lives on the GPU; the host ring is ReplayBuffer")
        if act_dim != 1:
            raise ValueError("the transition log carries one action column")
        self.lib = _ffi.lib()
        self.capacity, self.obs_dim, self.act_dim = int(capacity), int(obs_dim), int(act_dim)
        f = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=self.device)
        self.s, self.s2 = f(self.capacity, obs_dim), f(self.capacity, obs_dim)
        self.a, self.r = f(self.capacity, act_dim), f(self.capacity)
        self.t = f(self.capacity, dt=torch.uint8)
        self.count = 0            # records appended so far (the reference's running next_episode_number-like count)
        self.seed, self._batches_drawn = int(seed), 0

    def __len__(self):
        return min(self.count, self.capacity)

    size = __len__

    def _stream(self):
        return self._ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def ring_struct(self):
        r = self._ffi.ReplayRing()
        r.s, r.a, r.r, r.t, r.s2 = (x.data_ptr() for x in (self.s, self.a, self.r, self.t, self.s2))
        r.capacity, r.obs_dim, r.act_dim = self.capacity, self.obs_dim, self.act_dim
        return r

    def append_chunk(self, chunk, reward_scale=1.0, last_steps=None):
        """``ReplayBuffer.add`` for every record of a :class:`TransitionChunk` (step-major, then env);
        ``last_steps`` keeps only the newest steps of the chunk."""
        K = chunk.K if last_steps is None else min(int(last_steps), chunk.K)
        log = chunk.as_struct()
        if K < chunk.K:                       # skip the first chunk.K - K steps: advance every column pointer
            skip = chunk.K - K
            for c in range(chunk.obs_dim):
                log.obs[c] = chunk.obs[c][skip:].data_ptr()
                log.obs2[c] = chunk.obs2[c][skip:].data_ptr()
            log.act, log.rew, log.done = chunk.act[skip:].data_ptr(), chunk.rew[skip:].data_ptr(), chunk.done[skip:].data_ptr()
        ring = self.ring_struct()
        with self._torch.cuda.device(self.device):
            self._ffi.check(self.lib.ssc_replay_append(self._ctypes.byref(ring), self._ctypes.byref(log), K, chunk.N,
                                                       self.count, float(reward_scale), self._stream()))
        self.count += K * chunk.N

    def sample_indices(self, n_batches, batch_size):
        """int32 [n_batches, batch_size] row indices, distinct inside a batch."""
        if len(self) < batch_size:
            raise ValueError("fewer records than one batch (the reference trains only once len >= batch_size)")
        idx = self._torch.empty((n_batches, batch_size), dtype=self._torch.int32, device=self.device)
        with self._torch.cuda.device(self.device):
            self._ffi.check(self.lib.ssc_replay_sample(self.seed, self._batches_drawn, len(self), n_batches, batch_size,
                                                       self._ffi.ptr(idx), self._stream()))
        self._batches_drawn += n_batches
        return idx

    def sample_batch(self, batch_size):
        """(s, a, r, t, s2) device tensors of one batch -- the ReplayBuffer.sample_batch tuple."""
        i = self.sample_indices(1, batch_size)[0].long()
        return self.s[i], self.a[i], self.r[i], self.t[i].bool(), self.s2[i]
