"""Host-side experience replay with episode bookkeeping -- same public interface and semantics as
smartstart/RLAgents/replay_buffer.py (FIFO of ``(s, a, r, t, s2)`` records, episode-start markers so
that the path to any stored state can be recovered, single-writer guard ``main_agent``), plus
``add_chunk`` to ingest the SoA transition log produced by the fused rollout kernel.

Storage is a numpy ring (not a deque of tuples): ``sample_batch`` is O(batch) instead of the
reference's O(|buffer|) list copy per step (replay_buffer.py:79-83).
"""
from __future__ import annotations

import bisect
import random
from collections import deque

import numpy as np


class ReplayBuffer:
    """Host replay ring.  Records live in preallocated numpy arrays (``_head`` / ``_len`` ring); episode starts are kept
    as ABSOLUTE record numbers (``_starts``: how many records had been added when the episode began) beside the count of
    records ever added (``_added``).  The reference's bookkeeping (replay_buffer.py:33-44,57-72: relative
    ``episode_starting_indices`` that are re-based, element by element, whenever the oldest start is evicted, and a
    ``next_episode_number`` that moves with them) is a VIEW of that state -- ``index = start - _base``,
    ``next_episode_number = _added - _base`` -- where ``_base`` is the absolute number the reference currently calls 0;
    an eviction is then O(1) (pop the start, move ``_base``).  The values read through the two properties are pinned by
    traces of the reference's own buffer (tests/golden/replay_buffer_kats.npz)."""

    def __init__(self, main_agent, max_buffer_size):
        self.main_agent = main_agent
        self.max_buffer_size = int(max_buffer_size)
        self._starts = deque()     # absolute record numbers at which the episodes still (partly) in the ring began
        self._added = 0            # records ever added
        self._base = 0             # absolute record number of the reference's episode number 0
        self._s = self._a = self._r = self._t = self._s2 = None
        self._head = 0     # ring index of the oldest record
        self._len = 0

    @property
    def episode_starting_indices(self):
        """The reference's list (replay_buffer.py:33-44), derived: a fresh list -- mutate through ``start_new_episode``."""
        return [a - self._base for a in self._starts]

    @property
    def next_episode_number(self):
        return self._added - self._base

    @next_episode_number.setter
    def next_episode_number(self, value):
        # the reference assigns the counter and leaves the VALUES in episode_starting_indices alone (clear() :105-107,
        # load() :131): re-express the absolute starts so that the view keeps reading the same numbers
        view = [a - self._base for a in self._starts]
        self._base = self._added - int(value)
        self._starts = deque(v + self._base for v in view)

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
            # the oldest record leaves.  If an episode began exactly there, that start is forgotten and numbering restarts
            # at the oldest start still known -- or from scratch when none is (replay_buffer.py:57-72)
            if self._starts and self._starts[0] == self._added - self.max_buffer_size:
                self._starts.popleft()
                self._base = self._starts[0] if self._starts else self._added
            p = self._head
            self._head = (self._head + 1) % self.max_buffer_size
        self._s[p] = np.asarray(s, dtype=np.float64).reshape(-1)
        self._a[p] = np.asarray(a, dtype=np.float64).reshape(-1)
        self._r[p] = float(r)
        self._t[p] = bool(t)
        self._s2[p] = np.asarray(s2, dtype=np.float64).reshape(-1)
        self._added += 1

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
        """replay_buffer.py:105-107: the records go, next_episode_number = 0, episode_starting_indices keeps its values."""
        self._head = self._len = 0
        self.next_episode_number = 0

    # ------------------------------------------------------------------------ episodes --
    def start_new_episode(self, observing_agent):
        """replay_buffer.py:109-115"""
        if observing_agent is not self.main_agent:
            return
        if self._starts and self._starts[-1] == self._added:
            return  # the reference prints a warning and ignores the duplicate
        self._starts.append(self._added)

    def episode_number_to_buffer_index(self, episode_number):
        return self._len - (self.next_episode_number - episode_number)

    def buffer_index_to_episode_number(self, buffer_index):
        return buffer_index - self._len + self.next_episode_number

    def get_possible_smart_start_indices(self, n_ss):
        """replay_buffer.py:136-152"""
        if not self._starts:
            return None
        first = self.episode_number_to_buffer_index(self._starts[0] - self._base)
        number_of_states = min(n_ss, self._len - first)
        return np.array(random.sample(range(first, self._len), number_of_states))

    def get_episodic_path_to_buffer_index(self, buffer_index):
        """replay_buffer.py:154-176: states of the episode containing ``buffer_index`` up to and
        including its s2."""
        if not self._starts:
            raise ValueError(": (   -   no episodes have been recorded")
        # the last start at or before the record (the starts are increasing: a bisection instead of the reference's scan)
        record = self._added - self._len + int(buffer_index)             # absolute number of the record
        starts = list(self._starts)
        k = bisect.bisect_right(starts, record) - 1
        if k < 0:
            raise TypeError("buffer index %d precedes every recorded episode start" % int(buffer_index))   # the reference fails on start = None here
        b0 = starts[k] - (self._added - self._len)
        idx = np.arange(b0, buffer_index + 1)
        s = self._s[self._phys(idx)]
        return [row for row in s] + [self._s2[self._phys(buffer_index)]]

    # ------------------------------------------------------------------------ persistence --
    def save(self, path='replay_buffer.obj'):
        """replay_buffer.py:117-124: pickles the record sequence (a list of ``(s, a, r, t, s2)`` tuples like the
        reference's deque contents) -- to ``replay_buffer.obj`` in the working directory by default, like the reference."""
        import pickle
        with open(path, 'wb') as f:
            pickle.dump([self.buffer[i] for i in range(self._len)], f)

    def load(self, path='replay_buffer.obj'):
        """replay_buffer.py:126-134: restores the records; like the reference the episode markers are NOT restored
        (``next_episode_number = len(buffer)``) and a missing file is reported, not raised."""
        import pickle
        try:
            with open(path, 'rb') as f:
                records = list(pickle.load(f))
        except (OSError, EOFError):
            print('there was no file to load')
            return False
        self._head = self._len = 0
        self._s = None
        self._starts.clear()
        self._added = self._base = 0
        agent = self.main_agent
        for (s, a, r, t, s2) in records[-self.max_buffer_size:]:
            self.add(agent, s, a, r, t, s2)
        self.next_episode_number = self._len
        return True

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

    ``track_episodes=True`` (needs ``n_envs``, the env count of the chunks it is fed) keeps the episode index on the
    device too -- per record the step count of its env's running episode (``ssc_replay_ring::ep_steps``) -- and with it
    the SmartStart queries of the reference buffer: :meth:`get_possible_smart_start_indices`
    (replay_buffer.py:136-152), :meth:`get_episodic_path_to_buffer_index` (:154-176) and :meth:`get_all_states`
    (:102-103), all returning DEVICE tensors.  Buffer indices count from the oldest record in the ring (0) to the newest
    (len - 1), records in append order (step-major, then env)."""

    def __init__(self, capacity, obs_dim, act_dim=1, device="cuda", seed=0, track_episodes=False, n_envs=None,
                 max_path_len=1001):
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
        self.track_episodes = bool(track_episodes)
        self.ep_steps = self.ep_run = None
        if self.track_episodes:
            if n_envs is None:
                raise ValueError("track_episodes needs n_envs (records of one env lie n_envs apart in the ring)")
            self.n_envs = int(n_envs)
            self.ep_steps = f(self.capacity, dt=torch.int32)
            self.ep_run = f(self.n_envs, dt=torch.int32)
            self.max_path_len = int(max_path_len)
            self._next_step0 = None       # global step index the next contiguous chunk starts at
            self._queries = 0
            self._path = f(self.max_path_len + 1, obs_dim)
            self._path_len = f(1, dt=torch.int32)

    def __len__(self):
        return min(self.count, self.capacity)

    size = __len__

    def _stream(self):
        return self._ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def ring_struct(self):
        r = self._ffi.ReplayRing()
        r.s, r.a, r.r, r.t, r.s2 = (x.data_ptr() for x in (self.s, self.a, self.r, self.t, self.s2))
        r.capacity, r.obs_dim, r.act_dim = self.capacity, self.obs_dim, self.act_dim
        if self.track_episodes:
            r.ep_steps, r.ep_run = self.ep_steps.data_ptr(), self.ep_run.data_ptr()
        return r

    def append_chunk(self, chunk, reward_scale=1.0, last_steps=None):
        """``ReplayBuffer.add`` for every record of a :class:`TransitionChunk` (step-major, then env);
        ``last_steps`` keeps only the newest steps of the chunk."""
        K = chunk.K if last_steps is None else min(int(last_steps), chunk.K)
        if self.track_episodes:
            if chunk.N != self.n_envs:
                raise ValueError(f"this ring indexes episodes of {self.n_envs} envs, the chunk has {chunk.N}")
            # an env's running step count carries over only when this chunk continues where the last append stopped
            first_step = chunk.step0 + chunk.K - K
            if self._next_step0 is not None and first_step != self._next_step0:
                self.ep_run.zero_()
            self._next_step0 = chunk.step0 + chunk.K
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

    def append_shards(self, shards, n_total, reward_scale=1.0):
        """One step-major append of ``n_total`` envs per step delivered as shards: ``shards`` = [(chunk, env_off), ...],
        every chunk holding the same K steps of envs [env_off, env_off + chunk.N) (``ssc_replay_append_shard``).  The ring
        ends up exactly as after ``append_chunk`` of the one chunk over all ``n_total`` envs -- the learner of a sharded
        run appends every rank's gathered records this way (record number = step * n_total + global env)."""
        if self.track_episodes and n_total != self.n_envs:
            raise ValueError(f"this ring indexes episodes of {self.n_envs} envs, the shards cover {n_total}")
        K = shards[0][0].K
        if sum(c.N for c, _ in shards) != n_total or any(c.K != K for c, _ in shards):
            raise ValueError("the shards must cover the n_total envs of the same K steps exactly once")
        ring = self.ring_struct()
        with self._torch.cuda.device(self.device):
            for chunk, env_off in shards:
                log = chunk.as_struct()
                self._ffi.check(self.lib.ssc_replay_append_shard(self._ctypes.byref(ring), self._ctypes.byref(log), K, chunk.N,
                                                                 self.count, int(n_total), int(env_off), float(reward_scale),
                                                                 self._stream()))
        self.count += K * int(n_total)

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

    # ------------------------------------------------------------ SmartStart queries (track_episodes) --
    def _need_index(self):
        if not self.track_episodes:
            raise RuntimeError("this DeviceReplayBuffer was built without track_episodes=True")

    def get_all_states(self):
        """replay_buffer.py:102-103: every s (oldest first) plus the newest record's s2 -- a device tensor."""
        torch = self._torch
        size = len(self)
        if size == 0:
            return torch.zeros((0, self.obs_dim), device=self.device)
        if self.count <= self.capacity:
            s = self.s[:size]
        else:
            h = self.count % self.capacity
            s = torch.cat([self.s[h:], self.s[:h]])
        return torch.cat([s, self.s2[(self.count - 1) % self.capacity][None, :]])

    def physical(self, buffer_index):
        """ring rows of buffer indices (tensor or int; 0 = oldest record)."""
        return (buffer_index + (self.count - len(self))) % self.capacity

    def get_possible_smart_start_indices(self, n_ss):
        """replay_buffer.py:136-152 on the device: up to ``n_ss`` distinct buffer indices (int64 device tensor) drawn
        uniformly from the records whose episode start is still in the ring; None when there is none."""
        self._need_index()
        torch, ct, ffi = self._torch, self._ctypes, self._ffi
        n_ss = int(n_ss)
        if len(self) == 0 or n_ss <= 0:
            return None
        if n_ss > 4096:
            raise ValueError("n_ss <= 4096 on the device path")
        idx = torch.empty(n_ss, dtype=torch.int32, device=self.device)
        n_out = torch.zeros(1, dtype=torch.int32, device=self.device)
        nb = self.lib.ssc_replay_smart_start_workspace_bytes(n_ss)
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        ring = self.ring_struct()
        with torch.cuda.device(self.device):
            ffi.check(self.lib.ssc_replay_smart_start_indices(ct.byref(ring), self.count, self.n_envs, n_ss, self.seed ^ 0x5353,
                                                              self._queries, ffi.ptr(idx), ffi.ptr(n_out), ffi.ptr(ws), nb,
                                                              self._stream()))
        self._queries += 1
        got = idx[idx >= 0].long()
        return got if got.numel() > 0 else None

    def get_episodic_path_to_buffer_index(self, buffer_index):
        """replay_buffer.py:154-176 on the device: [L + 1, obs_dim] device tensor -- the states of the episode containing
        ``buffer_index`` (an int or a 1-element device tensor, e.g. one entry of get_possible_smart_start_indices) up to
        that record, then its s2.  Raises if the episode start has left the ring."""
        self._need_index()
        torch, ct, ffi = self._torch, self._ctypes, self._ffi
        bi = torch.as_tensor(buffer_index, device=self.device).reshape(1).to(torch.int32)
        ring = self.ring_struct()
        with torch.cuda.device(self.device):
            ffi.check(self.lib.ssc_replay_episode_path(ct.byref(ring), self.count, self.n_envs, ffi.ptr(bi), self.max_path_len,
                                                       ffi.ptr(self._path), ffi.ptr(self._path_len), self._stream()))
        rows = int(self._path_len.item())          # the ONE host read of a smart-start query: the path length
        if rows == 0:
            raise ValueError(": (   -   the episode start of this record is no longer in the ring")
        return self._path[:rows].clone()
