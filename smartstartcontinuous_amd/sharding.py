"""Multi-GPU sharding of the rollout: one process per GPU, envs partitioned by contiguous
ranges of one global env-id space, and the per-chunk exchange step.

Reference counterparts (SURVEY.md section 2 "Collective / IPC call sites"):
  * rollout gather -- ``CollectSamples.collect_samples`` funnels every worker's rollout back to
    the parent through ``apply_async`` callbacks
    (NN_Dynamics_Model/collect_samples_threaded.py:31-50,110-111); here: an RCCL ``gather`` of
    the packed transition records to the learner rank over xGMI.
  * stats -- the epoch-stats ``allreduce`` of training_editted.py:173; here an
    ``all_reduce(SUM)`` of the 4 chunk statistics.
  * parameter sync -- ``MpiAdam.sync`` / ``target_init_updates`` broadcasting the root's parameters
    (ddpg_editted.py:331-336); here ONE ``broadcast`` of the learner's flat actor-parameter array per chunk
    (``rl_train_sharded_ddpg``), the actors' weight views alias it so no copy follows.

Envs are independent, so the rollout itself needs no collective; all RNG is keyed by the
GLOBAL env id, so any world size reproduces the same per-env streams.

Bounded vs full gather: a rank produces 1.68 GB per 1024-step chunk
in ~0.5 ms, two orders of magnitude more than its xGMI links can move in that time, and more
than any replay buffer ingests (buffer_size 1e5 in every example).  The default therefore
ships the LAST ``g_steps`` steps of every chunk (``g_steps * N_total <= 2^20`` records); the
full stream is available (``g_steps = K``) and is link-bound.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.distributed as dist


def shard_range(n_total, world, rank):
    """Contiguous env-id range [lo, hi) owned by ``rank`` (balanced to within one env)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_bytes(obs_dim):
    """Bytes of one (s, a, r, t, s2) record: 25 for MountainCar, 33 for Pendulum."""
    return 8 * obs_dim + 9


def _backend(group=None):
    return dist.get_backend(group) if dist.is_available() and dist.is_initialized() else None


class TransitionGather:
    """Packs the last ``g_steps`` steps of a TransitionChunk plus a snapshot of the chunk statistics
    into one contiguous byte buffer and gathers it to ``dst`` -- ONE collective per chunk; the learner
    sums the statistics out of the payload (``allreduce_stats=True`` adds an all-reduce so that every
    rank knows them).  On CUDA the pack is a single kernel (``ssc_pack_transitions``) on the producing
    stream and the collective runs on a side stream, overlapping the next chunk's rollout.

    Shards may differ in size (``shard_range`` balances to within one env): every rank's ``n`` is exchanged
    once at construction, every send / receive slot is sized for the largest shard (a gather needs equal
    message sizes), and ``unpack(src)`` cuts the views with the SOURCE rank's ``n``.

    Both the send and the receive side are double-buffered by ``slot`` (= chunk index & 1): the collective of
    chunk i fills ``recv[i & 1]`` while the learner may still be reading chunk i-1 out of ``recv[(i-1) & 1]``.

    ``host_staging`` (automatic for a gloo group with CUDA buffers): the packed payload travels through pinned
    host memory and the collective runs on host tensors -- the rehearsal path for the N > 1 control flow on a
    box without RCCL peers; it is synchronous."""

    def __init__(self, obs_dim, g_steps, n, world, rank, device, dst=0, group=None, allreduce_stats=False,
                 host_staging=None, n_all=None):
        self.allreduce_stats = allreduce_stats
        self.obs_dim, self.g_steps, self.n = obs_dim, int(g_steps), int(n)
        self.world, self.rank, self.dst, self.group = world, rank, dst, group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if host_staging is None:
            host_staging = self.cuda and world > 1 and _backend(group) == "gloo"
        self.host_staging = bool(host_staging)
        # every rank's shard size, exchanged once (unequal shards must not reach dist.gather unsized)
        if n_all is not None:
            self.n_all = [int(x) for x in n_all]
        elif world > 1:
            mine = torch.tensor([self.n], dtype=torch.int64, device="cpu" if (not self.cuda or self.host_staging) else self.device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            self.n_all = [int(t.item()) for t in every]
        else:
            self.n_all = [self.n]
        if len(self.n_all) != world or self.n_all[rank] != self.n:
            raise ValueError(f"TransitionGather: shard sizes {self.n_all} do not match world {world} / local n {self.n}")
        self.n_total = sum(self.n_all)
        self._nbytes_of = [self._layout(m)[1] for m in self.n_all]
        self.stats_off, self.nbytes = self._layout(self.n)
        self.slot_bytes = max(self._nbytes_of)
        self.send = [torch.zeros(self.slot_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.recv = None
        if rank == dst:
            self.recv = [[torch.empty(self.slot_bytes, dtype=torch.uint8, device=self.device) for _ in range(world)]
                         for _ in range(2)]
        if self.host_staging:
            self._h_send = torch.zeros(self.slot_bytes, dtype=torch.uint8).pin_memory()
            self._h_recv = [torch.empty(self.slot_bytes, dtype=torch.uint8).pin_memory() for _ in range(world)] if rank == dst else None
        self._global_stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._stats_from_payload = False
        self.side = torch.cuda.Stream(self.device) if (self.cuda and not self.host_staging) else None
        self.packed = [None, None]        # event: send[slot] was consumed by its collective
        self.received = [None, None]      # event: recv[slot] holds the chunk (dst only)
        self.last_slot = None
        self.chunks_gathered = 0
        if self.cuda:
            import ctypes
            from . import _ffi
            self._ffi, self._ct = _ffi, ctypes
            assert _ffi.lib().ssc_pack_bytes(obs_dim, self.g_steps, self.n) == self.nbytes

    def _layout(self, n):
        rec = self.g_steps * n * record_bytes(self.obs_dim)
        off = (rec + 7) & ~7
        return off, off + 32

    # layout of the packed buffer: [obs(obs_dim x g x n) f32 | act | rew | obs2 | done(u8) | pad | stats f64[4]]
    def _views(self, buf, n=None):
        g, n, d = self.g_steps, (self.n if n is None else n), self.obs_dim
        f = buf[: 4 * g * n * (2 * d + 2)].view(torch.float32)
        o = 0
        obs = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        act = f[o:o + g * n].view(g, n); o += g * n
        rew = f[o:o + g * n].view(g, n); o += g * n
        obs2 = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        done = buf[4 * o:4 * o + g * n].view(g, n)
        return obs, act, rew, obs2, done

    def _stats_view(self, buf, n=None):
        off = self.stats_off if n is None else self._layout(n)[0]
        return buf[off:off + 32].view(torch.float64)

    def pack(self, chunk, slot, stats):
        g = self.g_steps
        if self.cuda:
            log = chunk.as_struct()
            ffi, ct = self._ffi, self._ct
            ffi.check(ffi.lib().ssc_pack_transitions(ct.byref(log), self.obs_dim, chunk.K, g, self.n, ffi.ptr(stats),
                                                     ffi.ptr(self.send[slot]),
                                                     ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            return
        obs, act, rew, obs2, done = self._views(self.send[slot])
        obs.copy_(chunk.obs[:, chunk.K - g:, :])
        act.copy_(chunk.act[chunk.K - g:])
        rew.copy_(chunk.rew[chunk.K - g:])
        obs2.copy_(chunk.obs2[:, chunk.K - g:, :])
        done.copy_(chunk.done[chunk.K - g:])
        self._stats_view(self.send[slot]).copy_(stats)

    def unpack(self, src_rank, slot=None):
        """Views (obs, act, rew, obs2, done) of the records received from ``src_rank`` in the newest gathered chunk
        (or in receive slot ``slot``); dst only.  Shapes follow the SOURCE rank's shard size."""
        slot = self.last_slot if slot is None else slot
        return self._views(self.recv[slot][src_rank], self.n_all[src_rank])

    def received_stats(self, src_rank, slot=None):
        slot = self.last_slot if slot is None else slot
        return self._stats_view(self.recv[slot][src_rank], self.n_all[src_rank])

    def wait_received(self, slot):
        """Make the current stream wait until receive slot ``slot`` holds its chunk (dst only; no host sync)."""
        if self.cuda and self.received[slot] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.received[slot])

    def _collective(self, slot):
        if self.host_staging:
            self._h_send.copy_(self.send[slot])                       # D2H (synchronous: pinned destination, same stream)
            torch.cuda.current_stream(self.device).synchronize()
            dist.gather(self._h_send, self._h_recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.recv[slot][r].copy_(self._h_recv[r], non_blocking=True)
            if self.allreduce_stats:
                h = self._stats_view(self._h_send).clone()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                self._global_stats.copy_(h)
            else:
                self._stats_from_payload = True
        else:
            dist.gather(self.send[slot], self.recv[slot] if self.rank == self.dst else None, dst=self.dst, group=self.group)
            if self.allreduce_stats:
                # every rank learns the global statistics (the analogue of training_editted.py:173)
                self._global_stats.copy_(self._stats_view(self.send[slot]))
                dist.all_reduce(self._global_stats, op=dist.ReduceOp.SUM, group=self.group)
            else:
                # the (cumulative) statistics ride in the gathered payload: the learner sums the newest snapshots
                # when somebody asks (global_stats) -- no second collective and no per-chunk reduction kernels
                self._stats_from_payload = True
        self.last_slot = slot
        self.chunks_gathered += 1

    @property
    def global_stats(self):
        """[sum of rewards, #done, #env-steps, #episodes] over all ranks as of the last gathered chunk
        (float64 [4]; on the learner rank, or on every rank with ``allreduce_stats``)."""
        if self._stats_from_payload and self.rank == self.dst:
            self.finish()
            self._global_stats.copy_(torch.stack([self.received_stats(r) for r in range(self.world)]).sum(dim=0))
            self._stats_from_payload = False
        return self._global_stats

    def submit(self, chunk, slot, stats, after=None):
        """Call right after the rollout that filled ``chunk`` was enqueued (same stream).  ``after``: an event the
        COLLECTIVE (not the rollout stream) additionally waits for -- e.g. "the learner stream has finished reading
        recv[slot]" when the consumer runs on a stream of its own.

        The pack (one small kernel: 12 us for 26 MB, 1.5 us for 3 MB) runs IN ORDER on the producing stream, so the rollout
        stream never waits on another stream for its 1.7 GB chunk buffer; only the collective runs on
        the side stream, double-buffered through the two send slots (and the two receive slots on ``dst``:
        the caller must be done reading ``recv[slot]`` -- i.e. the chunk gathered two submits ago -- on the
        CURRENT stream before it calls ``submit`` with that slot again; the side stream waits for the current
        stream's work up to here)."""
        if not self.cuda or self.host_staging:
            self.pack(chunk, slot, stats)
            self._collective(slot)
            return
        main = torch.cuda.current_stream(self.device)
        if self.packed[slot] is not None:
            main.wait_event(self.packed[slot])        # send[slot] was consumed by its collective
        self.pack(chunk, slot, stats)
        ready = torch.cuda.Event()
        ready.record(main)                            # also orders the learner's reads of recv[slot] before the refill
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if after is not None:
                self.side.wait_event(after)
            self._collective(slot)
            sent = torch.cuda.Event()
            sent.record(self.side)
            self.packed[slot] = sent
            self.received[slot] = sent

    def finish(self):
        if self.cuda and self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)


def broadcast_flat(flat, src=0, group=None):
    """Everybody's ``flat`` becomes the ``src`` rank's (one collective for all of a network's parameters)."""
    dist.broadcast(flat, src=src, group=group)
    return flat


def _views_like(flat, like):
    """dict of views into ``flat`` with the shapes (and order) of the weight dict ``like`` (agents.flatten_params)."""
    views, o = {}, 0
    for k, v in like.items():
        n = v.numel()
        views[k] = flat[o:o + n].view(v.shape)
        o += n
    return views


def rl_train_sharded_ddpg(env, agent, num_chunks, chunk_steps, rank, world, learner=0, gather_steps=None,
                          replay_capacity=1 << 20, train_iters=None, seed=0, group=None, ring_capacity=1 << 20,
                          pipelined=False, drain_every=None, on_chunk=None):
    """The actor-learner loop of ``rl_train_vec_ddpg`` over ``world`` GPUs (one process each, ``env`` = this rank's
    shard of one global env-id space).  Per chunk:

      every rank     rolls its envs out under the actor (fused kernel, OU noise), packs the last
                     ``gather_steps`` steps (default: 2^20 / N_total) and joins ONE gather to the learner;
      learner rank   appends every rank's records to its device replay ring -- record number = step * N_total + global
                     env id (``append_shards``), i.e. the ring of the single-GPU run whatever the world size --, runs
                     ``train_iters`` DDPG iterations (``ssc_ddpg_train``), decays epsilon once per generation of
                     N_total finished episodes out of the gathered statistics (:class:`rl_train.DecaySchedule`), then
      every rank     joins ONE broadcast of the learner's ``agent.actor_sync`` = [flat actor parameters | epsilon]
                     (``MpiAdam.sync``, ddpg_editted.py:331-336); the rollout kernel reads both where they landed.

    Envs are keyed by their GLOBAL id and the schedule by GLOBAL counts, so the transitions every env produces, the
    learner's replay ring and every parameter generation are the same for any world size (tests: 2 ranks == 1 rank, bit
    for bit).  Nothing at a chunk boundary reads the device: the finished-episode records go to the host every
    ``drain_every`` chunks.

    ``pipelined=False`` (the reference's own ordering: act, store, train, act ...): the gather is waited for, the
    learner trains, and the broadcast lands in ``agent.actor_sync`` -- which the rollout policy reads through views --
    before the next chunk starts.  Every actor idles while the learner trains.

    ``pipelined=True``: nothing on a rollout stream waits for the learner -- on the learner rank either: the append
    and the DDPG iterations run on a LEARNER STREAM of their own (ordered against the receive slots and the parameter
    buffers by events), gather AND broadcast run on the side stream; the synchronised array is double-buffered
    (generation g = parameters trained on the chunks <= g + the epsilon after chunk g, lands in buffer g & 1) and chunk j
    is rolled with generation j - 2, so the learner trains on chunk j-1 while every rank (its own included) already rolls
    chunk j: ONE CHUNK STALE compared with the synchronous loop (which rolls chunk j with generation j - 1).  The
    per-chunk critical path is max(T_rollout, T_gather + T_train + T_broadcast) instead of their sum.  The receive side
    is double-buffered too (``TransitionGather``), and the gather that refills a receive slot waits for the learner
    stream to have consumed it.

    ``on_chunk(i, chunk, env)`` -- if given -- is called right after rollout i was enqueued (same stream), e.g. to copy
    the chunk's transition log.

    Returns (Summary of THIS rank's finished episodes, losses per chunk [learner only], replay [learner only])."""
    from .replay_buffer import DeviceReplayBuffer
    from .rl_train import Summary, default_drain_every, epsilon_schedule
    from .vec_env import EpisodeRing, TransitionChunk
    gather = TransitionGather(env.obs_dim, 1, env.n, world, rank, env.device, dst=learner, group=group) \
        if gather_steps is None else None
    if gather is not None:      # the default G needs the TRUE global N (shards may differ by one env)
        g = max(1, min(chunk_steps, (1 << 20) // gather.n_total))
        n_all = gather.n_all
    else:
        g, n_all = int(gather_steps), None
    gather = TransitionGather(env.obs_dim, g, env.n, world, rank, env.device, dst=learner, group=group, n_all=n_all)
    n_total = gather.n_total
    env_off = [sum(gather.n_all[:r]) for r in range(world)]
    summary = Summary("sharded_ddpg_" + env.spec.id)
    ring = EpisodeRing(ring_capacity, env.device)
    if drain_every is None:
        drain_every = default_drain_every(ring_capacity, env.n, chunk_steps)
    chunks = [TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device) for _ in range(2 if pipelined else 1)]
    replay = DeviceReplayBuffer(replay_capacity, env.obs_dim, 1, env.device, seed=seed) if rank == learner else None
    schedule = epsilon_schedule(agent, n_total, env.device) if rank == learner else None
    broadcast_flat(agent.actor_sync, src=learner, group=group)          # MpiAdam.sync: start from the root's parameters (+ epsilon)
    stats0 = env.stats.clone()                                          # the statistics that travel count from here
    delta = torch.zeros_like(env.stats)
    finished = torch.zeros(1, dtype=torch.float64, device=env.device)   # learner: episodes finished on all ranks
    losses = []
    dropped = 0
    cuda = env.device.type == "cuda" and gather.side is not None
    if pipelined:
        wbuf = [agent.actor_sync.clone(), agent.actor_sync.clone()]     # generation g lives in wbuf[g & 1]
        base = agent.as_policy()
        pds = [env.policy_desc(dataclasses.replace(base, weights=_views_like(b[:-1], agent.weights), d_ou_epsilon=b[-1:]))
               for b in wbuf]                                           # the SAME policy the synchronous loop builds
        bcast_done = [None, None]
        learned = [None, None]                                          # learner stream is done with recv[slot] / wrote wbuf[slot]
        lstream = torch.cuda.Stream(env.device) if (cuda and rank == learner) else None
        if lstream is not None:
            lstream.wait_stream(torch.cuda.current_stream(env.device))  # replay / parameter setup happened on the main stream
    else:
        pd = env.policy_desc(agent.as_policy(device_epsilon=True))

    def learn(slot):
        gather.wait_received(slot)
        replay.append_shards([(TransitionChunk.from_columns(*gather.unpack(src, slot)), env_off[src]) for src in range(world)],
                             n_total, reward_scale=agent.reward_scale)
        l = agent.train_from(replay, train_iters)
        if l is not None:
            losses.append(l)
        torch.sum(torch.stack([gather.received_stats(src, slot)[3] for src in range(world)]), dim=0, keepdim=True, out=finished)
        schedule.update(finished)                                       # -> agent.d_epsilon, the tail of agent.actor_sync

    def drain():
        nonlocal dropped
        (ids, lens, rets), d = ring.drain()
        dropped += d
        summary.extend_records(lens, rets)

    for i in range(num_chunks):
        slot = i & 1
        if not pipelined:
            env.rollout(chunk_steps, out=chunks[0], ring=ring, policy_desc=pd)
            if on_chunk is not None:
                on_chunk(i, chunks[0], env)
            torch.sub(env.stats, stats0, out=delta)
            gather.submit(chunks[0], slot, delta)
            gather.finish()
            if rank == learner:
                learn(slot)
            broadcast_flat(agent.actor_sync, src=learner, group=group)
        else:
            main = torch.cuda.current_stream(env.device) if cuda else None
            if cuda and bcast_done[slot] is not None:
                main.wait_event(bcast_done[slot])                       # generation i-2 has landed in wbuf[slot]
            env.rollout(chunk_steps, out=chunks[slot], ring=ring, policy_desc=pds[slot])   # reads generation i-2
            if on_chunk is not None:
                on_chunk(i, chunks[slot], env)
            torch.sub(env.stats, stats0, out=delta)
            # side stream: gather(i) after this rollout -- and after the learner stream has read chunk i-2 out of recv[slot]
            gather.submit(chunks[slot], slot, delta, after=learned[slot])
            if cuda:
                trained = torch.cuda.Event()
                if rank == learner:
                    # learner stream: waits for gather(i) (hence for rollout(i), which read generation i-2 out of
                    # wbuf[slot]), trains generation i and parks it in wbuf[slot]; the main stream goes on to rollout(i+1)
                    with torch.cuda.stream(lstream):
                        learn(slot)
                        wbuf[slot].copy_(agent.actor_sync)
                        trained.record(lstream)
                    learned[slot] = trained
                else:
                    trained.record(main)                                # broadcast(i) overwrites wbuf[slot]: after rollout(i)
                with torch.cuda.stream(gather.side):                    # same order on every rank: gather(i), broadcast(i)
                    gather.side.wait_event(trained)
                    broadcast_flat(wbuf[slot], src=learner, group=group)
                    ev = torch.cuda.Event()
                    ev.record(gather.side)
                    bcast_done[slot] = ev
            else:
                if rank == learner:
                    learn(slot)
                    wbuf[slot].copy_(agent.actor_sync)
                broadcast_flat(wbuf[slot], src=learner, group=group)
        if (i + 1) % drain_every == 0:
            drain()
    if pipelined:
        gather.finish()
        if lstream is not None:
            torch.cuda.current_stream(env.device).wait_stream(lstream)
        if rank != learner:
            # the actors end with the newest generation they received
            agent.actor_sync.copy_(wbuf[(num_chunks - 1) & 1])
    drain()
    summary.dropped_episode_records = dropped
    # the host-side noise object follows the device schedule (exact on the learner, the broadcast fp32 value elsewhere)
    agent.decaying_ou_action_noise.epsilon = schedule.read()[1][0] if rank == learner else float(agent.d_epsilon.item())
    return summary, losses, replay
