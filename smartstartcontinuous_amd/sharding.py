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


class TransitionGather:
    """Packs the last ``g_steps`` steps of a TransitionChunk plus a snapshot of the chunk statistics
    into one contiguous byte buffer and gathers it to ``dst`` -- ONE collective per chunk; the learner
    sums the statistics out of the payload (``allreduce_stats=True`` adds an all-reduce so that every
    rank knows them).  On CUDA the pack is a single kernel (``ssc_pack_transitions``) on the producing
    stream and the collective runs on a side stream, overlapping the next chunk's rollout."""

    def __init__(self, obs_dim, g_steps, n, world, rank, device, dst=0, group=None, allreduce_stats=False):
        self.allreduce_stats = allreduce_stats
        self.obs_dim, self.g_steps, self.n = obs_dim, int(g_steps), int(n)
        self.world, self.rank, self.dst, self.group = world, rank, dst, group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        rec = self.g_steps * self.n * record_bytes(obs_dim)
        self.stats_off = (rec + 7) & ~7
        self.nbytes = self.stats_off + 32
        self.send = [torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.recv = None
        if rank == dst:
            self.recv = [torch.empty(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(world)]
        self._global_stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._stats_from_payload = False
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self.packed = [None, None]
        self.chunks_gathered = 0
        if self.cuda:
            import ctypes
            from . import _ffi
            self._ffi, self._ct = _ffi, ctypes
            assert _ffi.lib().ssc_pack_bytes(obs_dim, self.g_steps, self.n) == self.nbytes

    # layout of the packed buffer: [obs(obs_dim x g x n) f32 | act | rew | obs2 | done(u8) | pad | stats f64[4]]
    def _views(self, buf):
        g, n, d = self.g_steps, self.n, self.obs_dim
        f = buf[: 4 * g * n * (2 * d + 2)].view(torch.float32)
        o = 0
        obs = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        act = f[o:o + g * n].view(g, n); o += g * n
        rew = f[o:o + g * n].view(g, n); o += g * n
        obs2 = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        done = buf[4 * o:4 * o + g * n].view(g, n)
        return obs, act, rew, obs2, done

    def _stats_view(self, buf):
        return buf[self.stats_off:self.stats_off + 32].view(torch.float64)

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

    def unpack(self, src_rank):
        """Views (obs, act, rew, obs2, done) of the records received from ``src_rank`` (dst only)."""
        return self._views(self.recv[src_rank])

    def received_stats(self, src_rank):
        return self._stats_view(self.recv[src_rank])

    def wait_buffer_free(self, slot):
        """Kept for callers that overwrite a chunk from another stream; with ``submit`` the pack runs
        in order on the producing stream, so the chunk buffer is free as soon as ``submit`` returns."""
        return None

    def _collective(self, slot):
        dist.gather(self.send[slot], self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.allreduce_stats:
            # every rank learns the global statistics (the analogue of training_editted.py:173)
            self._global_stats.copy_(self._stats_view(self.send[slot]))
            dist.all_reduce(self._global_stats, op=dist.ReduceOp.SUM, group=self.group)
        else:
            # the (cumulative) statistics ride in the gathered payload: the learner sums the newest snapshots
            # when somebody asks (global_stats) -- no second collective and no per-chunk reduction kernels
            self._stats_from_payload = True
        self.chunks_gathered += 1

    @property
    def global_stats(self):
        """[sum of rewards, #done, #env-steps, #episodes] over all ranks as of the last gathered chunk
        (float64 [4]; on the learner rank, or on every rank with ``allreduce_stats``)."""
        if self._stats_from_payload and self.rank == self.dst:
            self.finish()
            self._global_stats.copy_(torch.stack([self._stats_view(r) for r in self.recv]).sum(dim=0))
            self._stats_from_payload = False
        return self._global_stats

    def submit(self, chunk, slot, stats):
        """Call right after the rollout that filled ``chunk`` was enqueued (same stream).

        The pack (one small kernel, ~10 us) runs IN ORDER on the producing stream, so the rollout
        stream never waits on another stream for its 1.7 GB chunk buffer; only the collective runs on
        the side stream, double-buffered through the two send slots."""
        if not self.cuda:
            self.pack(chunk, slot, stats)
            self._collective(slot)
            return
        main = torch.cuda.current_stream(self.device)
        if self.packed[slot] is not None:
            main.wait_event(self.packed[slot])        # send[slot] was consumed by its collective
        self.pack(chunk, slot, stats)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self._collective(slot)
            sent = torch.cuda.Event()
            sent.record(self.side)
            self.packed[slot] = sent

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)


def broadcast_flat(flat, src=0, group=None):
    """Everybody's ``flat`` becomes the ``src`` rank's (one collective for all of a network's parameters)."""
    dist.broadcast(flat, src=src, group=group)
    return flat


def rl_train_sharded_ddpg(env, agent, num_chunks, chunk_steps, rank, world, learner=0, gather_steps=None,
                          replay_capacity=1 << 20, train_iters=None, seed=0, group=None, ring_capacity=1 << 20):
    """The actor-learner loop of ``rl_train_vec_ddpg`` over ``world`` GPUs (one process each, ``env`` = this rank's
    shard of one global env-id space).  Per chunk:

      every rank     rolls its envs out under the CURRENT actor (fused kernel, OU noise), packs the last
                     ``gather_steps`` steps (default: 2^20 / N_total) and joins ONE gather to the learner;
      learner rank   appends every rank's records to its device replay ring, runs ``train_iters`` DDPG iterations
                     (``ssc_ddpg_train``), then
      every rank     joins ONE broadcast of the learner's flat actor parameters -- the rollout policy reads views
                     of that array, so the next chunk acts with the new weights.

    Returns (Summary of THIS rank's finished episodes, losses per chunk [learner only], replay [learner only])."""
    from .replay_buffer import DeviceReplayBuffer
    from .rl_train import Summary
    from .vec_env import EpisodeRing, TransitionChunk
    n_total = env.n * world
    g = int(gather_steps) if gather_steps is not None else max(1, min(chunk_steps, (1 << 20) // n_total))
    gather = TransitionGather(env.obs_dim, g, env.n, world, rank, env.device, dst=learner, group=group)
    summary = Summary("sharded_ddpg_" + env.spec.id)
    ring = EpisodeRing(ring_capacity, env.device)
    chunk = TransitionChunk(env.obs_dim, chunk_steps, env.n, env.device)
    replay = DeviceReplayBuffer(replay_capacity, env.obs_dim, 1, env.device, seed=seed) if rank == learner else None
    broadcast_flat(agent.actor_flat, src=learner, group=group)          # MpiAdam.sync: start from the root's parameters
    losses = []
    generations = 0.0
    for i in range(num_chunks):
        pd = env.policy_desc(agent.as_policy())
        env.rollout(chunk_steps, out=chunk, ring=ring, policy_desc=pd)
        gather.submit(chunk, i & 1, env.stats)
        gather.finish()
        if rank == learner:
            for src in range(world):
                replay.append_chunk(TransitionChunk.from_columns(*gather.unpack(src)),
                                    reward_scale=agent.reward_scale)
            l = agent.train_from(replay, train_iters)
            if l is not None:
                losses.append(l)
        broadcast_flat(agent.actor_flat, src=learner, group=group)
        (ids, lens, rets), _d = ring.drain()
        summary.extend_records(lens, rets)
        generations += len(lens) / float(env.n)      # epsilon decays once per episode per env (DDPG_Baselines_agent.py:255-258)
        while generations >= 1.0:
            agent.decaying_ou_action_noise.reduce_epsilon()
            generations -= 1.0
    return summary, losses, replay

