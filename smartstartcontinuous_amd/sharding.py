"""Multi-GPU sharding of the rollout: one process per GPU, envs partitioned by contiguous
ranges of one global env-id space, and the per-chunk exchange step.

Reference counterparts (SURVEY.md section 2 "Collective / IPC call sites"):
  * rollout gather -- ``CollectSamples.collect_samples`` funnels every worker's rollout back to
    the parent through ``apply_async`` callbacks
    (NN_Dynamics_Model/collect_samples_threaded.py:31-50,110-111); here: an RCCL ``gather`` of
    the packed transition records to the learner rank over xGMI.
  * stats -- the epoch-stats ``allreduce`` of training_editted.py:173; here an
    ``all_reduce(SUM)`` of the 4 chunk statistics.

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
    """Packs the last ``g_steps`` steps of a TransitionChunk into one contiguous byte buffer and
    gathers it to ``dst`` (one collective per chunk), then all-reduces the chunk statistics.
    On CUDA the work runs on a side stream so that it overlaps the next chunk's rollout;
    ``wait_buffer_free(slot)`` orders the next overwrite of a chunk buffer after its pack."""

    def __init__(self, obs_dim, g_steps, n, world, rank, device, dst=0, group=None):
        self.obs_dim, self.g_steps, self.n = obs_dim, int(g_steps), int(n)
        self.world, self.rank, self.dst, self.group = world, rank, dst, group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.nbytes = self.g_steps * self.n * record_bytes(obs_dim)
        self.send = [torch.empty(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.recv = None
        if rank == dst:
            self.recv = [torch.empty(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(world)]
        self.global_stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self.packed = [None, None]
        self.chunks_gathered = 0

    # layout of the packed buffer: [obs(obs_dim x g x n) f32 | act | rew | obs2 | done(u8)]
    def _views(self, buf):
        g, n, d = self.g_steps, self.n, self.obs_dim
        f = buf[: 4 * g * n * (2 * d + 2)].view(torch.float32)
        o = 0
        obs = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        act = f[o:o + g * n].view(g, n); o += g * n
        rew = f[o:o + g * n].view(g, n); o += g * n
        obs2 = f[o:o + d * g * n].view(d, g, n); o += d * g * n
        done = buf[4 * o:].view(g, n)
        return obs, act, rew, obs2, done

    def pack(self, chunk, slot):
        g = self.g_steps
        obs, act, rew, obs2, done = self._views(self.send[slot])
        obs.copy_(chunk.obs[:, chunk.K - g:, :])
        act.copy_(chunk.act[chunk.K - g:])
        rew.copy_(chunk.rew[chunk.K - g:])
        obs2.copy_(chunk.obs2[:, chunk.K - g:, :])
        done.copy_(chunk.done[chunk.K - g:])

    def unpack(self, src_rank):
        """Views (obs, act, rew, obs2, done) of the records received from ``src_rank`` (dst only)."""
        return self._views(self.recv[src_rank])

    def wait_buffer_free(self, slot):
        ev = self.packed[slot]
        if ev is not None and self.cuda:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _exchange(self, chunk, slot, stats):
        self.pack(chunk, slot)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.packed[slot] = ev
        dist.gather(self.send[slot], self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        self.global_stats.copy_(stats)
        dist.all_reduce(self.global_stats, op=dist.ReduceOp.SUM, group=self.group)
        self.chunks_gathered += 1

    def submit(self, chunk, slot, stats):
        """Call right after the rollout that filled ``chunk`` was enqueued."""
        if not self.cuda:
            self._exchange(chunk, slot, stats)
            return
        produced = torch.cuda.Event()
        produced.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(produced)
            self._exchange(chunk, slot, stats)

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
