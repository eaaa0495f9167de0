"""The N>1 exchange path on CPU: world_size-2 gloo processes gather packed transition records
to the learner rank and all-reduce the chunk statistics; plus the shard arithmetic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smartstartcontinuous_amd.sharding import TransitionGather, broadcast_flat, record_bytes, shard_range
from smartstartcontinuous_amd.vec_env import TransitionChunk


def test_shard_ranges_cover_the_id_space():
    for n, w in [(65536 * 8, 8), (10, 3), (7, 8), (1, 1), (524288, 6)]:
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert record_bytes(2) == 25 and record_bytes(3) == 33


class FakeChunk:
    """A TransitionChunk-shaped object on the CPU with deterministic content per (rank, k, i)."""

    def __init__(self, obs_dim, K, N, rank):
        self.K, self.N, self.obs_dim = K, N, obs_dim
        base = torch.arange(K * N, dtype=torch.float32).reshape(K, N) + 1000.0 * rank
        self.obs = torch.stack([base + 0.25 * c for c in range(obs_dim)])
        self.obs2 = self.obs + 0.5
        self.act = -base
        self.rew = base * 2
        self.done = ((torch.arange(K * N).reshape(K, N) + rank) % 7 == 0).to(torch.uint8)


def _worker(rank, world, port, obs_dim, g, tmp, allreduce=True, unequal=False, n_total=75):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, N = 12, 37
    if unequal:                         # shard_range(75, 2): 38 + 37 envs -- the shards differ by one env
        lo, hi = shard_range(n_total, world, rank)
        N = hi - lo
    tg = TransitionGather(obs_dim, g, N, world, rank, "cpu", allreduce_stats=allreduce)
    assert tg.n_all == [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)] if unequal \
        else tg.n_all == [N] * world
    assert tg.n_total == sum(tg.n_all)
    for it in range(3):
        chunk = FakeChunk(obs_dim, K, N, rank)
        chunk.act += it
        stats = torch.tensor([1.5 * (rank + 1), rank, K * N, it], dtype=torch.float64)
        tg.submit(chunk, it & 1, stats)
    tg.finish()
    ok = True
    if rank == 0:
        for src in range(world):
            ref = FakeChunk(obs_dim, K, tg.n_all[src], src)
            ref.act += 2
            obs, act, rew, obs2, done = tg.unpack(src)
            ok &= torch.equal(obs, ref.obs[:, K - g:]) and torch.equal(obs2, ref.obs2[:, K - g:])
            ok &= torch.equal(act, ref.act[K - g:]) and torch.equal(rew, ref.rew[K - g:])
            ok &= torch.equal(done, ref.done[K - g:])
    exp = torch.tensor([1.5 * sum(r + 1 for r in range(world)), sum(range(world)), K * tg.n_total, 2 * world],
                       dtype=torch.float64)
    if allreduce or rank == 0:
        ok &= torch.equal(tg.global_stats, exp)
    ok &= tg.chunks_gathered == 3
    if rank == 0:
        # what the learner of rl_train_sharded_ddpg appends to its replay ring: a chunk over the received columns
        got = TransitionChunk.from_columns(*tg.unpack(world - 1))
        nl = tg.n_all[world - 1]
        ok &= (got.K, got.N, got.obs_dim) == (g, nl, obs_dim) and got.act.stride(0) == nl
    # parameter sync (MpiAdam.sync, ddpg_editted.py:331-336): the views every rank's policy reads alias the flat array
    flat = torch.arange(10, dtype=torch.float32) + 100.0 * rank
    view = flat[2:8].view(2, 3)
    broadcast_flat(flat, src=0)
    ok &= torch.equal(view, (torch.arange(10, dtype=torch.float32))[2:8].view(2, 3))
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("obs_dim,g,allreduce,unequal", [(2, 3, True, False), (3, 12, False, False), (2, 5, False, True),
                                                         (3, 2, True, True)])
def test_gather_and_allreduce_world2_gloo(tmp_path, obs_dim, g, allreduce, unequal):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, obs_dim, g, str(tmp_path), allreduce, unequal), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]


def test_gather_world8_gloo_unequal_shards(tmp_path):
    """BASELINE configs[4] in miniature: 8 ranks, the shard sizes of shard_range(524288 + 3, 8, r) scaled down by 4096
    (three ranks own one env more), one gather per chunk to the learner, statistics summed out of the payloads."""
    n_total = (524288 + 3 * 4096) // 4096                      # 131 = 3 x 17 + 5 x 16
    sizes = [shard_range(n_total, 8, r)[1] - shard_range(n_total, 8, r)[0] for r in range(8)]
    big = [shard_range(524288 + 3, 8, r)[1] - shard_range(524288 + 3, 8, r)[0] for r in range(8)]
    assert sizes == [17, 17, 17, 16, 16, 16, 16, 16] and [b - 65536 for b in big] == [1, 1, 1, 0, 0, 0, 0, 0]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(8, port, 2, 4, str(tmp_path), False, True, n_total), nprocs=8, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(8)] == ["1"] * 8
