"""The N > 1 path in front of the driver: ``rl_train_sharded_ddpg`` with TWO rank processes
against the world-1 run of the same global env-id space, and ``bench.py --gpus 2`` self-spawned.

The rank processes are fresh children started by tests/conftest.py (``run_multirank_jobs``) when the collection is
over -- before this process touches the GPU; gloo group, both ranks on GPU 0, the gathered payloads staged through pinned
host memory (``TransitionGather(host_staging)``).  What this file checks is what the reference's own exchange sites do
(rollout gather: NN_Dynamics_Model/collect_samples_threaded.py:31-50; parameter sync: ddpg_editted.py:331-336):
every rank's envs behave as if the whole id space ran on one GPU."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu, pytest.mark.multirank]

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "multirank")
COLS = ("obs", "act", "rew", "done", "obs2")


def _tail(name, n=30):
    p = os.path.join(OUT, name)
    return "".join(open(p).readlines()[-n:]) if os.path.exists(p) else "<no log>"


@pytest.fixture(scope="module")
def runs():
    if torch.cuda.device_count() == 0:
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    sp = os.path.join(OUT, "status.json")
    assert os.path.exists(sp), "conftest.run_multirank_jobs did not run"
    status = json.load(open(sp))
    assert status["world1"] == [0], _tail("world1_rank0.log")
    assert status["world2"] == [0, 0], _tail("world2_rank0.log") + _tail("world2_rank1.log")
    one = np.load(os.path.join(OUT, "world1_rank0.npz"))
    two = [np.load(os.path.join(OUT, f"world2_rank{r}.npz")) for r in range(2)]
    return status, one, two


def test_shards_are_unequal_and_cover_the_id_space(runs):
    _, one, two = runs
    assert (int(one["lo"]), int(one["hi"])) == (0, 769)
    assert [(int(t["lo"]), int(t["hi"])) for t in two] == [(0, 385), (385, 769)]


@pytest.mark.parametrize("mode", ["sync", "pipelined"])
def test_two_ranks_reproduce_the_one_rank_run_bit_for_bit(runs, mode):
    """Union of the two ranks' transition logs == the world-1 log of the same id range, chunk by chunk; so are the final
    env states, the finished episodes and what every rank ends up acting with ([actor | epsilon]); the learner's replay
    ring, critic and losses equal the world-1 learner's."""
    _, one, two = runs
    for c in COLS:
        whole = one[f"{mode}_log_{c}"]                       # [chunks, (obs_dim,) K, n]
        parts = np.concatenate([t[f"{mode}_log_{c}"] for t in two], axis=-1)
        assert whole.shape == parts.shape and whole.shape[-1] == 769 and whole.shape[0] == 6
        assert np.array_equal(whole.view(np.uint8), parts.view(np.uint8)), f"{mode}: column {c} differs"
    for k in ("s0", "s1", "ou_x", "steps"):
        assert np.array_equal(one[f"{mode}_{k}"], np.concatenate([t[f"{mode}_{k}"] for t in two]))
    eps = np.concatenate([t[f"{mode}_episodes"] for t in two])
    assert np.array_equal(one[f"{mode}_episodes"], eps[np.lexsort((eps[:, 1], eps[:, 0]))]) and len(eps) >= 4 * 769
    assert one[f"{mode}_stats"][2] == 769 * 48 * 6 == sum(t[f"{mode}_stats"][2] for t in two)
    # every rank acts with the learner's parameters and epsilon
    for t in two:
        assert np.array_equal(t[f"{mode}_actor_sync"], one[f"{mode}_actor_sync"])
    # the learner (rank 0): same ring, same critic, same losses as the single-GPU learner
    assert int(two[0][f"{mode}_replay_count"]) == int(one[f"{mode}_replay_count"]) == 6 * 8 * 769
    for k in ("replay_s", "replay_a", "replay_r", "replay_t", "replay_s2", "critic", "losses"):
        assert np.array_equal(two[0][f"{mode}_{k}"], one[f"{mode}_{k}"]), f"{mode}: {k} differs"
    assert f"{mode}_replay_s" not in two[1].files
    # epsilon decayed once per generation of 769 finished episodes (4 full generations of 60-step episodes in 288 steps)
    assert abs(float(one[f"{mode}_host_epsilon"]) - 0.8 ** 4) < 1e-12
    assert abs(float(two[1][f"{mode}_host_epsilon"]) - 0.8 ** 4) < 1e-7          # the broadcast fp32 copy
    # the two modes are different schedules (generation j-1 against j-2) ...
    assert not np.array_equal(one["sync_log_act"], one["pipelined_log_act"])


def test_pipelined_run_is_deterministic(runs):
    _, one, two = runs
    for r in [one] + two:
        for k in r.files:
            if k.startswith("pipelined_again_"):
                assert np.array_equal(r[k], r["pipelined_" + k[len("pipelined_again_"):]]), k


def test_the_ring_of_the_sharded_learner_is_step_major_over_the_global_ids(runs):
    """record number = step * N_total + global env id (``ssc_replay_append_shard``): the learner's ring rows are the
    last 8 steps of every chunk of the two ranks' logs interleaved by GLOBAL env id."""
    _, one, two = runs
    obs = np.concatenate([t["sync_log_obs"] for t in two], axis=-1)      # [chunks, 2, K, 769]
    tail = obs[:, :, -8:, :].transpose(0, 2, 3, 1).reshape(-1, 2)        # chunk, step, env
    assert np.array_equal(two[0]["sync_replay_s"][: tail.shape[0]], tail)
    rew = np.concatenate([t["sync_log_rew"] for t in two], axis=-1)[:, -8:, :].reshape(-1)
    assert np.array_equal(two[0]["sync_replay_r"][: rew.shape[0]], rew)


def test_bench_gpus_2_self_spawned_prints_one_line(runs):
    """`python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --settle-launches 0`: the launcher form the driver
    uses, two ranks, ONE JSON line from rank 0, the payload statistics adding up to every rank's env-steps (asserted by
    bench.py itself after the line: exit code 0)."""
    status, _, _ = runs
    assert status["bench_gpus2"] == [0], _tail("bench_gpus2.err")
    lines = [l for l in open(os.path.join(OUT, "bench_gpus2.log")).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["global_envs"] == 2 * 65536 and r["config"]["backend"].startswith("gloo")
    assert r["config"]["gather"] == "bounded" and r["config"]["gather_steps_per_message"] >= 1
    assert r["value"] > 0 and r["roofline"]["traffic"] is None and "cpu_baseline" not in r
