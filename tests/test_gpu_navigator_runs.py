"""The navigator end to end against the only evidence the reference holds for it: the 12 saved runs of
smartstart/RLAgents/NND_MB_agent_main.py under data/nnd_mb_tests/ (fixture tests/golden/nnd_mb_runs.npz: per-episode
(steps, return) of every run + the state paths its navigator traversed).  The same script with this engine -- the
dynamics model trained on the reference's own dataX/Y/Z (``load_existing_training_data``), N = 500 candidates, horizon 4,
retraining on the aggregated replay data every episode -- must reach the goal as reliably and as fast as the reference's
29 goal-reaching episodes did: a STATISTICAL pin of dynamics-model training, forward simulation, MPC scoring and the
waypoint bookkeeping together (each is compared with the oracle separately in test_gpu_navigator.py)."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def reference_runs(golden_dir):
    runs = np.load(f"{golden_dir}/nnd_mb_runs.npz")
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    return runs, dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"])


def test_fixture_holds_the_reference_runs(reference_runs):
    runs, _ = reference_runs
    ok = runs["steps"] < 999
    assert len(runs["steps"]) == 35 and ok.sum() == 29          # runs 0-2 (earlier settings) never reached the goal
    assert runs["steps"][ok].min() == 68 and runs["steps"][ok].max() == 96
    assert 93.64 < runs["returns"][ok].min() and runs["returns"][ok].max() < 94.5
    assert len(runs["path_lens"]) == 24 and runs["path_states"].shape == (runs["path_lens"].sum(), 2)
    assert (runs["path_states"][np.cumsum(runs["path_lens"]) - 1, 0] >= 0.45).all()      # every path ends at the goal


@pytest.mark.gpu
@pytest.mark.parametrize("precision,seeds", [("f32", (1234, 1235, 1236)), ("bf16_mfma", (1234, 1235))])
def test_navigator_reaches_the_goal_like_the_reference_runs(reference_runs, precision, seeds):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    from exp_nnd_mb_runs import run
    runs, data = reference_runs
    ok = runs["steps"] < 999
    lo_n, hi_n = runs["steps"][ok].min(), runs["steps"][ok].max()              # 68 .. 96
    lo_r, hi_r = runs["returns"][ok].min(), runs["returns"][ok].max()          # 93.65 .. 94.49
    offs = np.concatenate([[0], np.cumsum(runs["path_lens"])])
    steps, rets = [], []
    for k, seed in enumerate(seeds):
        target = runs["path_states"][offs[k]:offs[k + 1]]                     # a path the reference's navigator traversed
        for n, r in run(precision, 10, seed, target, data):
            steps.append(n)
            rets.append(r)
    steps, rets = np.asarray(steps), np.asarray(rets)
    # every episode reaches the goal (the reference's did in 29 of 29 with these settings) ...
    assert (steps < 200).all(), steps
    # ... and the typical episode lies inside the reference's own range; single episodes may leave it by a few steps,
    # the targets being paths that were already navigated once
    assert lo_n <= np.median(steps) <= hi_n, (np.median(steps), steps)
    assert lo_r <= np.median(rets) <= hi_r, (np.median(rets), rets)
    assert (steps <= hi_n + 24).all() and (rets >= lo_r - 0.6).all(), (steps, rets)
