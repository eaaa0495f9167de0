"""Seeded prefixes of the randomised sweeps in tools/fuzz_*.py as GPU tests: random launch shapes / network shapes /
problem shapes through the C ABI against the oracle (the full sweeps and their logs: profiles/r02/fuzz/).  The seeds are
the tools' defaults, so every case here is one that the committed logs cover."""
import os
import runpy
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,expect", [
    ("fuzz_rollout.py", 16, "Pendulum random-policy rollout: 8 random configurations ok"),
    ("fuzz_actor_rollout.py", 14, "actor rollout: 14 random configurations ok"),
    ("fuzz_actor_wide.py", 12, "wide actor rollout: 12 random configurations ok"),
    ("fuzz_dyn_shapes.py", 12, "forward simulation: 12 random shapes ok"),
    ("fuzz_dyn_train.py", 10, "dynamics-model training step: 10 random shapes ok"),
    ("fuzz_mpc_score.py", 10, "MPC scoring: 10 random configurations ok"),
])
def test_randomised_sweep_prefix(tool, cases, expect, capsys, monkeypatch):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    monkeypatch.setattr(sys, "argv", [tool, str(cases)])
    runpy.run_path(os.path.join(ROOT, "tools", tool), run_name="__main__")     # raises AssertionError on a mismatch
    assert expect in capsys.readouterr().out
