"""bench.py's bookkeeping (CPU): launch-time statistics, the source stamp of the PMC traffic figure and its staleness
rule -- the parts of the measurement contract that do not need a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def test_dist_stats():
    s = bench.dist_stats([0.3, 0.1, 0.2, 0.5, 0.4, 0.6, 0.9, 0.7, 0.8, 1.0])
    assert s["n"] == 10 and s["min"] == 0.1 and s["max"] == 1.0
    assert s["median"] == 0.6 and s["p90"] == 1.0 and abs(s["mean"] - 0.55) < 1e-12
    assert bench.dist_stats([2.0]) == {"n": 1, "min": 2.0, "median": 2.0, "p90": 2.0, "max": 2.0, "mean": 2.0}


def test_traffic_is_reported_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 PMC pass; it is reported only when that pass ran on the
    rollout sources being benchmarked (sha-256 stamp), otherwise the line says which file is stale."""
    sha = bench.source_sha()
    assert len(sha) == 16 and int(sha, 16) >= 0
    # the committed profile must be current for the committed kernel (a stale one is a hygiene failure of the repo)
    got = bench.profiled_traffic()
    assert got is not None and got[0] is not None, got
    alg = bench.N_ENVS_PER_GPU * bench.CHUNK * bench.BYTES_PER_STEP
    assert 0.99 * alg <= got[0] <= 1.05 * alg          # nothing re-read or written twice
    # a fake tree: one profile stamped with another sha -> stale; add a matching one -> its value wins
    csrc = tmp_path / "smartstartcontinuous_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in ("rollout.hip", "ssc_device.h"):
        (csrc / f).write_bytes(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    (tmp_path / "profiles" / "old").mkdir(parents=True)
    (tmp_path / "profiles" / "old" / "traffic.json").write_text(json.dumps(
        {"write_bytes": 1.0, "fetch_bytes_corrected": 2.0, "source_sha": "0123456789abcdef"}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.source_sha() == sha
    value, note = bench.profiled_traffic()
    assert value is None and "stale" in note and "0123456789abcdef" in note
    (tmp_path / "profiles" / "zz_new").mkdir()
    (tmp_path / "profiles" / "zz_new" / "traffic.json").write_text(json.dumps(
        {"write_bytes": 10.0, "fetch_bytes_corrected": 5.0, "source_sha": sha}))
    assert bench.profiled_traffic() == (15.0, os.path.join("profiles", "zz_new", "traffic.json"))
    (csrc / "rollout.hip").write_bytes(b"// edited\n")          # the kernel changes: every committed figure goes stale
    value, note = bench.profiled_traffic()
    assert value is None and "stale" in note


def test_issue_and_pipe_stamps_are_current_for_the_committed_kernels():
    """roofline.valu_issue (config 3) and roofline.pipe_busy (config 4) come from committed PMC passes too; a stamp that
    no longer matches the kernel source it was measured on is a hygiene failure of the repo."""
    sys.path.insert(0, ROOT)
    import bench
    vi, pb = bench.profiled_valu_issue(), bench.profiled_pipe_busy()
    assert vi is not None and "stale" not in vi and 0.5 < vi["frac"] < 1.0, vi
    assert pb is not None and "stale" not in pb and 0.5 < pb["frac"] < 1.0, pb
    assert abs(pb["SQ_VALU_MFMA_BUSY_CYCLES"] - 16 * pb["SQ_INSTS_MFMA"]) < 1.0        # 16 cycles per 16x16x32 bf16 MFMA


def _run_bench(cmd, extra_env=None):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_starts_its_own_ranks_when_there_is_no_launcher():
    """Both launch forms of the N > 1 bench reach the rendezvous and rank 0 prints exactly ONE JSON line: (i) plain
    `python bench.py --gpus 2` (the way the driver starts the N = 1 line) spawns torch.distributed.run as a child
    process before anything touches a GPU and passes the exit code on; (ii) the torch.distributed.run form.  --dry-run
    stops after the rendezvous (gloo, CPU), which is as far as a box without GPUs can go."""
    r, lines = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["dry_run"] and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3
    port = bench.free_port()
    r, lines = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                           "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    # a failing rank makes the self-spawned run fail too (exit code relayed): without --dry-run there is no GPU here
    import torch
    if not torch.cuda.is_available():
        r, lines = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"])
        assert r.returncode != 0 and not lines
