import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multirank: compares the output of child rank processes started by conftest.py")


MULTIRANK_DIR = os.path.join(ROOT, "tests", "_build", "multirank")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_multirank_jobs(out_dir=MULTIRANK_DIR, timeout=900):
    """Start the rank processes of tests/test_gpu_multirank.py as FRESH children and wait for them: one world-1 process,
    the two ranks of a world-2 group (all on GPU 0, gloo) and `bench.py --gpus 2 --backend gloo` (which spawns its own two
    ranks).  Called from pytest_collection_finish, i.e. before any test body -- before this process has touched the GPU:
    a process that has initialised the GPU must not start other programs on this pool.  Every child's stdout / stderr and
    exit code land in ``out_dir``; the tests read them."""
    import json
    import shutil
    import subprocess
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    worker = os.path.join(ROOT, "tests", "multirank_worker.py")
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    status = {}

    def wait(name, procs):
        codes = []
        for p, log in procs:
            try:
                codes.append(p.wait(timeout=timeout))
            except subprocess.TimeoutExpired:
                p.kill()
                codes.append("timeout")
            log.close()
        status[name] = codes

    for world in (1, 2):
        port = str(_free_port())
        procs = []
        for rank in range(world):
            log = open(os.path.join(out_dir, f"world{world}_rank{rank}.log"), "w")
            env = dict(base, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT=port)
            procs.append((subprocess.Popen([sys.executable, worker, out_dir], env=env, stdout=log, stderr=subprocess.STDOUT,
                                           cwd=ROOT), log))
        wait(f"world{world}", procs)
    log = open(os.path.join(out_dir, "bench_gpus2.log"), "w")
    err = open(os.path.join(out_dir, "bench_gpus2.err"), "w")
    env = {k: v for k, v in base.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                          "--warmup", "1", "--settle-launches", "0"],
                         env=env, stdout=log, stderr=err, cwd=ROOT)
    wait("bench_gpus2", [(p, log)])
    err.close()
    with open(os.path.join(out_dir, "status.json"), "w") as f:
        json.dump(status, f)
    return status


def pytest_collection_finish(session):
    if not any(item.get_closest_marker("multirank") for item in session.items):
        return
    import torch
    if torch.cuda.device_count() == 0:      # (counting devices does not initialise the GPU)
        return                               # the tests fail with "needs a GPU" themselves
    run_multirank_jobs()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_clib():
    """ctypes handle to oracle/_build/libssc_oracle.so (built on demand with gcc)."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "libssc_oracle.so")
    src = os.path.join(ROOT, "oracle", "ssc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)
