"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/ssc.h declares; the ctypes table in _ffi.py covers the same set.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ssc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssc_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from smartstartcontinuous_amd.build import build
    return build()


def test_header_symbols_exported_and_bound(built):
    from smartstartcontinuous_amd import _ffi
    syms = declared_symbols()
    assert "ssc_rollout" in syms and "ssc_mc_step" in syms
    handle = ctypes.CDLL(built)
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in ssc.h but not exported by libssc.so"
    assert sorted(_ffi._SIGNATURES) == syms, "ctypes table and ssc.h disagree"


def test_version_and_defaults(built):
    from smartstartcontinuous_amd import _ffi
    lib = _ffi.lib()
    assert lib.ssc_version() == 108
    p = _ffi.default_params(_ffi.SSC_ENV_MOUNTAINCAR, 0.4, 1000)
    assert abs(p.power - 0.0006) < 1e-9 and p.max_episode_steps == 1000
    assert abs(p.goal_position - 0.45) < 1e-7 and abs(p.min_position + 1.2) < 1e-7
    with pytest.raises(_ffi.SscError) as e:
        _ffi.default_params(7)
    assert e.value.code == _ffi.SSC_EINVAL and "unknown env kind" in str(e.value)


def test_argument_validation_without_gpu(built):
    """Bad arguments are rejected on the host before any HIP call."""
    from smartstartcontinuous_amd import _ffi
    lib = _ffi.lib()
    p = _ffi.default_params(_ffi.SSC_ENV_MOUNTAINCAR)
    rc = lib.ssc_mc_step(ctypes.byref(p), -1, None, None, None, None, None, None, None)
    assert rc == _ffi.SSC_EINVAL
    rc = lib.ssc_mc_step(ctypes.byref(p), 8, None, None, None, None, None, None, None)
    assert rc == _ffi.SSC_EINVAL and b"NULL" in lib.ssc_last_error()
    p.max_position = 3.0
    rc = lib.ssc_mc_step(ctypes.byref(p), 0, None, None, None, None, None, None, None)
    assert rc == _ffi.SSC_EUNSUPPORTED
    assert lib.ssc_mc_step(ctypes.byref(_ffi.default_params(0)), 0, None, None, None, None, None, None, None) == 0


def test_argument_validation_of_the_dataset_and_training_entry_points(built):
    """Negative sizes, NULL pointers, unsupported column counts and too-small workspaces come back as SSC_EINVAL
    with a message -- before any HIP call, so this runs without a GPU.  Empty inputs are fine where they mean
    'nothing to do'."""
    from smartstartcontinuous_amd import _ffi
    lib = _ffi.lib()
    log = _ffi.TransitionLog()
    E = _ffi.SSC_EINVAL
    assert lib.ssc_dataset_scan(None, 4, 8, None, None, None, 0, None) == E
    assert lib.ssc_dataset_scan(ctypes.byref(log), -1, 8, None, None, None, 0, None) == E
    assert lib.ssc_dataset_scan(ctypes.byref(log), 4, 8, None, None, None, 0, None) == E and b"d_off" in lib.ssc_last_error()
    assert lib.ssc_dataset_scan_workspace_bytes(65536) == (65536 // 64 + 1) * 8 and lib.ssc_dataset_scan_workspace_bytes(-1) == 0
    assert lib.ssc_dataset_build(ctypes.byref(log), 0, 4, 8, None, None, 10, None, None, None, None) == E      # obs_dim
    assert lib.ssc_dataset_build(ctypes.byref(log), 2, 4, 8, None, None, 10, None, None, None, None) == E      # NULL
    assert lib.ssc_dataset_build(ctypes.byref(log), 2, 1, 8, None, None, 10, None, None, None, None) == 0      # K = 1: no rows
    assert lib.ssc_dataset_build(ctypes.byref(log), 2, 4, 0, None, None, 10, None, None, None, None) == 0      # no envs
    assert lib.ssc_column_stats_workspace_bytes(3) == 2048 * 3 * 8 and lib.ssc_column_stats_workspace_bytes(65) == 0
    assert lib.ssc_column_stats(None, 10, 65, None, None, None, 0, None) == E
    assert lib.ssc_column_stats(None, 0, 3, None, None, None, 0, None) == E and b"at least one row" in lib.ssc_last_error()
    assert lib.ssc_zscore(None, 5, 3, None, None, None, 2, 0, None) == E                                       # out_stride < cols
    assert lib.ssc_zscore(None, 0, 3, None, None, None, 4, 1, None) == 0
    assert lib.ssc_zscore_concat(None, 0, None, None, None, 1, None, None, 5, None, None) == E                     # cols_x < 1
    assert lib.ssc_zscore_concat(None, 2, None, None, None, 1, None, None, 0, None, None) == 0
    assert lib.ssc_add_noise(None, 5, 300, None, 0.01, 1, 0, None) == E
    assert lib.ssc_add_noise(None, 5, 3, None, 0.01, 1, 1 << 50, None) == E and b"stream_id" in lib.ssc_last_error()
    assert lib.ssc_add_noise(None, 0, 3, None, 0.01, 1, 0, None) == 0
    net = _ffi.MlpTrainDesc()
    net.n_layers = 2
    net.dims[0], net.dims[1], net.dims[2] = 3, 32, 2
    assert lib.ssc_mlp_train_workspace_bytes(ctypes.byref(net), 512) > 0 and lib.ssc_mlp_train_workspace_bytes(None, 512) == 0
    assert lib.ssc_mlp_train_steps(ctypes.byref(net), None, None, None, 512, 3, None, None, 0, None) == E       # NULL parameters
    assert lib.ssc_mlp_train_steps(ctypes.byref(net), None, None, None, 0, 3, None, None, 0, None) == E         # batch 0
    net.n_layers = 9
    assert lib.ssc_mlp_train_steps(ctypes.byref(net), None, None, None, 512, 1, None, None, 0, None) == E and \
        b"n_layers" in lib.ssc_last_error()


def test_struct_layouts_match_header(built, tmp_path):
    """sizeof() of every descriptor struct as the C compiler sees it == ctypes.sizeof."""
    import subprocess
    from smartstartcontinuous_amd import _ffi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "ssc.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(ssc_env_params),sizeof(ssc_actor_desc),sizeof(ssc_ou_desc),sizeof(ssc_policy_desc),'
                   'sizeof(ssc_rollout_state),sizeof(ssc_transition_log),sizeof(ssc_episode_ring),'
                   'sizeof(ssc_mlp_desc),sizeof(ssc_norm),sizeof(ssc_mpc_problems),sizeof(ssc_critic_desc),sizeof(ssc_ddpg_desc),sizeof(ssc_replay_view),sizeof(ssc_mlp_train_desc),sizeof(ssc_replay_ring),sizeof(ssc_smartstart_step));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    expect = [ctypes.sizeof(c) for c in (_ffi.EnvParams, _ffi.ActorDesc, _ffi.OuDesc, _ffi.PolicyDesc,
                                         _ffi.RolloutState, _ffi.TransitionLog, _ffi.EpisodeRing, _ffi.MlpDesc,
                                         _ffi.Norm, _ffi.MpcProblems, _ffi.CriticDesc,
                                         _ffi.DdpgDesc, _ffi.ReplayView, _ffi.MlpTrainDesc, _ffi.ReplayRing, _ffi.SmartStartStep)]
    assert sizes == expect


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file of the product package mentions it."""
    pkg = os.path.join(ROOT, "smartstartcontinuous_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "libssc_oracle" not in text and "host_harness" not in text.replace(
                    "tests/host_harness", ""), f
