"""Random-policy data collection for the dynamics model, with the reference's names.

The reference gathers its training and validation sets with ``perform_rollouts`` -> ``CollectSamples``
(NN_Dynamics_Model/helper_funcs.py:19-27, collect_samples_threaded.py:8-111: 25 / 20 rollouts of 333 steps
over ``multiprocessing.Pool(8)``, each stopping at its first terminal step), formats them with
``generate_training_data_inputs`` / ``generate_training_data_outputs`` (data_manipulation.py:58-88), optionally
adds noise (helper_funcs.py:10-17) and z-scores the three arrays (NND_MB_agent.py:230-319).

Here ALL rollouts are one fused launch (``ssc_rollout``, one env per rollout) and the formatting runs on the
device too: ``ssc_dataset_scan`` / ``ssc_dataset_build`` compact the SoA chunk into row-major
``dataX, dataY, dataZ``; ``ssc_column_stats`` / ``ssc_zscore`` / ``ssc_add_noise`` do :302-319 and the noise.
``CollectSamples.collect_samples`` still returns the reference's lists of per-rollout numpy arrays."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _ffi
from .vec_env import RandomPolicy, VecEnv


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Policy_Random:
    """NN_Dynamics_Model/policy_random.py:3-15.  ``get_action`` is the host protocol; ``CollectSamples``
    recognises the class and runs the device random policy (same U(low, high), Philox-keyed) instead."""

    def __init__(self, env):
        self.env = env
        self.low_val = env.action_space.low
        self.high_val = env.action_space.high
        self.shape = env.action_space.shape

    def get_action(self, observation):
        return np.random.uniform(self.low_val, self.high_val, self.shape), 0


class TrainingSet:
    """Device-resident ``(dataX, dataY, dataZ)`` of one collection (row-major fp32 views of ``rows`` rows) plus
    the rollout lengths they came from."""

    def __init__(self, dataX, dataY, dataZ, lens, offsets):
        self.dataX, self.dataY, self.dataZ, self.lens, self.offsets = dataX, dataY, dataZ, lens, offsets

    def __len__(self):
        return self.dataX.shape[0]

    def numpy(self):
        return dict(dataX=self.dataX.double().cpu().numpy(), dataY=self.dataY.double().cpu().numpy(),
                    dataZ=self.dataZ.double().cpu().numpy())

    def normalised(self, norm):
        """(inputs [rows, d + a], outputs [rows, d]) z-scored with EXISTING statistics (``norm``: the dict of
        NND_MB_agent / DynamicsModel) -- how aggregated rows enter a retraining (NND_MB_agent.py:455-461)."""
        dev = self.dataX.device
        st = lambda k: torch.as_tensor(np.asarray(norm[k], np.float64).reshape(-1), device=dev)
        d, a = self.dataX.shape[1], self.dataY.shape[1]
        inputs = zscore_concat(self.dataX, st("mean_x"), st("std_x"), self.dataY, st("mean_y"), st("std_y"))
        return inputs, zscore_into(self.dataZ, st("mean_z"), st("std_z"), torch.empty_like(self.dataZ), 0)


def _scan(chunk):
    """(lens [n] i32, offsets [n+1] i64) of the first episode segment of every env of ``chunk``."""
    lib = _ffi.lib()
    dev, n = chunk.act.device, chunk.N
    lens = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(int(lib.ssc_dataset_scan_workspace_bytes(n)), 8), dtype=torch.uint8, device=dev)
    log = chunk.as_struct()
    with torch.cuda.device(dev):
        _ffi.check(lib.ssc_dataset_scan(ctypes.byref(log), chunk.K, n, _ffi.ptr(lens), _ffi.ptr(off), _ffi.ptr(ws),
                                        ws.numel(), _stream()))
    return lens, off


def dataset_from_chunk(chunk):
    """(s_i, a_i, s_{i+1} - s_i) rows of every env's first episode segment of ``chunk`` (data_manipulation.py:58-88
    applied to the rollouts of collect_samples_threaded.py:52-111), on the device."""
    lib = _ffi.lib()
    dev, n, K, d = chunk.act.device, chunk.N, chunk.K, chunk.obs_dim
    lens, off = _scan(chunk)
    log = chunk.as_struct()
    with torch.cuda.device(dev):
        rows = int(off[n].item())                  # the one device -> host read: the size of the result
        X = torch.empty((rows, d), dtype=torch.float32, device=dev)
        Y = torch.empty((rows, 1), dtype=torch.float32, device=dev)
        Z = torch.empty((rows, d), dtype=torch.float32, device=dev)
        _ffi.check(lib.ssc_dataset_build(ctypes.byref(log), d, K, n, _ffi.ptr(lens), _ffi.ptr(off), rows, _ffi.ptr(X),
                                         _ffi.ptr(Y), _ffi.ptr(Z), _stream()))
    return TrainingSet(X, Y, Z, lens, off)


def column_stats(x):
    """(mean, std) per column of a row-major fp32 device matrix as f64 device tensors (NND_MB_agent.py:302-304)."""
    lib = _ffi.lib()
    x = x.contiguous()
    rows, cols = x.shape
    mean = torch.empty(cols, dtype=torch.float64, device=x.device)
    std = torch.empty(cols, dtype=torch.float64, device=x.device)
    ws = torch.empty(max(int(lib.ssc_column_stats_workspace_bytes(cols)), 8), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _ffi.check(lib.ssc_column_stats(_ffi.ptr(x), rows, cols, _ffi.ptr(mean), _ffi.ptr(std), _ffi.ptr(ws), ws.numel(),
                                        _stream()))
    return mean, std


def zscore_into(x, mean, std, out, col0=0):
    """out[:, col0:col0+cols] = nan_to_num((x - mean) / std) (NND_MB_agent.py:303-305, :318)."""
    lib = _ffi.lib()
    x = x.contiguous()
    rows, cols = x.shape
    if out.shape[0] != rows or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 matrix with one row per data row")
    with torch.cuda.device(x.device):
        _ffi.check(lib.ssc_zscore(_ffi.ptr(x), rows, cols, _ffi.ptr(mean), _ffi.ptr(std), _ffi.ptr(out), out.shape[1],
                                  col0, _stream()))
    return out


def zscore_concat(x, mean_x, std_x, y, mean_y, std_y, out=None):
    """[nan_to_num((x - mean_x) / std_x) | nan_to_num((y - mean_y) / std_y)] as one [rows, cols_x + cols_y] matrix
    (NND_MB_agent.py:303-318: the two z-scored matrices, np.concatenate(..., axis=1)) in a single pass."""
    lib = _ffi.lib()
    x, y = x.contiguous(), y.contiguous()
    rows, cx = x.shape
    cy = y.shape[1]
    if y.shape[0] != rows:
        raise ValueError("x and y must have one row per data row")
    if out is None:
        out = torch.empty((rows, cx + cy), dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (rows, cx + cy) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 [rows, cols_x + cols_y] matrix")
    with torch.cuda.device(x.device):
        _ffi.check(lib.ssc_zscore_concat(_ffi.ptr(x), cx, _ffi.ptr(mean_x), _ffi.ptr(std_x), _ffi.ptr(y), cy, _ffi.ptr(mean_y),
                                         _ffi.ptr(std_y), rows, _ffi.ptr(out), _stream()))
    return out


def add_noise_device(x, noise_to_signal, seed, stream_id=0, mean=None):
    """helper_funcs.add_noise (NN_Dynamics_Model/helper_funcs.py:10-17) in place on a device matrix."""
    lib = _ffi.lib()
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    if mean is None:
        mean, _ = column_stats(x)
    with torch.cuda.device(x.device):
        _ffi.check(lib.ssc_add_noise(_ffi.ptr(x), x.shape[0], x.shape[1], _ffi.ptr(mean), float(noise_to_signal),
                                     int(seed), int(stream_id), _stream()))
    return x


class CollectSamples:
    """collect_samples_threaded.py:8-111 / collect_samples.py.  ``env`` is a ``VecEnv`` or a ``SingleEnvView``;
    every rollout runs in its own device env (the reference deep-copies ``env`` per rollout, :60), all of them in
    one ``ssc_rollout`` launch.  Calls are numbered: call ``j`` uses seed ``seed + j``, so training and validation
    collections differ like two passes over the reference's RNG stream would."""

    def __init__(self, env, policy=None, visualize_rollouts=False, dt_steps=1, dt_from_xml=1, seed=None):
        if visualize_rollouts:
            raise NotImplementedError("rendering is out of scope (SURVEY.md section 8: visualisation)")
        if policy is not None and not isinstance(policy, (Policy_Random, RandomPolicy)):
            raise NotImplementedError("CollectSamples runs the random policy (policy_random.py); use VecEnv.rollout "
                                      "for actor or MPC rollouts")
        self.main_env = env.vec if hasattr(env, "vec") else env
        self.policy = policy
        self.stateDim = self.main_env.observation_space.shape[0]
        self.actionDim = self.main_env.action_space.shape[0]
        self.dt_steps, self.dt_from_xml = dt_steps, dt_from_xml
        self.seed = int(self.main_env._seed if seed is None else seed)
        self._calls = 0

    def _rollout_chunk(self, num_rollouts, steps_per_rollout):
        m = self.main_env
        spec_id = m.spec.id if m.spec is not None else "MountainCarContinuous-v0"
        venv = VecEnv(spec_id, num_rollouts, device=m.device, power_scalar=m.power_scalar,
                      max_episode_steps=int(m.params.max_episode_steps), seed=self.seed + self._calls)
        self._calls += 1
        return venv.rollout(steps_per_rollout, policy=RandomPolicy())

    def collect_dataset(self, num_rollouts, steps_per_rollout):
        """-> ``TrainingSet`` on the device (no host copy of the rollouts)."""
        return dataset_from_chunk(self._rollout_chunk(num_rollouts, steps_per_rollout))

    def collect_samples(self, num_rollouts, steps_per_rollout):
        """-> (list_observations, list_actions, list_starting_states, []) exactly as :46-50: one
        [L_i, stateDim] / [L_i, actionDim] float64 array per rollout."""
        chunk = self._rollout_chunk(num_rollouts, steps_per_rollout)
        lens = _scan(chunk)[0].cpu().numpy()
        obs = chunk.obs.permute(2, 1, 0).double().cpu().numpy()      # [n, K, d]
        act = chunk.act.t().double().cpu().numpy()[:, :, None]       # [n, K, 1]
        observations = [obs[i, :lens[i]] for i in range(num_rollouts)]
        actions = [act[i, :lens[i]] for i in range(num_rollouts)]
        starting = [obs[i, 0].copy() for i in range(num_rollouts)]
        return observations, actions, starting, []


def perform_rollouts(policy, num_rollouts, steps_per_rollout, visualize_rollouts, CollectSamples, env, dt_steps=1,
                     dt_from_xml=1):
    """NN_Dynamics_Model/helper_funcs.py:19-27."""
    c = CollectSamples(env, policy, visualize_rollouts, dt_steps, dt_from_xml)
    return c.collect_samples(num_rollouts, steps_per_rollout)


def generate_training_data_inputs(states0, controls0):
    """NN_Dynamics_Model/data_manipulation.py:58-79 for host lists of rollouts."""
    new_states = [np.asarray(s)[0:len(s) - 1, :] for s in states0]
    new_controls = [np.asarray(c)[0:len(c) - 1, :] for c in controls0]
    return np.concatenate(new_states, axis=0), np.concatenate(new_controls, axis=0)


def generate_training_data_outputs(states):
    """NN_Dynamics_Model/data_manipulation.py:81-88."""
    return np.concatenate([np.asarray(s)[1:len(s), :] - np.asarray(s)[0:len(s) - 1, :] for s in states], axis=0)
