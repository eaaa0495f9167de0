"""Device-side pieces of the SmartStart navigator (smartstart/RLAgents/NND_MB_agent.py):
the learned dynamics model (``Dyn_Model``: ``feedforward_network`` + ``do_forward_sim``) and the
MPC over it (action sampling, ``generate_scores_add_delta``, action selection).

Everything here is a thin wrapper over the C ABI (include/ssc.h): tensors in, tensors out, on
``torch.cuda.current_stream()``; weights are fp32 tensors in TensorFlow layout ``W[in][out]``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _ffi


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_PREC = {"f32": _ffi.SSC_PREC_F32, "bf16_mfma": _ffi.SSC_PREC_BF16_MFMA}


class DynamicsModel:
    """``Dyn_Model`` forward parts (NN_Dynamics_Model/dynamics_model.py:14-50, 199-240).

    weights/biases: lists of fp32 arrays/tensors, ``weights[l]`` of shape [dims[l], dims[l+1]]
    (num_fc_layers hidden layers + the output layer, feedforward_network.py:3-23).
    norm: dict(mean_x, std_x, mean_y, std_y, mean_z, std_z) -- NND_MB_agent.py:302-315.
    """

    def __init__(self, weights, biases, norm, state_dim, act_dim, device="cuda", precision="f32"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DynamicsModel runs on the GPU only; there is no CPU fallback")
        self.lib = _ffi.lib()
        self.state_dim, self.act_dim = int(state_dim), int(act_dim)
        self.precision = precision
        self._ws = None
        self._image = None          # packed bf16 weight image of the MFMA path (ssc_dyn_prepare)
        self._image_stale = True
        self._mfma_ok = None        # does the MFMA kernel cover this network (decided on first use)
        self.set_weights(weights, biases)
        self.set_norm(norm)

    def set_weights(self, weights, biases):
        """Weights are kernel ARGUMENTS, not baked in: the navigator retrains the model every
        few episodes (NND_MB_agent.py:421-423) and the next call must see the new values."""
        self.W = [torch.as_tensor(w, dtype=torch.float32).to(self.device).contiguous() for w in weights]
        self.b = [torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous() for b in biases]
        if len(self.W) != len(self.b) or not 1 <= len(self.W) <= _ffi.SSC_MAX_LAYERS:
            raise ValueError("need 1..%d layers" % _ffi.SSC_MAX_LAYERS)
        d = _ffi.MlpDesc()
        d.n_layers = len(self.W)
        d.dims[0] = self.W[0].shape[0]
        for l, (w, b) in enumerate(zip(self.W, self.b)):
            if w.shape[0] != d.dims[l] or b.shape != (w.shape[1],):
                raise ValueError(f"layer {l}: inconsistent shapes {tuple(w.shape)} / {tuple(b.shape)}")
            d.dims[l + 1] = w.shape[1]
            d.W[l], d.b[l] = w.data_ptr(), b.data_ptr()
        self.desc = d
        self.in_dim, self.out_dim = d.dims[0], d.dims[d.n_layers]
        self._mfma_ok = None
        self._image = None
        if hasattr(self, "_adam"):
            del self._adam                    # Adam moments and step counter belong to the parameters they were built for
        self.invalidate()

    def save(self, path):
        """Weights, biases and normalisation statistics as one ``.npz`` (the stand-in for the TF saver of
        NND_MB_agent.py:470-480; TF checkpoints themselves are out of scope)."""
        arrays = {f"W{l}": w.cpu().numpy() for l, w in enumerate(self.W)}
        arrays.update({f"b{l}": b.cpu().numpy() for l, b in enumerate(self.b)})
        nm = self.norm
        sd, ad = self.state_dim, self.act_dim
        for k, n in (("mean_x", sd), ("std_x", sd), ("mean_y", ad), ("std_y", ad), ("mean_z", sd), ("std_z", sd)):
            arrays[k] = np.asarray([getattr(nm, k)[i] for i in range(n)], np.float64)
        np.savez(path, **arrays)

    def load(self, path):
        """Restore what ``save`` wrote (same layer sizes or not: the descriptors are rebuilt)."""
        z = np.load(path)
        n = sum(1 for k in z.files if k.startswith("W"))
        self.set_weights([z[f"W{l}"] for l in range(n)], [z[f"b{l}"] for l in range(n)])
        self.set_norm({k: z[k] for k in ("mean_x", "std_x", "mean_y", "std_y", "mean_z", "std_z")})
        if hasattr(self, "_adam"):
            del self._adam                    # the optimiser state belongs to the old parameters

    def invalidate(self):
        """The weights or statistics changed: the packed image of the MFMA path is rebuilt on the next
        call.  ``set_weights``, ``set_norm`` and ``train_step`` call this; code that writes into
        ``self.W`` / ``self.b`` in place must call it too."""
        self._image_stale = True

    def set_norm(self, norm):
        nm = _ffi.Norm()
        for key, n in (("mean_x", self.state_dim), ("std_x", self.state_dim), ("mean_y", self.act_dim),
                       ("std_y", self.act_dim), ("mean_z", self.state_dim), ("std_z", self.state_dim)):
            v = np.asarray(norm[key], np.float32).reshape(-1)
            if v.size != n:
                raise ValueError(f"norm[{key}] has {v.size} entries, expected {n}")
            arr = getattr(nm, key)
            for i in range(n):
                arr[i] = float(v[i])
        self.norm = nm
        self.invalidate()

    # ---- training (Dyn_Model.train, dynamics_model.py:52-171) ----------------------------------------
    def _train_desc(self, lr):
        if not hasattr(self, "_adam"):
            z = lambda t: torch.zeros_like(t)
            self._adam = dict(mW=[z(w) for w in self.W], vW=[z(w) for w in self.W], mb=[z(b) for b in self.b],
                              vb=[z(b) for b in self.b], t=torch.zeros(1, dtype=torch.int32, device=self.device))
        d = _ffi.MlpTrainDesc()
        d.n_layers = len(self.W)
        for l in range(len(self.W) + 1):
            d.dims[l] = self.desc.dims[l]
        for l in range(len(self.W)):
            d.W[l], d.b[l] = self.W[l].data_ptr(), self.b[l].data_ptr()
            d.mW[l], d.vW[l] = self._adam["mW"][l].data_ptr(), self._adam["vW"][l].data_ptr()
            d.mb[l], d.vb[l] = self._adam["mb"][l].data_ptr(), self._adam["vb"][l].data_ptr()
        d.adam_t = self._adam["t"].data_ptr()
        d.lr, d.beta1, d.beta2, d.epsilon = float(lr), 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults
        return d

    def train_step(self, X, Z, idx, lr=0.001, loss=None):
        """One Adam step on rows ``idx`` of the device data sets X [n, in], Z [n, out] (fp32, already
        normalised).  ``loss`` (a 1-element fp32 tensor) receives the batch MSE if given."""
        d = self._train_desc(lr)
        B = idx.numel()
        with torch.cuda.device(self.device):
            ws = self._workspace(self.lib.ssc_mlp_train_workspace_bytes(ctypes.byref(d), B))
            _ffi.check(self.lib.ssc_mlp_train_step(ctypes.byref(d), _ffi.ptr(X), _ffi.ptr(Z), _ffi.ptr(idx), B,
                                                   _ffi.ptr(loss), _ffi.ptr(ws), ws.numel(), _stream()))
        self.invalidate()

    def train_steps(self, X, Z, idx, lr=0.001):
        """``idx.shape[0]`` consecutive Adam steps, step k on rows ``idx[k]`` (int32 [n_steps, B] on the device);
        returns the per-step batch MSEs as a device tensor.  One C call, no host synchronisation."""
        d = self._train_desc(lr)
        n_steps, B = idx.shape
        idx = idx.contiguous()
        losses = torch.empty(n_steps, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(self.lib.ssc_mlp_train_workspace_bytes(ctypes.byref(d), B))
            _ffi.check(self.lib.ssc_mlp_train_steps(ctypes.byref(d), _ffi.ptr(X), _ffi.ptr(Z), _ffi.ptr(idx), B, n_steps,
                                                    _ffi.ptr(losses), _ffi.ptr(ws), ws.numel(), _stream()))
        self.invalidate()
        return losses

    def train(self, dataX, dataZ, dataX_new, dataZ_new, nEpoch, fraction_use_new, batchsize=512, lr=0.001,
              rng=None):
        """``Dyn_Model.train`` (dynamics_model.py:52-171): every batch mixes ``batchsize*fraction_use_new``
        rows of the new (aggregated) data with a walk through the shuffled old data.  The data sets are
        uploaded once; only index vectors travel per iteration.  Returns the mean training loss of the
        last epoch, like the reference's first return value."""
        rng = rng if rng is not None else np.random
        def f(a, cols):       # device tensors (collect_samples.TrainingSet) stay where they are
            if torch.is_tensor(a):
                return a.to(device=self.device, dtype=torch.float32).reshape(-1, cols)
            return torch.as_tensor(np.asarray(a, np.float32).reshape(-1, cols), device=self.device)
        n_old, n_new = len(dataX), len(dataX_new)
        X = torch.cat([f(dataX, self.in_dim), f(dataX_new, self.in_dim)])
        Z = torch.cat([f(dataZ, self.out_dim), f(dataZ_new, self.out_dim)])
        b_new = n_new if n_new < batchsize * fraction_use_new else int(batchsize * fraction_use_new)   # :60-67
        b_old = int(batchsize - b_new)
        perm_new = np.arange(n_new)
        epoch_losses = []
        for _ in range(nEpoch):
            old = rng.choice(np.arange(n_old), size=(n_old,), replace=False)                          # :78
            if b_old > 0:
                batches = []
                for batch in range(int(np.floor(n_old / b_old))):                                     # :82
                    new_idx = rng.randint(0, n_new, (b_new,)) if n_new > 0 else np.zeros(0, np.int64)   # :88
                    batches.append(np.concatenate([old[batch * b_old:(batch + 1) * b_old], n_old + perm_new[new_idx]]))
            else:
                batches = [n_old + perm_new[b * b_new:(b + 1) * b_new] for b in range(int(np.floor(n_new / b_new)))]
                perm_new = perm_new[rng.permutation(n_new)]                                           # :120-122
            if batches:       # the epoch's index vectors travel in one copy, its steps are enqueued by one call
                idx = torch.as_tensor(np.stack(batches).astype(np.int32), device=self.device)
                epoch_losses.append(self.train_steps(X, Z, idx, lr=lr))
        # the only device -> host read of the whole training: the mean loss of the last epoch
        return float(epoch_losses[-1].mean().item()) if epoch_losses else 0.0

    def run_validation(self, inputs, outputs, batchsize=512, precision="f32"):
        """``Dyn_Model.run_validation`` (dynamics_model.py:173-196): the mean over the floor(n / batchsize) full batches of the
        batch MSE between the network's outputs for ``inputs`` and ``outputs`` (both already normalised, like the training
        sets); the rows behind the last full batch are not looked at.  Fewer rows than one batch divide by zero in the
        reference (:196) and raise ZeroDivisionError here.  Returns a Python float (the reference returns the numpy scalar)."""
        def f(a, cols):
            if torch.is_tensor(a):
                return a.to(device=self.device, dtype=torch.float32).reshape(-1, cols)
            return torch.as_tensor(np.asarray(a, np.float32).reshape(-1, cols), device=self.device)
        X, Z = f(inputs, self.in_dim), f(outputs, self.out_dim)
        n_batches = X.shape[0] // int(batchsize)
        if n_batches == 0:
            raise ZeroDivisionError("run_validation: fewer rows than one batch (dynamics_model.py:196 divides by iters_in_batch = 0)")
        rows = n_batches * int(batchsize)
        pred = self.forward(X[:rows], precision=precision)
        out = torch.empty(n_batches + 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.ssc_mse_batches(_ffi.ptr(pred), _ffi.ptr(Z[:rows].contiguous()), n_batches, int(batchsize) * self.out_dim,
                                                _ffi.ptr(out), out.data_ptr() + 4 * n_batches, _stream()))
        return float(out[n_batches].item())

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        return self._ws

    def _mfma_image(self):
        """Workspace holding the packed weight image, (re)written only after the weights changed -- the
        weights stay resident between calls like TF variables across ``sess.run`` (dynamics_model.py:226-233)."""
        nbytes = self.lib.ssc_dyn_workspace_bytes(ctypes.byref(self.desc), 1, _ffi.SSC_PREC_BF16_MFMA)
        if self._image is None or self._image.numel() < nbytes:
            self._image = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._image_stale = True
        if self._image_stale:
            _ffi.check(self.lib.ssc_dyn_prepare(ctypes.byref(self.desc), ctypes.byref(self.norm),
                                                _ffi.ptr(self._image), self._image.numel(), _stream()))
            self._image_stale = False
        return self._image

    def refresh_prepared_image(self):
        """Re-pack the MFMA weight image if the weights changed since it was written (same buffer, so launches
        captured in a HIP graph keep reading the right bytes)."""
        if self._image is not None and self._image_stale and self._mfma_ok:
            self._mfma_image()

    def _resolve(self, precision):
        """SSC_PREC_* for a call.  A network the MFMA kernel does not cover (more than 2 hidden layers, depth >
        512, ...) runs on the fp32 GPU kernels instead when the model's DEFAULT precision asked for MFMA; an
        explicit ``precision="bf16_mfma"`` argument still raises SSC_EUNSUPPORTED."""
        prec = _PREC[precision or self.precision]
        if prec == _ffi.SSC_PREC_BF16_MFMA and precision is None:
            if self._mfma_ok is None:
                try:
                    self._mfma_image()
                    self._mfma_ok = True
                except _ffi.SscError as e:
                    if e.code != _ffi.SSC_EUNSUPPORTED:
                        raise
                    self._mfma_ok = False
            if not self._mfma_ok:
                return _ffi.SSC_PREC_F32
        return prec

    def forward(self, x, precision=None):
        """z = feedforward_network(x): x [m, in] -> [m, out]."""
        prec = self._resolve(precision)
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        m = x.shape[0]
        y = torch.empty((m, self.out_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            if prec == _ffi.SSC_PREC_BF16_MFMA:
                ws, prec = self._mfma_image(), _ffi.SSC_PREC_BF16_MFMA_PREPARED
            else:
                ws = self._workspace(self.lib.ssc_mlp_workspace_bytes(ctypes.byref(self.desc), m, prec))
            _ffi.check(self.lib.ssc_mlp_forward(ctypes.byref(self.desc), m, _ffi.ptr(x), _ffi.ptr(y), prec,
                                                _ffi.ptr(ws), ws.numel(), _stream()))
        return y

    def do_forward_sim_sampled(self, state0, sampling, m, H, precision=None, out=None, A_out=None):
        """``do_forward_sim`` with the candidate action sequences drawn INSIDE the kernel from ``sampling``
        (``mpc_sampling(...)``; bit-identical to ``mpc_sample_actions`` + ``do_forward_sim``): one launch instead of two
        and no [m, H, act] matrix unless ``A_out`` asks for it (the fp32 path needs it)."""
        prec = self._resolve(precision)
        s0 = torch.as_tensor(state0, dtype=torch.float32, device=self.device).contiguous()
        s0_rows = 1 if s0.dim() == 1 else s0.shape[0]
        S = out if out is not None else torch.empty((H + 1, m, self.state_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            if prec == _ffi.SSC_PREC_BF16_MFMA:
                ws, prec = self._mfma_image(), _ffi.SSC_PREC_BF16_MFMA_PREPARED
            else:
                ws = self._workspace(self.lib.ssc_dyn_workspace_bytes(ctypes.byref(self.desc), m, prec))
                if A_out is None:
                    A_out = torch.empty((m, H, self.act_dim), dtype=torch.float32, device=self.device)
            _ffi.check(self.lib.ssc_mpc_forward_sim(ctypes.byref(self.desc), ctypes.byref(self.norm), ctypes.byref(sampling),
                                                    m, H, self.state_dim, self.act_dim, _ffi.ptr(s0), s0_rows,
                                                    _ffi.ptr(A_out), _ffi.ptr(S), prec, _ffi.ptr(ws), ws.numel(), _stream()))
        return S

    def do_forward_sim(self, state0, actions, precision=None, out=None):
        """``Dyn_Model.do_forward_sim(..., many_in_parallel=True)``: state0 [d] (tiled, :215-217) or
        [m, d]; actions [m, H, act] -> states [H+1, m, d]."""
        prec = self._resolve(precision)
        A = torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
        m, H = A.shape[0], A.shape[1]
        s0 = torch.as_tensor(state0, dtype=torch.float32, device=self.device).contiguous()
        s0_rows = 1 if s0.dim() == 1 else s0.shape[0]
        S = out if out is not None else torch.empty((H + 1, m, self.state_dim), dtype=torch.float32,
                                                    device=self.device)
        with torch.cuda.device(self.device):
            if prec == _ffi.SSC_PREC_BF16_MFMA:
                ws, prec = self._mfma_image(), _ffi.SSC_PREC_BF16_MFMA_PREPARED
            else:
                ws = self._workspace(self.lib.ssc_dyn_workspace_bytes(ctypes.byref(self.desc), m, prec))
            _ffi.check(self.lib.ssc_dyn_forward_sim(ctypes.byref(self.desc), ctypes.byref(self.norm), m, H,
                                                    self.state_dim, self.act_dim, _ffi.ptr(s0), s0_rows,
                                                    _ffi.ptr(A), _ffi.ptr(S), prec, _ffi.ptr(ws), ws.numel(),
                                                    _stream()))
        return S


def mpc_sampling(N, low, high, seed, problem_id0=0, t=0, t_base=None, active=None, live_list=None, n_live=None):
    """``ssc_mpc_sampling``: the candidate action sequences of an MPC step as a specification (NND_MB_agent.py:500-501).
    ``active``: optional uint8 device tensor, one byte per problem -- rows of problems marked 0 need not be simulated.
    ``live_list`` / ``n_live``: the compact list of live problems and its device count (``ssc_nav_compact``): the fused
    small-network kernel then spends threads on those problems' rows only."""
    sp = _ffi.MpcSampling()
    low = np.asarray(low, np.float32).reshape(-1)
    high = np.asarray(high, np.float32).reshape(-1)
    sp.n_samples = int(N)
    for i in range(low.size):
        sp.low[i], sp.high[i] = float(low[i]), float(high[i])
    sp.seed, sp.problem_id0, sp.t = int(seed), int(problem_id0), int(t)
    sp.d_t_base = None if t_base is None else t_base.data_ptr()
    sp.d_problem_active = None if active is None else active.data_ptr()
    if live_list is not None and n_live is not None:
        sp.d_live_list, sp.d_n_live = live_list.data_ptr(), n_live.data_ptr()
    sp._keep = (t_base, active, live_list, n_live)
    return sp


class MpcProblemSet:
    """Waypoint data of P navigation problems, packed for the scorer
    (NND_MB_agent.start_new_episode_plan, NND_MB_agent.py:375-423)."""

    def __init__(self, waypoints, lefts, radii, cur_idx, device="cuda", theta=1.0, gamma=0.75,
                 horizontal_penalty_factor=0.5, per_row_projection=False):
        self.device = torch.device(device)
        P = len(waypoints)
        d = np.asarray(waypoints[0]).shape[1]
        off = np.zeros(P + 1, np.int32)
        for p, w in enumerate(waypoints):
            if len(w) < 2:
                raise ValueError("every problem needs at least 2 waypoints (the reference indexes wp[b+1])")
            off[p + 1] = off[p] + len(w)
        self.P, self.d = P, d
        self.wp = torch.as_tensor(np.concatenate([np.asarray(w, np.float32) for w in waypoints]), device=self.device)
        self.left = torch.as_tensor(np.concatenate([np.asarray(l, np.float32) for l in lefts]), device=self.device)
        self.wp_off = torch.as_tensor(off, device=self.device)
        self.cur_idx = torch.as_tensor(np.asarray(cur_idx, np.int32), device=self.device)
        self.radii = torch.as_tensor(np.asarray(radii, np.float32).reshape(P, d), device=self.device).contiguous()
        self.theta, self.gamma, self.hpf = float(theta), float(gamma), float(horizontal_penalty_factor)
        self.per_row = bool(per_row_projection)

    @classmethod
    def from_packed(cls, wp, left, wp_off, radii, cur_idx, device="cuda", theta=1.0, gamma=0.75,
                    horizontal_penalty_factor=0.5, per_row_projection=False):
        """The same problem set from ALREADY PACKED arrays (or device tensors): ``wp`` [sum W_p, d], ``left`` [sum W_p],
        ``wp_off`` [P + 1] (int32 prefix sums of the waypoint counts), ``radii`` [P, d], ``cur_idx`` [P] -- the form
        to use for tens of thousands of problems (one navigator per env), where a Python list per problem is the
        slow part."""
        self = cls.__new__(cls)
        self.device = torch.device(device)
        t = lambda x, dt: torch.as_tensor(x, dtype=dt).to(self.device).contiguous()
        self.wp, self.left = t(wp, torch.float32), t(left, torch.float32).reshape(-1)
        self.wp_off, self.cur_idx = t(wp_off, torch.int32).reshape(-1), t(cur_idx, torch.int32).reshape(-1)
        self.P, self.d = self.wp_off.numel() - 1, self.wp.shape[1]
        self.radii = t(radii, torch.float32).reshape(self.P, self.d)
        if self.cur_idx.numel() != self.P or self.left.numel() != self.wp.shape[0]:
            raise ValueError("packed problem set: inconsistent array sizes")
        self.theta, self.gamma, self.hpf = float(theta), float(gamma), float(horizontal_penalty_factor)
        self.per_row = bool(per_row_projection)
        return self

    def as_struct(self, n_samples, horizon):
        s = _ffi.MpcProblems()
        s.n_problems, s.n_samples, s.horizon, s.state_dim = self.P, n_samples, horizon, self.d
        s.wp, s.left, s.wp_off = self.wp.data_ptr(), self.left.data_ptr(), self.wp_off.data_ptr()
        s.cur_idx, s.radii = self.cur_idx.data_ptr(), self.radii.data_ptr()
        s.theta, s.gamma, s.horizontal_penalty_factor = self.theta, self.gamma, self.hpf
        s.per_row_projection = int(self.per_row)
        return s


class PlanPool:
    """The problem set of the vectorised SmartStart loop: ``n_envs`` envs, each following ONE OF A FEW stored plans.
    ``n_slots`` plan slots of up to ``w_max`` waypoints live in fixed device arrays (``wp``, ``left``, ``wp_len``,
    ``radii``); ``plan_of[p]`` names the slot env p follows and ``cur_idx[p]`` its waypoint.  New plans (one smart-start
    selection = ``n_plans`` of them) are written round-robin into the slots, so a plan stays intact for
    ``n_slots / n_plans`` refreshes -- size the pool so that this covers an episode.  ``pool`` (int32[3] on the device:
    first slot on offer, number on offer, slots) is what the step kernel draws a finished env's next plan from.
    Duck-types :class:`MpcProblemSet` for :class:`NavigatorBatch` and the scorer."""

    def __init__(self, n_envs, n_slots, w_max, d, device="cuda", theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5,
                 per_row_projection=False):
        self.device = torch.device(device)
        self.P, self.d, self.n_slots, self.w_max = int(n_envs), int(d), int(n_slots), int(w_max)
        dev = self.device
        self.wp = torch.zeros((self.n_slots * self.w_max, d), dtype=torch.float32, device=dev)
        self.left = torch.zeros(self.n_slots * self.w_max, dtype=torch.float32, device=dev)
        self.wp_off = (torch.arange(self.n_slots + 1, dtype=torch.int32) * self.w_max).to(dev)
        # every slot starts as a harmless two-waypoint plan (the scorer runs for every env, navigating or not)
        self.wp.view(self.n_slots, self.w_max, d)[:, 1, :] = 1.0
        self.left.view(self.n_slots, self.w_max)[:, 0] = float(np.sqrt(d))
        self.wp_len = torch.full((self.n_slots,), 2, dtype=torch.int32, device=dev)
        self.radii = torch.ones((self.n_slots, d), dtype=torch.float32, device=dev)
        self.plan_of = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.cur_idx = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.pool = torch.tensor([0, 0, self.n_slots], dtype=torch.int32, device=dev)
        self.theta, self.gamma, self.hpf = float(theta), float(gamma), float(horizontal_penalty_factor)
        self.per_row = bool(per_row_projection)
        self._next = 0
        self.published = 0
        self.active = None      # optional uint8 [n_envs]: only problems marked non-zero are simulated / scored

    def publish(self, plans, now=None, min_age=None):
        """``plans``: list of (waypoints [W, d], distances_left [W], radii [d]) -- the per-episode quantities of
        NND_MB_agent.start_new_episode_plan (NND_MB_agent.py:375-423).  They go into the next slots and become the
        plans on offer; the slots they replace were offered ``n_slots / len(plans)`` refreshes ago.
        ``now`` (env-steps so far) and ``min_age`` (the longest episode): a slot that went off offer fewer than ``min_age``
        steps ago may still be followed by an env -- overwriting it is reported once (RuntimeWarning; the kernels clamp
        the waypoint index, so such an env heads for the new plan's goal instead of reading outside it)."""
        if not plans:
            return
        if len(plans) > self.n_slots:
            raise ValueError("more plans than slots")
        first = self._next
        if now is not None:
            ages = self.__dict__.setdefault("_off_offer_at", {})
            if min_age is not None and not getattr(self, "_warned", False):
                young = [q for q in ((first + j) % self.n_slots for j in range(len(plans))) if q in ages and now - ages[q] < min_age]
                if young:
                    import warnings
                    warnings.warn("PlanPool: slot %d is re-published %d env-steps after it went off offer (< %d, the longest "
                                  "episode): the pool (%d slots) is too small for this refresh cadence" %
                                  (young[0], now - ages[young[0]], min_age, self.n_slots), RuntimeWarning, stacklevel=2)
                    self._warned = True
            for q in getattr(self, "_on_offer", []):       # the plans on offer until now go off offer
                ages[q] = now
            self._on_offer = [(first + j) % self.n_slots for j in range(len(plans))]
        # all plans of this refresh travel together: four host -> device copies and four scatters by slot, not eight small
        # operations per plan (they sit on the rollout's stream between two chunks)
        n = len(plans)
        wp_h = np.zeros((n, self.w_max, self.d), np.float32)
        left_h = np.zeros((n, self.w_max), np.float32)
        len_h = np.zeros(n, np.int32)
        radii_h = np.zeros((n, self.d), np.float32)
        for j, (w, l, r) in enumerate(plans):
            w = np.asarray(w, np.float32)
            l = np.asarray(l, np.float32)
            if len(w) < 2:                      # the reference indexes desired_states[b + 1]
                w, l = np.concatenate([w, w], axis=0), np.zeros(2, np.float32)
            if len(w) > self.w_max:
                raise ValueError("plan of %d waypoints > w_max %d" % (len(w), self.w_max))
            wp_h[j, :len(w)], left_h[j, :len(w)], len_h[j], radii_h[j] = w, l, len(w), np.asarray(r, np.float32)
            # (rows behind a plan's length keep whatever the slot held before in the reference-free sense: nothing reads
            # them -- every index is clamped to wp_len; they are zero here)
        slots = torch.as_tensor([(first + j) % self.n_slots for j in range(n)], dtype=torch.int64, device=self.device)
        dev = lambda a: torch.as_tensor(a, device=self.device)
        self.wp.view(self.n_slots, self.w_max, self.d).index_copy_(0, slots, dev(wp_h))
        self.left.view(self.n_slots, self.w_max).index_copy_(0, slots, dev(left_h))
        self.wp_len.index_copy_(0, slots, dev(len_h))
        self.radii.index_copy_(0, slots, dev(radii_h))
        self.pool.copy_(torch.tensor([first, len(plans), self.n_slots], dtype=torch.int32))
        self._next = (first + len(plans)) % self.n_slots
        self.published += len(plans)

    def as_struct(self, n_samples, horizon):
        s = _ffi.MpcProblems()
        s.n_problems, s.n_samples, s.horizon, s.state_dim = self.P, n_samples, horizon, self.d
        s.wp, s.left, s.wp_off = self.wp.data_ptr(), self.left.data_ptr(), self.wp_off.data_ptr()
        s.cur_idx, s.radii = self.cur_idx.data_ptr(), self.radii.data_ptr()
        s.theta, s.gamma, s.horizontal_penalty_factor = self.theta, self.gamma, self.hpf
        s.per_row_projection = int(self.per_row)
        s.plan_of, s.wp_len = self.plan_of.data_ptr(), self.wp_len.data_ptr()
        s.active = None if self.active is None else self.active.data_ptr()
        if getattr(self, "live_list", None) is not None:       # compact work list of the live problems (ssc_nav_compact)
            s.live_list, s.n_live = self.live_list.data_ptr(), self.n_live.data_ptr()
        return s


def mpc_sample_actions(P, N, H, low, high, seed, problem_id0=0, t=0, device="cuda", out=None, t_base=None):
    """``npr.uniform(low, high, (N, H, act))`` per problem (NND_MB_agent.py:500-501) -> [P*N, H, act].
    ``t_base``: a one-element int64 device tensor added to ``t`` on the device (HIP-graph replay)."""
    low = np.asarray(low, np.float32).reshape(-1)
    high = np.asarray(high, np.float32).reshape(-1)
    act = low.size
    A = out if out is not None else torch.empty((P * N, H, act), dtype=torch.float32, device=device)
    lo = (ctypes.c_float * act)(*low.tolist())
    hi = (ctypes.c_float * act)(*high.tolist())
    with torch.cuda.device(A.device):
        _ffi.check(_ffi.lib().ssc_mpc_sample_actions(P, N, H, act, lo, hi, int(seed), int(problem_id0), int(t),
                                                     _ffi.ptr(t_base), _ffi.ptr(A), _stream()))
    return A


def mpc_score(problems, S):
    """``generate_scores_add_delta`` + argmax (NND_MB_agent.py:566-628) for P problems.
    S: [H+1, P*N, d].  Returns (scores [P, N], best_idx [P] int32, best_score [P])."""
    S = S.contiguous()
    H1, M, d = S.shape
    P = problems.P
    N = M // P
    if N * P != M or d != problems.d:
        raise ValueError("S has the wrong shape for this problem set")
    lib = _ffi.lib()
    scores = torch.empty(M, dtype=torch.float32, device=S.device)
    best_idx = torch.empty(P, dtype=torch.int32, device=S.device)
    best_score = torch.empty(P, dtype=torch.float32, device=S.device)
    st = problems.as_struct(N, H1 - 1)
    with torch.cuda.device(S.device):
        nbytes = lib.ssc_mpc_score_workspace_bytes(P, N, H1 - 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=S.device)
        _ffi.check(lib.ssc_mpc_score(ctypes.byref(st), _ffi.ptr(S), _ffi.ptr(scores), _ffi.ptr(best_idx),
                                     _ffi.ptr(best_score), _ffi.ptr(ws), nbytes, _stream()))
    return scores.view(P, N), best_idx, best_score


def mpc_score_select(problems, S, sampling=None, A=None, act_dim=1, noise_amount=0.005, seed=0, problem_id0=0, t=0,
                     want_path=True, out=None):
    """``mpc_score`` + ``mpc_select_action`` in the same two launches (``ssc_mpc_score_select``): the block that finishes a
    problem's argmax also writes its action and predicted path.  The winner's first action comes from ``A`` or is
    regenerated from ``sampling`` (give exactly one).  Returns (scores [P, N], best_idx [P], action [P, act], path or None).
    ``out``: dict of preallocated scores / best / best_score / action / path / ws tensors (HIP-graph capture)."""
    S = S.contiguous()
    H1, M, d = S.shape
    P = problems.P
    N = M // P
    if N * P != M or d != problems.d:
        raise ValueError("S has the wrong shape for this problem set")
    lib = _ffi.lib()
    o = out if out is not None else {}
    dev = S.device
    scores = o.get("scores") if o.get("scores") is not None else torch.empty(M, dtype=torch.float32, device=dev)
    best_idx = o.get("best") if o.get("best") is not None else torch.empty(P, dtype=torch.int32, device=dev)
    best_score = o.get("best_score") if o.get("best_score") is not None else torch.empty(P, dtype=torch.float32, device=dev)
    action = o.get("action") if o.get("action") is not None else torch.empty((P, act_dim), dtype=torch.float32, device=dev)
    path = o.get("path") if o.get("path") is not None else \
        (torch.empty((P, H1, d), dtype=torch.float32, device=dev) if want_path else None)
    st = problems.as_struct(N, H1 - 1)
    with torch.cuda.device(dev):
        nbytes = lib.ssc_mpc_score_workspace_bytes(P, N, H1 - 1)
        ws = o.get("ws") if o.get("ws") is not None else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ffi.check(lib.ssc_mpc_score_select(ctypes.byref(st), _ffi.ptr(S), _ffi.ptr(scores), _ffi.ptr(best_idx),
                                            _ffi.ptr(best_score), _ffi.ptr(A), ctypes.byref(sampling) if sampling is not None else None,
                                            int(act_dim), float(noise_amount), int(seed), int(problem_id0), int(t),
                                            _ffi.ptr(action), _ffi.ptr(path), _ffi.ptr(ws), ws.numel(), _stream()))
    return scores.view(P, N), best_idx, action, path


def mpc_select_action(A, S, best_idx, P, noise_amount, seed, problem_id0=0, t=0, want_path=True):
    """``get_action_with_predicted_states`` tail (NND_MB_agent.py:339-358): first action of the best
    sequence + ``noise_amount * N(0,1)`` (no clip) and the predicted path [P, H+1, d]."""
    M, H, act = A.shape
    N = M // P
    d = S.shape[2]
    action = torch.empty((P, act), dtype=torch.float32, device=A.device)
    path = torch.empty((P, H + 1, d), dtype=torch.float32, device=A.device) if want_path else None
    with torch.cuda.device(A.device):
        _ffi.check(_ffi.lib().ssc_mpc_select_action(P, N, H, d, act, _ffi.ptr(A), _ffi.ptr(S), _ffi.ptr(best_idx),
                                                    float(noise_amount), int(seed), int(problem_id0), int(t),
                                                    _ffi.ptr(action), _ffi.ptr(path), _stream()))
    return action, path


class NavigatorBatch:
    """P navigators (one per env) sharing one dynamics model: the vectorised counterpart of
    ``NND_MB_agent.get_action`` / ``observe`` (NND_MB_agent.py:339-373, 498-520) -- per step ONE sample /
    forward-sim / score / select pipeline over all P x N candidate sequences, and one waypoint-advance
    kernel.  Plans (waypoints, distances_left, radii) come from ``NND_MB_agent.start_new_episode_plan`` or
    any code that fills :class:`MpcProblemSet`."""

    def __init__(self, dyn_model, problems, num_control_samples=5000, horizon=4, action_low=(-1.0,),
                 action_high=(1.0,), noise_amount=0.005, steps_before_giving_up_on_waypoint=5, final_steps=10,
                 seed=1234, problem_id0=0):
        self.model, self.problems = dyn_model, problems
        self.P, self.N, self.H = problems.P, int(num_control_samples), int(horizon)
        self.low, self.high = np.asarray(action_low, np.float32), np.asarray(action_high, np.float32)
        self.noise_amount, self.give_up, self.final_steps = noise_amount, steps_before_giving_up_on_waypoint, final_steps
        self.seed, self.problem_id0 = int(seed), int(problem_id0)
        dev = problems.wp.device
        self.actions_done = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.at_goal = torch.zeros(self.P, dtype=torch.uint8, device=dev)
        self.start_idx = problems.cur_idx.clone()
        self._S = torch.empty((self.H + 1, self.P * self.N, problems.d), dtype=torch.float32, device=dev)

    def get_action(self, states, t):
        """states [P, d] (device) -> (action [P, act], best_idx [P]) for global step ``t``."""
        self.actions_done += 1                                                            # :340
        # three launches: forward simulation (drawing its own candidate sequences), scoring pass A, scoring pass B +
        # selection; every problem's state is the start of its N candidate rows (np.tile, :215-217): s0_rows = P
        sp = mpc_sampling(self.N, self.low, self.high, self.seed, self.problem_id0, t)
        S = self.model.do_forward_sim_sampled(states.float().contiguous(), sp, self.P * self.N, self.H, out=self._S)
        _, best, action, _ = mpc_score_select(self.problems, S, sampling=sp, act_dim=len(self.low),
                                              noise_amount=self.noise_amount, seed=self.seed, problem_id0=self.problem_id0,
                                              t=t, want_path=False)
        return action, best

    def observe(self, new_states):
        """Waypoint bookkeeping after env.step (NND_MB_agent.observe :360-373; goal test :425-432)."""
        st = self.problems.as_struct(self.N, self.H)
        ns = new_states.float().contiguous()
        with torch.cuda.device(ns.device):
            _ffi.check(_ffi.lib().ssc_mpc_observe(ctypes.byref(st), _ffi.ptr(ns), _ffi.ptr(self.problems.cur_idx),
                                                  _ffi.ptr(self.actions_done), self.give_up, self.final_steps,
                                                  _ffi.ptr(self.at_goal), _stream()))

    def restart(self, mask):
        """Envs that were reset start their plan over (start_new_episode_plan :383-384)."""
        m = mask.bool()
        self.problems.cur_idx[m] = self.start_idx[m]
        self.actions_done[m] = 0

    # ---- fused, HIP-graph replayable step (VecEnv.rollout(K, MpcPolicy)) ------------------------------------
    def _fused_buffers(self, device):
        if getattr(self, "_fb", None) is None:
            M, d = self.P * self.N, self.problems.d
            lib = _ffi.lib()
            nbytes = lib.ssc_mpc_score_workspace_bytes(self.P, self.N, self.H)
            self._fb = dict(
                A=torch.empty((M, self.H, len(self.low)), dtype=torch.float32, device=device),
                scores=torch.empty(M, dtype=torch.float32, device=device),
                best=torch.zeros(self.P, dtype=torch.int32, device=device),
                best_score=torch.empty(self.P, dtype=torch.float32, device=device),
                ws=torch.empty(nbytes, dtype=torch.uint8, device=device),
                plan=torch.empty((self.P, d), dtype=torch.float32, device=device),
                t=torch.zeros(1, dtype=torch.int64, device=device),      # global step counter (uint64 on the device)
                k=torch.zeros(1, dtype=torch.int32, device=device),      # log row inside the chunk
                ticket=torch.zeros(1, dtype=torch.int32, device=device))
        return self._fb

    def fused_step(self, env, chunk, ring):
        """Enqueue ONE MPC-policy step for every env of ``env`` (a VecEnv with one env per problem): sample ->
        forward sim from the planning states -> score -> ``ssc_mpc_rollout_step``.  Step index and log row come
        from device counters, so the sequence is captured once as a HIP graph and replayed."""
        fb = self._fused_buffers(env.device)
        lib = _ffi.lib()
        if len(self.low) != 1:
            raise ValueError("the envs of this engine take one action component")
        # (the forward simulation draws the candidate sequences itself and leaves them in fb["A"] for the step kernel)
        sp = mpc_sampling(self.N, self.low, self.high, self.seed, self.problem_id0, 0, t_base=fb["t"])
        S = self.model.do_forward_sim_sampled(fb["plan"], sp, self.P * self.N, self.H, out=self._S, A_out=fb["A"])
        st = self.problems.as_struct(self.N, self.H)
        nav = _ffi.MpcNavState(self.problems.cur_idx.data_ptr(), self.start_idx.data_ptr(), self.actions_done.data_ptr(),
                               self.at_goal.data_ptr(), self.give_up, self.final_steps)
        rs = _ffi.RolloutState(env.s0.data_ptr(), env.s1.data_ptr(), env.steps.data_ptr(), env.ep_ret.data_ptr(),
                               env.ou_x.data_ptr())
        log_s = chunk.as_struct() if chunk is not None else None
        ring_s = ring.as_struct() if ring is not None else None
        with torch.cuda.device(env.device):
            _ffi.check(lib.ssc_mpc_score(ctypes.byref(st), _ffi.ptr(S), _ffi.ptr(fb["scores"]), _ffi.ptr(fb["best"]),
                                         _ffi.ptr(fb["best_score"]), _ffi.ptr(fb["ws"]), fb["ws"].numel(), _stream()))
            _ffi.check(lib.ssc_mpc_rollout_step(
                ctypes.byref(env.params), ctypes.byref(st), ctypes.byref(nav), _ffi.ptr(fb["A"]), _ffi.ptr(fb["best"]),
                float(self.noise_amount), self.seed, self.problem_id0, ctypes.byref(rs),
                ctypes.byref(log_s) if log_s is not None else None,
                ctypes.byref(ring_s) if ring_s is not None else None,
                _ffi.ptr(env.stats), env._seed, env.env_id0, _ffi.ptr(fb["t"]), _ffi.ptr(fb["k"]), _ffi.ptr(fb["ticket"]),
                _ffi.ptr(fb["plan"]), _stream()))
