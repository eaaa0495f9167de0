#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Run ONCE in the build container (``python tests/golden/make_goldens.py``); the produced
``*.npz`` files are committed.  /root/reference does not exist on the GPU box, so nothing
in tests/, smoke() or bench.py reads it at run time -- they read these fixtures.

What is extracted:
  1. mc_reference_rollouts.npz -- the reference's own recorded fp64 random-policy rollouts of
     stock MountainCarContinuous-v0 (models/NND_MB_agent/default/training_data/*.npy).
     These are DATA files the reference ships, i.e. outputs of the reference's env.step.
  2. mc_summary_paths.npz -- best/last paths + stored returns + episode records from a few of
     the reference's experiment summaries (data/**/*.json): pins reward, done>=0.45, action
     clipping and the TimeLimit lengths (999 stock / 1000 edited env, power_scalar 1 and 0.4).
  3. numerical_kats.npz -- known-answer vectors produced by IMPORTING the reference's
     smartstart/utilities/numerical.py by file path (it only needs numpy/scipy) and calling
     its functions on seeded inputs, including the batched-projection quirk and an MPC
     scoring loop assembled from the reference's own helper functions.

  4. replay_buffer_kats.npz / reference_summary.json -- traces of the reference's ReplayBuffer and one of its
     Summary dumps.
  5. data_manipulation_kats.npz -- ragged rollout lists pushed through the reference's
     generate_training_data_inputs / generate_training_data_outputs
     (smartstart/RLContinuousAlgorithms/NN_Dynamics_Model/data_manipulation.py:58-88, imported by path).

No reference source text is copied; only numeric inputs/outputs are stored.
"""
import glob
import importlib.util
import json
import os
import sys

import numpy as np

REF = os.environ.get("SSC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, rel):
    np.product = np.prod  # numerical.py:164 uses the numpy<2 alias
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rollouts():
    d = os.path.join(REF, "models/NND_MB_agent/default/training_data")
    out = {k: np.load(os.path.join(d, k + ".npy")) for k in
           ["states_val", "controls_val", "dataX", "dataY", "dataZ", "forwardsim_x_true", "forwardsim_y"]}
    np.savez_compressed(os.path.join(OUT, "mc_reference_rollouts.npz"), **out)
    print("rollouts:", {k: v.shape for k, v in out.items()})


def summaries():
    picks = sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params/*_0.json")))[:1]
    picks += sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params/*_17.json")))[:1]
    picks += sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params/*_101.json")))[:1]
    picks += sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params_cont_mc_editted/*_0.json")))[:1]
    picks += sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params_cont_mc_editted/*_2.json")))[:1]
    picks += sorted(glob.glob(os.path.join(
        REF, "data/smart_start_continuous_summaries/ddpg_baselines/hyper_parameter_search/*.json")))[:2]
    out = {}
    meta = []
    for i, f in enumerate(picks):
        d = json.load(open(f))
        power_scalar = 0.4 if "ActionX0.4" in d["name"] else 1.0
        max_steps = 1000 if "ActionX" in d["name"] else 999
        out[f"f{i}_best_path"] = np.asarray(d["best_path"], np.float64)
        out[f"f{i}_best_reward"] = np.float64(d["best_reward"])
        for j, (p, r) in enumerate(zip(d["last_paths"], d["last_rewards"])):
            out[f"f{i}_last_path{j}"] = np.asarray(p, np.float64)
            out[f"f{i}_last_reward{j}"] = np.float64(r)
        out[f"f{i}_episodes"] = np.asarray(d["episodes"], np.float64)
        out[f"f{i}_power_scalar"] = np.float64(power_scalar)
        out[f"f{i}_max_steps"] = np.int64(max_steps)
        meta.append(os.path.relpath(f, REF))
    out["n_files"] = np.int64(len(picks))
    out["sources"] = np.asarray(meta)
    np.savez_compressed(os.path.join(OUT, "mc_summary_paths.npz"), **out)
    print("summaries:", meta)


def numerical_kats():
    num = load_by_path("ref_numerical", "smartstart/utilities/numerical.py")
    rng = np.random.default_rng(1234)
    out = {}
    # --- distance / projection / segment distance, single-row and batched (quirk) ---
    for case, (n, d) in enumerate([(1, 2), (7, 2), (64, 2), (33, 3), (500, 2)]):
        radii = rng.uniform(0.01, 2.0, size=d)
        a = rng.normal(size=(n, d))
        b = rng.normal(size=(n, d))
        pt = rng.normal(size=(n, d))
        dist = num.elliptical_euclidean_distance_function_generator(radii)
        out[f"g{case}_radii"] = radii
        out[f"g{case}_a"] = a
        out[f"g{case}_b"] = b
        out[f"g{case}_pt"] = pt
        out[f"g{case}_dist_ab"] = dist(a, b)
        out[f"g{case}_proj"] = num.projection_of_a_onto_b(a, b)
        out[f"g{case}_proj_radii"] = num.projection_of_a_onto_b(a, b, radii=radii)
        out[f"g{case}_segdist"] = num.dist_line_seg_to_point(a, b, pt, dist, radii)
        # row-by-row calls (what a non-batched caller would get)
        out[f"g{case}_segdist_rowwise"] = np.asarray(
            [num.dist_line_seg_to_point(a[i], b[i], pt[i], dist, radii) for i in range(n)])
    out["n_geom"] = np.int64(5)
    # --- path statistics, radii, shortcutter ---
    n_paths = 6
    for case in range(n_paths):
        L = int(rng.integers(5, 120))
        d = 2 if case % 2 == 0 else 3
        steps = rng.normal(scale=[0.02, 0.004, 0.01][:d], size=(L, d))
        path = np.cumsum(steps, axis=0)
        if case >= 3:  # make the walk revisit itself so that shortcuts exist
            path = np.concatenate([path, path[::-1][: L // 2] + rng.normal(scale=1e-3, size=(L // 2, d))])
        stds, means = num.path_deltas_stds_and_means_per_dim(path)
        radii = num.radii_calc(means, stds, 1, 1, 1)
        dist = num.elliptical_euclidean_distance_function_generator(radii)
        short = num.path_shortcutter(path, dist, 1)
        out[f"p{case}_path"] = path
        out[f"p{case}_stds"] = stds
        out[f"p{case}_means"] = means
        out[f"p{case}_radii"] = radii
        out[f"p{case}_short"] = np.asarray(short)
    out["n_paths"] = np.int64(n_paths)
    # activity solver on random interval sets
    n_act = 12
    for case in range(n_act):
        m = int(rng.integers(1, 40))
        starts = rng.integers(0, 50, size=m)
        lens = rng.integers(2, 20, size=m)
        acts = np.stack([starts, starts + lens], axis=1)
        w, chosen = num.length_weighted_activities_solver(acts.tolist(), sub_extra=1)
        out[f"act{case}_in"] = acts
        out[f"act{case}_w"] = np.int64(w)
        out[f"act{case}_chosen"] = np.asarray(chosen, np.int64).reshape(-1, 2)
    out["n_act"] = np.int64(n_act)
    # hyperellipsoid volume
    vr = rng.uniform(0.1, 3.0, size=(8, 3))
    out["vol_radii"] = vr
    out["vol_2d"] = np.asarray([num.volume_of_n_dimensional_hyperellipsoid(list(r[:2])) for r in vr])
    out["vol_3d"] = np.asarray([num.volume_of_n_dimensional_hyperellipsoid(list(r)) for r in vr])

    # --- MPC scoring assembled from the reference's helpers -------------------------------
    # NND_MB_agent.py itself needs TensorFlow and cannot be imported; the loop below follows
    # NND_MB_agent.py:566-628 and calls the REFERENCE's distance / dist_line_seg_to_point /
    # projection functions, so the batch-global quirk in the expected scores is the reference's.
    n_mpc = 4
    for case, (N, H, W) in enumerate([(50, 4, 12), (500, 4, 40), (257, 20, 7), (64, 3, 2)]):
        d = 2
        wp = np.cumsum(rng.normal(scale=[0.02, 0.004], size=(W, d)), axis=0) + [-0.5, 0.0]
        stds, means = num.path_deltas_stds_and_means_per_dim(wp)
        radii = num.radii_calc(means, stds, 1, 1, 1)
        dist = num.elliptical_euclidean_distance_function_generator(radii)
        td = [dist(wp[x - 1], wp[x]) for x in range(1, W)] + [0]
        left = np.asarray([sum(td[i:]) for i in range(W)])
        cur = int(rng.integers(0, W))
        start = wp[cur] + rng.normal(scale=radii * 0.7)
        S = [np.tile(start, (N, 1))]
        for _ in range(H):
            S.append(S[-1] + rng.normal(scale=radii * 0.9, size=(N, d)))
        S = np.asarray(S)
        theta, gamma, hpf = 1, 0.75, 0.5
        scores = np.zeros((N,))
        idx = np.tile(cur, (N,)).astype(int)
        prev = left[idx] + dist(S[0], wp[idx])
        for t in range(S.shape[0]):
            pts = S[t]
            dc = dist(wp[idx], pts)
            dn = dist(wp[np.minimum(idx + 1, W - 1)], pts)
            move = np.logical_and(np.logical_or(dc <= theta, dn <= dc), idx != W - 1)
            idx[move] += 1
            dc[move] = dn[move]
            end = left[idx] + dc
            scores += (prev - end) * (gamma ** t)
            np.copyto(prev, end)
            b = np.maximum(idx - 1, 0)
            dd = num.dist_line_seg_to_point(wp[b], wp[b + 1], pts, dist, radii)
            scores -= dd * hpf * gamma
        out[f"m{case}_S"] = S
        out[f"m{case}_wp"] = wp
        out[f"m{case}_left"] = left
        out[f"m{case}_radii"] = radii
        out[f"m{case}_cur"] = np.int64(cur)
        out[f"m{case}_scores"] = scores
        out[f"m{case}_final_idx"] = idx
        out[f"m{case}_best"] = np.int64(np.argmax(scores))
    out["n_mpc"] = np.int64(n_mpc)
    np.savez_compressed(os.path.join(OUT, "numerical_kats.npz"), **out)
    print("numerical kats:", len(out), "arrays")


def replay_buffer_kats():
    rb = load_by_path("ref_replay_buffer", "smartstart/RLAgents/replay_buffer.py")
    rng = np.random.default_rng(99)
    out = {}
    agent = object()
    buf = rb.ReplayBuffer(agent, 50)
    trace_len, trace_next, trace_starts = [], [], []
    k = 0
    ep_lens = rng.integers(3, 30, size=12)
    for L in ep_lens:
        buf.start_new_episode(agent)
        for _ in range(int(L)):
            buf.add(agent, np.array([k, 0.0]), np.array([0.0]), 0.0, False, np.array([k + 1, 0.0]))
            k += 1
            trace_len.append(len(buf.buffer))
            trace_next.append(buf.next_episode_number)
            trace_starts.append(list(buf.episode_starting_indices) + [-1] * (16 - len(buf.episode_starting_indices)))
    out["ep_lens"] = ep_lens
    out["trace_len"] = np.asarray(trace_len)
    out["trace_next"] = np.asarray(trace_next)
    out["trace_starts"] = np.asarray(trace_starts)
    first = buf.episode_number_to_buffer_index(buf.episode_starting_indices[0])
    out["final_first_index"] = np.int64(first)
    paths = []
    for bi in [first, first + 3, len(buf.buffer) - 1]:
        p = np.asarray(buf.get_episodic_path_to_buffer_index(int(bi)))
        out[f"path_to_{len(paths)}"] = p
        out[f"path_idx_{len(paths)}"] = np.int64(bi)
        paths.append(p)
    out["all_states"] = buf.get_all_states()
    np.savez_compressed(os.path.join(OUT, "replay_buffer_kats.npz"), **out)
    print("replay kats ok")


def replay_buffer_clear_kats():
    """clear() and a direct assignment of next_episode_number on the reference's ReplayBuffer (replay_buffer.py:105-107,
    :131): the VALUES in episode_starting_indices stay, only the counter moves.  Trace of (len, next_episode_number,
    episode_starting_indices) after every operation of a fixed script."""
    rb = load_by_path("ref_replay_buffer", "smartstart/RLAgents/replay_buffer.py")
    agent = object()
    buf = rb.ReplayBuffer(agent, 40)
    ops, trace = [], []
    k = 0

    def snap(op):
        ops.append(op)
        trace.append([len(buf.buffer), buf.next_episode_number] + list(buf.episode_starting_indices) +
                     [-99] * (16 - len(buf.episode_starting_indices)))
    script = [("ep", 7), ("ep", 5), ("clear", 0), ("ep", 4), ("ep", 9), ("set", 3), ("ep", 6), ("ep", 30), ("clear", 0), ("ep", 3)]
    for kind, arg in script:
        if kind == "ep":
            buf.start_new_episode(agent)
            snap(0)
            for _ in range(arg):
                buf.add(agent, np.array([k, 0.0]), np.array([0.0]), 0.0, False, np.array([k + 1, 0.0]))
                k += 1
                snap(1)
        elif kind == "clear":
            buf.clear()
            snap(2)
        else:
            buf.next_episode_number = arg
            snap(3)
    np.savez_compressed(os.path.join(OUT, "replay_buffer_clear_kats.npz"), script_kind=np.array([["ep", "clear", "set"].index(a) for a, _ in script]),
                        script_arg=np.array([b for _, b in script]), ops=np.asarray(ops), trace=np.asarray(trace, np.int64))
    print("replay clear kats ok:", len(ops), "operations")


def summary_json_fixture():
    """One of the reference's own Summary dumps (data/**/*.json, written by Summary.save,
    smartstart/utilities/datacontainers.py:288-326), trimmed to its first 60 episode records so that the
    fixture stays small.  Pins the on-disk schema for Summary.load / to_json."""
    f = sorted(glob.glob(os.path.join(
        REF, "data/smart_start_continuous_summaries/ddpg_baselines/hyper_parameter_search/*.json")))[0]
    d = json.load(open(f))
    d["episodes"] = d["episodes"][:60]
    d["smart_start_episodes"] = [e for e in d["smart_start_episodes"] if e < 60]
    with open(os.path.join(OUT, "reference_summary.json"), "w") as fh:
        json.dump(d, fh)
    print("summary fixture:", os.path.relpath(f, REF), sorted(d.keys()))


def data_manipulation_kats():
    """Rollout lists -> (dataX, dataY, dataZ) through the reference's own functions.  Case 0: seeded ragged
    rollouts (lengths 1 and 2 included: they contribute 0 and 1 rows); case 1: the reference's recorded
    validation rollouts, truncated to ragged lengths."""
    dm = load_by_path("ref_data_manipulation",
                      "smartstart/RLContinuousAlgorithms/NN_Dynamics_Model/data_manipulation.py")
    rng = np.random.default_rng(4242)
    out = {}
    lens0 = [5, 1, 2, 64, 65, 129, 333, 3]
    states0 = [rng.normal(size=(L, 3)) for L in lens0]
    controls0 = [rng.uniform(-2, 2, size=(L, 1)) for L in lens0]
    d = os.path.join(REF, "models/NND_MB_agent/default/training_data")
    sv, cv = np.load(os.path.join(d, "states_val.npy")), np.load(os.path.join(d, "controls_val.npy"))
    lens1 = [int(x) for x in rng.integers(2, sv.shape[1] + 1, size=sv.shape[0])]
    states1 = [sv[i, :L] for i, L in enumerate(lens1)]
    controls1 = [cv[i, :L] for i, L in enumerate(lens1)]
    for tag, lens, st, ct in ((0, lens0, states0, controls0), (1, lens1, states1, controls1)):
        # the reference np.copy()s the ragged list, which numpy 1.15 (its pin) turns into an object array and
        # numpy 2 refuses -- hand it that object array directly
        ragged = lambda lst: np.array(lst + [None], dtype=object)[:-1]
        X, Y = dm.generate_training_data_inputs(ragged(st), ragged(ct))
        Z = dm.generate_training_data_outputs(st)
        out[f"c{tag}_lens"] = np.asarray(lens, np.int64)
        out[f"c{tag}_states"] = np.concatenate(st, axis=0)
        out[f"c{tag}_controls"] = np.concatenate(ct, axis=0)
        out[f"c{tag}_dataX"], out[f"c{tag}_dataY"], out[f"c{tag}_dataZ"] = X, Y, Z
    np.savez_compressed(os.path.join(OUT, "data_manipulation_kats.npz"), **out)
    print("data_manipulation kats:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not found at %s" % REF)
    jobs = dict(rollouts=rollouts, summaries=summaries, numerical_kats=numerical_kats,
                replay_buffer_kats=replay_buffer_kats, summary_json_fixture=summary_json_fixture,
                data_manipulation_kats=data_manipulation_kats)
    for name in (sys.argv[1:] or list(jobs)):       # `make_goldens.py data_manipulation_kats` regenerates one file
        jobs[name]()
