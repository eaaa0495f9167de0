"""GPU tests of the vectorised SmartStart loop (smartexplorationcontinuous.py:307-376 for N envs at once):
``ssc_smartstart_rollout_step`` / ``VecSmartStart`` / ``rl_train_vec_smartstart``.  Every logged step is re-derived by the
oracle from the state the env was in -- who acted (navigator or base agent), the action, the env transition, the waypoint
bookkeeping, the hand-over, and what a finished episode starts next."""
import numpy as np
import pytest

from oracle import ssc_oracle as O
from tests.gpu_util import actor_weights
from tests.test_gpu_navigator import make_mlp, make_norm

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def _setup(ssc, n, max_steps, eta, seed, N=96, H=3, give_up=2, log_modes=True, n_plans=3, chunk=16, env_id0=40,
           env_name="MountainCarContinuous-v0"):
    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    rng = np.random.default_rng(seed)
    d = 2 if env_name.startswith("Mountain") else 3
    env = ssc.VecEnv(env_name, n, seed=seed, max_episode_steps=max_steps, env_id0=env_id0)
    env.reset()
    agent = DDPG_Baselines_agent(ssc.make(env_name), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                                 lastLayerTanh=True, seed=5, training=False, ou_mu=0.4, ou_sigma=0.6, precision="f32")
    w = actor_weights(d, 64, 32, seed=77, w3_scale=0.5)
    agent.set_weights({k: torch.as_tensor(v) for k, v in w.items()})
    Ws, bs = make_mlp(rng, (d + 1, 32, d))
    norm = make_norm(rng, d, 1)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision="f32")
    smart = ssc.VecSmartStart(env, agent, model, eta=eta, n_plans=n_plans, num_control_samples=N, horizon=H,
                              steps_before_giving_up_on_waypoint=give_up, final_steps=4, chunk_steps=chunk, seed=seed + 1,
                              log_modes=log_modes, w_max=max(max_steps + 1, 16))
    return env, agent, w, (Ws, bs, norm), smart


def _plans(rng, n, near_reset):
    """Hand-made plans: random walks that start in the reset region; ``near_reset`` ones also END there, so that
    close_enough_to_goal(reset state) is true for some envs and they never navigate."""
    out = []
    for j in range(n):
        L = int(rng.integers(6, 14))
        path = np.stack([np.cumsum(rng.normal(scale=0.02, size=L)) - 0.5, np.cumsum(rng.normal(scale=0.004, size=L))], 1)
        if j < near_reset:
            path[-1] = [-0.5, 0.0]
        stds, means = O.path_deltas_stds_and_means_per_dim(path)
        r = O.radii_calc(means, stds, 1, 1, 1) + 1e-3
        if j < near_reset:
            r = r + np.array([0.2, 0.05])      # a goal region as wide as the reset interval
        out.append((path, O.distances_left(path, O.distance_func(r)), r))
    return out


@pytest.mark.parametrize("n,K,max_steps,eta,seed,N,H,n_plans,near", [(40, 48, 17, 0.7, 9, 96, 3, 3, 1), (70, 30, 11, 1.0, 21, 40, 4, 5, 2),
                                                                      (33, 40, 23, 0.35, 4, 64, 2, 2, 1)])
def test_vec_smartstart_every_step_against_the_scalar_logic(ssc, n, K, max_steps, eta, seed, N, H, n_plans, near):
    env, agent, w, (Ws, bs, norm), smart = _setup(ssc, n, max_steps, eta, seed, N=N, H=H, chunk=K, n_plans=n_plans)
    rng = np.random.default_rng(seed + 3)
    plans = _plans(rng, n_plans, near_reset=near)
    smart.pool.publish(plans)
    first, count, slots = smart.pool.pool.cpu().tolist()
    assert (first, count) == (0, n_plans)
    agent.decaying_ou_action_noise.epsilon = 0.8
    chunk = ssc.TransitionChunk(2, K, n, env.device)
    obs_start = env.observe().cpu().numpy().copy()
    t0 = 5
    env.t = t0
    smart.rollout(K, chunk, graph=False)
    torch.cuda.synchronize()
    obs, act, rew = chunk.obs.cpu().numpy(), chunk.act.cpu().numpy(), chunk.rew.cpu().numpy()
    done, obs2 = chunk.done.cpu().numpy().astype(bool), chunk.obs2.cpu().numpy()
    modes = smart.mode_log.cpu().numpy()
    final_obs = env.observe().cpu().numpy()
    nm32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}
    N, H, nav_seed = smart.nav.N, smart.nav.H, smart.nav.seed
    mode = np.zeros(n, bool); plan = np.zeros(n, int); idx = np.zeros(n, int); dact = np.zeros(n, int); ou = np.zeros(n)
    ids = np.uint64(env.env_id0) + np.arange(n, dtype=np.uint64)
    n_nav = n_agent = n_handover = n_new_nav = n_close = 0
    assert np.array_equal(obs[:, 0, :].T, obs_start)
    for k in range(K):
        t = t0 + k
        g = O.ou_gaussian(env._seed, ids, np.uint64(t))
        a_net = O.actor_forward(obs[:, k, :].T, **w, obs_clip=5.0)[:, 0]
        for i in range(n):
            s = obs[:, k, i]
            assert modes[k, i] == mode[i], (k, i)
            if mode[i]:
                n_nav += 1
                wp, left, r = plans[plan[i]]
                A = O.mpc_action_samples(nav_seed, int(ids[i]), N, H, 1, t, [-1.0], [1.0])
                S = O.dyn_forward_sim(s, A, nm32, Ws, bs)
                scores, best_score, _, _ = O.mpc_scores_add_delta(S, np.asarray(wp, np.float32), np.asarray(left, np.float32),
                                                              np.asarray(r, np.float32), idx[i])
                noise = 0.005 * O.mpc_noise_gaussian(nav_seed, np.array([ids[i]], np.uint64), t, 0)[0]
                cand = np.argmin(np.abs(A[:, 0, 0] + noise - act[k, i]))
                assert abs(A[cand, 0, 0] + noise - act[k, i]) <= 1e-6, (k, i)
                assert scores[cand] >= best_score - 1e-3 * max(1.0, abs(best_score)), (k, i)
            else:
                n_agent += 1
                ou[i] = O.ou_step(ou[i], g[i], 0.4, 0.6)
                expect = O.ddpg_action(a_net[i], ou[i], 0.8)
                assert abs(expect - act[k, i]) <= 2e-5, (k, i, expect, act[k, i])
            p2, v2, r_, d_ = O.mc_step(s[0], s[1], act[k, i])
            assert abs(p2 - obs2[0, k, i]) <= 2.4e-7 and abs(v2 - obs2[1, k, i]) <= 1e-8
            assert abs(r_ - rew[k, i]) <= 1e-6 * max(1.0, abs(r_))
            if mode[i]:
                dact[i] += 1
                wp, left, r = plans[plan[i]]
                idx[i], dact[i], at_goal = O.nav_observe(obs2[:, k, i], wp, r, idx[i], dact[i], give_up=2, final_steps=4)
                if at_goal:
                    mode[i] = False
                    n_handover += 1
            if done[k, i]:
                ou[i], idx[i], dact[i], mode[i] = 0.0, 0, 0, False
                q = O.smartstart_new_episode(env._seed, int(ids[i]), t, eta, first, count, slots)
                if q >= 0:
                    plan[i] = q
                    reset_obs = obs[:, k + 1, i] if k + 1 < K else final_obs[i]
                    wp, left, r = plans[q]
                    close = O.distance_func(r)(reset_obs, wp[-1]) <= 1.0
                    mode[i] = not close
                    n_new_nav += int(not close)
                    n_close += int(close)
    # the loop exercised every branch
    assert n_nav > 30 and n_agent > 30 and n_new_nav > 3 and n_handover + n_close > 0, (n_nav, n_agent, n_handover, n_new_nav, n_close)
    assert np.array_equal(smart.mode.cpu().numpy().astype(bool), mode)
    assert np.array_equal(smart.pool.cur_idx.cpu().numpy(), idx) and np.array_equal(smart.nav.actions_done.cpu().numpy(), dact)
    nav_now = mode
    assert np.array_equal(smart.pool.plan_of.cpu().numpy()[nav_now], plan[nav_now])
    assert np.max(np.abs(env.ou_x.cpu().numpy() - ou)) < 2e-5
    assert env.stats.cpu().numpy()[2] == n * K and env.t == t0 + K


def _pend_plans(rng, n):
    """Plans in Pendulum observation space (cos, sin, theta-dot): short arcs near the upright position."""
    out = []
    for _ in range(n):
        L = int(rng.integers(6, 12))
        th = np.cumsum(rng.normal(scale=0.15, size=L)) + rng.uniform(-1, 1)
        thd = np.cumsum(rng.normal(scale=0.3, size=L))
        path = np.stack([np.cos(th), np.sin(th), thd], 1)
        stds, means = O.path_deltas_stds_and_means_per_dim(path)
        r = O.radii_calc(means, stds, 1, 1, 1) + 1e-2
        out.append((path, O.distances_left(path, O.distance_func(r)), r))
    return out


@pytest.mark.parametrize("env_name", ["MountainCarContinuous-v0", "Pendulum-v1"])
def test_vec_smartstart_graph_replay_equals_step_by_step(ssc, env_name):
    """The captured five-launch step replayed K times == the same launches enqueued one by one, over two chunks with a
    pool refresh in between (log, modes, env / navigator / OU state, statistics, episode records), for both envs."""
    res = []
    pend = env_name.startswith("Pendulum")
    plans = (lambda rng, k, near: _pend_plans(rng, k)) if pend else (lambda rng, k, near: _plans(rng, k, near_reset=near))
    for graph in (True, False):
        n, K = 300, 12
        env, agent, w, _, smart = _setup(ssc, n, 9, 0.6, 4, N=32, H=2, chunk=K, env_name=env_name)
        rng = np.random.default_rng(1)
        smart.pool.publish(plans(rng, 3, 1))
        ring = ssc.EpisodeRing(8192, "cuda")
        chunk = ssc.TransitionChunk(env.obs_dim, K, n, env.device)
        logs = []
        for c in range(2):
            smart.rollout(K, chunk, ring=ring, graph=graph)
            torch.cuda.synchronize()
            logs.append([getattr(chunk, col).clone() for col in ("obs", "act", "rew", "done", "obs2")] + [smart.mode_log.clone()])
            smart.pool.publish(plans(rng, 3, 0))
        res.append((env, smart, logs, ring))
    (eg, sg, lg, rg), (ee, se, le, re_) = res
    for c in range(2):
        for a, b in zip(lg[c], le[c]):
            assert torch.equal(a, b), c
    assert int(lg[0][5].sum()) > 0 and int(lg[1][5].sum()) > 0                     # somebody navigated
    for a, b in ((eg.s0, ee.s0), (eg.s1, ee.s1), (eg.steps, ee.steps), (eg.ou_x, ee.ou_x), (sg.mode, se.mode),
                 (sg.pool.plan_of, se.pool.plan_of), (sg.pool.cur_idx, se.pool.cur_idx), (sg.nav.actions_done, se.nav.actions_done)):
        assert torch.equal(a, b)
    a, b = eg.stats.cpu().numpy(), ee.stats.cpu().numpy()
    assert a[2] == b[2] == 2 * 12 * 300 and a[3] == b[3] and abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0]))
    (ig, lg_, _), dg = rg.drain()
    (ie, le_, _), de = re_.drain()
    assert dg == de == 0 and sorted(zip(ig.tolist(), lg_.tolist())) == sorted(zip(ie.tolist(), le_.tolist()))


def test_rl_train_vec_smartstart_end_to_end(ssc):
    """rl_train_vec_smartstart: selection on the device ring -> plans on offer -> envs navigate and hand over -> replay ->
    learner, for a few chunks.  Checks the plumbing end to end (selections happen, plans come from recorded episodes,
    some envs navigate and some of them hand over, the learner runs, eta / epsilon decay once per generation)."""
    n, K, chunks = 256, 16, 14
    env, agent, w, _, smart = _setup(ssc, n, 24, 0.9, 2, N=32, H=3, chunk=K, n_plans=2)
    agent.training_enabled = True
    agent.batch_size, agent.num_train_iterations = 64, 3
    smart.eta_decay_factor = 0.9
    nav_steps, seen = [], []

    def on_chunk(c, out, sm):
        nav_steps.append(int(sm.mode_log.sum()))
        seen.append(sm.pool.published)
    summary, losses, replay = ssc.rl_train_vec_smartstart(env, smart, chunks, chunk_steps=K, replay_capacity=1 << 15,
                                                          train_iters=3, on_chunk=on_chunk)
    torch.cuda.synchronize()
    assert len(summary.episodes) >= n                               # 24-step time limit: every env finished episodes
    assert smart.selections >= 5 and seen[-1] >= 10                 # a selection per chunk once the ring holds episodes
    assert sum(nav_steps) > 200, nav_steps                          # envs really navigated ...
    assert nav_steps[0] == 0                                        # ... but not before the first plans existed
    assert len(losses) >= chunks - 2 and all(bool(torch.isfinite(l).all()) for l in losses)
    assert smart.eta < 0.9 and agent.decaying_ou_action_noise.epsilon < 1.0
    # the plans on offer are recorded episodes: waypoints inside the env's state box
    wp = smart.pool.wp.view(smart.pool.n_slots, smart.pool.w_max, 2)
    first, count, slots = smart.pool.pool.cpu().tolist()
    L = int(smart.pool.wp_len[first])
    pts = wp[first, :L].cpu().numpy()
    assert count >= 1 and L >= 2 and pts[:, 0].min() >= -1.2001 and pts[:, 0].max() <= 0.6001 and np.abs(pts[:, 1]).max() <= 0.0701


def test_vec_smartstart_dynamics_model_aggregation(ssc):
    """train_dynamics_model on the device ring (NND_MB_agent.py:437-480 vectorised): a model with random weights, retrained on
    transitions the envs produced, predicts those transitions far better than before; the MFMA weight image follows; the
    loop's ``aggregate_every`` hook runs it."""
    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd.agents import NND_MB_agent
    n, K = 512, 32
    env, agent, w, (Ws, bs, _), smart = _setup(ssc, n, 60, 0.5, 6, N=16, H=3, chunk=K, n_plans=2, log_modes=False)
    # statistics of real MountainCar transitions, random-policy data
    denv = ssc.VecEnv("MountainCarContinuous-v0", 64, seed=3)
    ch = denv.rollout(200, ssc.RandomPolicy())
    ts = ssc.dataset_from_chunk(ch)
    from smartstartcontinuous_amd import collect_samples as cs
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    smart.model.set_norm(dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz)))
    smart.model.precision = "bf16_mfma"
    X, Z = ts.normalised(dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz)))
    mse = lambda: float(((smart.model.forward(X, precision="f32") - Z) ** 2).mean().item())
    before = mse()
    summary, losses, replay = ssc.rl_train_vec_smartstart(env, smart, 6, chunk_steps=K, replay_capacity=1 << 16, train_iters=1,
                                                          aggregate_every=2, aggregate_kwargs=dict(n_epoch=6, batchsize=256))
    torch.cuda.synchronize()
    assert getattr(smart, "_trainings", 0) == 2
    after = mse()
    assert after < 0.25 * before and np.isfinite(after), (before, after)
    # the fused kernel reads the retrained weights (image refreshed in place)
    y32 = smart.model.forward(X[:512], precision="f32")
    y16 = smart.model.forward(X[:512], precision="bf16_mfma")
    assert float((y32 - y16).abs().max()) < 5e-2 * max(1.0, float(y32.abs().max()))


def test_plan_pool_too_small_for_the_refresh_cadence_is_reported_and_harmless(ssc):
    """``n_slots`` is sized from the constructor's ``chunk_steps``; rolling SHORTER chunks (or a hand-sized pool)
    re-publishes a slot under envs that still follow it.  ``PlanPool.publish`` reports that once (RuntimeWarning) and the
    kernels clamp the waypoint index to the new plan, so nothing is read outside a plan: after every chunk each navigating
    env's index lies inside the plan it follows, and every logged value is finite."""
    n, K, max_steps = 96, 4, 17
    env, agent, w, _, smart = _setup(ssc, n, max_steps, 1.0, 31, N=32, H=3, chunk=64, n_plans=3)
    assert smart.pool.n_slots == 9                                   # 3 plans x (ceil(17 / 64) + 2): sized for 64-step chunks
    rng = np.random.default_rng(5)
    chunk = ssc.TransitionChunk(2, K, n, "cuda")
    with pytest.warns(RuntimeWarning, match="too small for this refresh cadence"):
        for c in range(12):                                          # a refresh every 4 steps: a slot comes round after 12 < 17 steps
            plans = _plans(rng, 3, 0)
            if c % 2:                                                # alternate long and short plans: indices beyond the new length
                plans = [(p[:3], l[:3], r) for p, l, r in plans]
            smart.pool.publish(plans, now=env.t, min_age=max_steps)
            smart.rollout(K, chunk)
            torch.cuda.synchronize()
            mode = smart.mode.cpu().numpy().astype(bool)
            cur = smart.pool.cur_idx.cpu().numpy()
            wlen = smart.pool.wp_len.cpu().numpy()[smart.pool.plan_of.cpu().numpy()]
            assert (cur[mode] < wlen[mode]).all() and (cur >= 0).all()
            for col in (chunk.obs, chunk.act, chunk.rew, chunk.obs2):
                assert bool(torch.isfinite(col).all())
    assert int(mode.sum()) > 0 or int(smart.mode_log.sum()) > 0       # envs did navigate


def test_rl_train_vec_smartstart_overlapped_selection(ssc):
    """``overlap_selection=True``: the selection of a chunk runs on a side stream while the chunk rolls and its plans go on
    offer one chunk later.  Deterministic (two runs agree: every cross-stream hand-over -- ring and networks into the
    selection, plans back onto the rollout's stream -- is ordered by events), plans still come from recorded episodes, envs
    navigate, and the first plans appear one refresh later than in the sequential loop."""
    def run(overlap):
        n, K, chunks = 512, 16, 12
        env, agent, w, _, smart = _setup(ssc, n, 24, 0.9, 2, N=32, H=3, chunk=K, n_plans=2)
        agent.training_enabled = True
        agent.batch_size, agent.num_train_iterations = 64, 3
        nav_steps, published = [], []

        def on_chunk(c, out, sm):
            nav_steps.append(int(sm.mode_log.sum()))
            published.append(sm.pool.published)
        summary, losses, replay = ssc.rl_train_vec_smartstart(env, smart, chunks, chunk_steps=K, replay_capacity=1 << 16, train_iters=3,
                                                              on_chunk=on_chunk, overlap_selection=overlap)
        torch.cuda.synchronize()
        return sorted(summary.episodes), nav_steps, published, replay.s.clone(), agent.actor_flat.clone(), smart.selections
    a, b, seq = run(True), run(True), run(False)
    assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert sum(a[1]) > 200 and a[5] >= 5                              # envs navigate on the overlapped loop's plans
    first = lambda pub: next(i for i, v in enumerate(pub) if v > 0)
    # sequential: plans published before chunk 2 rolls (on_chunk index 2); overlapped: selected during chunk 2, on offer from chunk 3
    assert first(a[2]) == first(seq[2]) and a[1][first(a[2])] == 0 and sum(seq[1][:first(seq[2]) + 1]) >= 0
