"""GPU end-to-end tests of the host drop-in layer: rlTrain with the GPU-backed agents, the
navigator's get_action against the oracle pipeline, and the vectorised loop."""
import numpy as np
import pytest

from oracle import ssc_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def test_rltrain_with_ddpg_agent(ssc):
    """The reference's DDPG_Baselines_example flow (examples/continuous/DDPG_Baselines_example.py:28-80)
    with unchanged call structure: make_timed_env -> agent -> rlTrain."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    env = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(0.33, max_episode_steps=60)
    assert env.spec.id == "MountainCarContinuousActionX0.33-v0"
    agent = DDPG_Baselines_agent(env, None, buffer_size=100000, batch_size=64, num_train_iterations=1,
                                 num_steps_before_train=1, ou_epsilon=1.0, ou_min_epsilon=0.01,
                                 ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6, ou_theta=.15, actor_lr=0.001,
                                 actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32,
                                 lastLayerTanh=True, seed=3)
    np.random.seed(0)
    summary = ssc.rlTrain(agent, env, print_results=False, print_steps=False, num_episodes=3, max_steps=1000)
    assert [e[0] for e in summary.episodes] == [60, 60, 60]          # TimeLimit(60)
    # a bare DDPG agent never marks episode starts (start_new_episode is `pass`, DDPG_Baselines_agent.py:249-250)
    assert len(agent.replay_buffer) == 180 and len(agent.replay_buffer.episode_starting_indices) == 0
    assert abs(agent.decaying_ou_action_noise.epsilon - 0.99 ** 3) < 1e-12
    s, a, r, t, s2 = agent.replay_buffer.all_batch()
    assert np.all(np.abs(a) <= 1.0) and t.sum() == 3 and np.allclose(r, -0.1 * a[:, 0] ** 2, atol=1e-6)
    # the stored transitions obey the reference dynamics (power_scalar 0.33)
    p2, v2, rr, _ = O.mc_step(s[:, 0], s[:, 1], a[:, 0], O.mc_power(0.33))
    assert np.max(np.abs(p2 - s2[:, 0])) <= 2.4e-7 and np.max(np.abs(v2 - s2[:, 1])) <= 1e-8
    # noise-free action == actor forward == oracle
    agent.decaying_ou_action_noise.epsilon = 0.0
    obs = np.array([-0.5, 0.01])
    w = {k: v.cpu().numpy() for k, v in agent.weights.items()}
    ref = O.actor_forward(obs[None, :].astype(np.float32), **w)[0]
    assert np.max(np.abs(agent.get_action(obs) - ref)) <= 1e-5
    # the same agent drives the fused kernel
    venv = ssc.VecEnv("MountainCarContinuous-v0", 256, seed=1)
    chunk = venv.rollout(8, agent.as_policy(precision="f32"))
    torch.cuda.synchronize()
    o = chunk.obs.cpu().numpy().transpose(1, 2, 0).reshape(-1, 2)
    assert np.max(np.abs(chunk.act.cpu().numpy().reshape(-1) - np.clip(O.actor_forward(o, **w)[:, 0], -1, 1))) <= 1e-5
    assert int(agent._adam_t[0].item()) > 100      # observe()/end_episode() trained once the buffer held 64 records


def test_navigator_get_action_matches_oracle_pipeline(ssc, golden_dir):
    from smartstartcontinuous_amd.agents import NND_MB_agent
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    env = ssc.make("MountainCarContinuous-v0", seed=2)
    agent = NND_MB_agent(env, None, horizon=4, num_control_samples=5000, num_fc_layers=1, depth_fc_layers=32,
                         training_data=dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"]),
                         precision="f32", seed=77)
    # plan along one of the reference's recorded rollouts
    path = g["states_val"][3, :120]
    start = path[0] + [0.002, 0.0005]
    agent.start_new_episode_plan(start, path)
    assert len(agent.desired_states) >= 2 and agent.distances_left[-1] == 0
    stds, means = O.path_deltas_stds_and_means_per_dim(path)
    assert np.allclose(agent.radii, O.radii_calc(means, stds, 1, 1, 1))
    assert np.array_equal(agent.desired_states, O.waypoints_from_path(O.path_shortcutter(path, O.distance_func(agent.radii), 1)))
    action, best_path = agent.get_action_with_predicted_states(start)
    assert action.shape == (1,) and best_path.shape == (5, 2) and np.allclose(best_path[0], start, atol=1e-7)
    # oracle pipeline on the same samples / weights
    A = O.mpc_action_samples(77, 0, 5000, 4, 1, 0, [-1.0], [1.0])
    Ws = [w.cpu().numpy() for w in agent.dyn_model.W]
    bs = [b.cpu().numpy() for b in agent.dyn_model.b]
    nm = NND_MB_agent.normalisation_from_data(g["dataX"], g["dataY"], g["dataZ"])
    nm32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in nm.items()}
    S = O.dyn_forward_sim(start.astype(np.float32), A, nm32, Ws, bs)
    ref_scores, ref_best_score, ref_best, _ = O.mpc_scores_add_delta(
        S, agent.desired_states, agent.distances_left, agent.radii, 0, theta=1, gamma=.75, hpf=.5)
    # the path the kernel predicted is the oracle's path for SOME sample whose reference score is within tol of the max
    d = np.max(np.abs(S - best_path[:, None, :]), axis=(0, 2))
    chosen = int(np.argmin(d))
    assert d[chosen] <= 1e-4 * max(1.0, np.abs(S).max())
    assert ref_scores[chosen] >= ref_best_score - 1e-3 * max(1.0, abs(ref_best_score))
    g0 = O.mpc_noise_gaussian(77, np.array([0], np.uint64), 0, 0)[0]
    assert abs(action[0] - (float(A[chosen, 0, 0]) + 0.005 * g0)) <= 1e-6
    # waypoint bookkeeping (NND_MB_agent.observe, :360-373)
    idx0 = agent.current_desired_state_index
    agent.observe(start, action, 0.0, agent.desired_states[min(idx0 + 1, len(agent.desired_states) - 1)], False)
    assert agent.current_desired_state_index == idx0 + 1 and agent.actions_done_for_current_waypoint == 0
    assert not agent.close_enough_to_goal(start) and agent.close_enough_to_goal(agent.desired_states[-1])


def test_navigator_retrains_every_third_plan(ssc, golden_dir):
    """NND_MB_agent.start_new_episode_plan :420-423: the dynamics model is retrained on initial + aggregated data at
    plans 0, 3, 6, ... (num_episodes_for_aggregation = 3) and left alone in between."""
    from smartstartcontinuous_amd.agents import NND_MB_agent
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    env = ssc.make("MountainCarContinuous-v0", seed=2)
    agent = NND_MB_agent(env, None, horizon=4, num_control_samples=64, num_fc_layers=1, depth_fc_layers=32,
                         training_data=dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"]),
                         precision="f32", seed=5, nEpochs=1)
    path = g["states_val"][1, :60]
    snap = lambda: [w.clone() for w in agent.dyn_model.W]
    same = lambda a, b: all(torch.equal(x, y) for x, y in zip(a, b))
    w0 = snap()
    np.random.seed(0)
    changed = []
    for k in range(5):
        if k == 2:       # transitions observed in between become the "new" rows of the next training
            agent.replay_buffer.start_new_episode(agent)
            for t in range(40):
                agent.replay_buffer.add(agent, g["states_val"][2, t], g["controls_val"][2, t], 0.0, False, g["states_val"][2, t + 1])
        agent.start_new_episode_plan(path[0], path)
        w1 = snap()
        changed.append(not same(w0, w1))
        w0 = w1
    assert changed == [True, False, False, True, False] and agent.num_episodes_finished == 5
    assert int(agent.dyn_model._adam["t"].item()) > 0
    # statistics-only agent: nothing to train on, plans still work
    nm = NND_MB_agent.normalisation_from_data(g["dataX"], g["dataY"], g["dataZ"])
    bare = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, norm=nm, precision="f32")
    bare.start_new_episode_plan(path[0], path)
    assert bare.num_episodes_finished == 1 and not hasattr(bare.dyn_model, "_adam")


def test_sharded_actor_learner_loop_world1_equals_vec_loop(ssc):
    """rl_train_sharded_ddpg (rollout -> RCCL gather -> learner -> parameter broadcast) rehearsed with a 1-rank RCCL
    group: with the same seeds it must reproduce rl_train_vec_ddpg fed the same last-g steps, bit for bit."""
    import os
    import torch.distributed as dist
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    from smartstartcontinuous_amd.sharding import rl_train_sharded_ddpg

    def setup():
        env = ssc.VecEnv("MountainCarContinuous-v0", 256, seed=11, max_episode_steps=60)
        one = ssc.SingleEnvView(ssc.VecEnv("MountainCarContinuous-v0", 1, seed=11))
        agent = DDPG_Baselines_agent(one, None, batch_size=64, num_train_iterations=5, actor_h1=64, actor_h2=32,
                                     critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=4)
        return env, agent
    env_a, agent_a = setup()
    s_a, losses_a, replay_a = ssc.rl_train_vec_ddpg(env_a, agent_a, num_chunks=4, chunk_steps=48, replay_capacity=4096,
                                                    replay_last_steps=8, seed=3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        env_b, agent_b = setup()
        s_b, losses_b, replay_b = rl_train_sharded_ddpg(env_b, agent_b, num_chunks=4, chunk_steps=48, rank=0, world=1,
                                                        gather_steps=8, replay_capacity=4096, seed=3)
        # pipelined mode: chunk j is rolled with parameter generation j - 2 (one chunk stale) -- replayed by hand
        env_c, agent_c = setup()
        s_c, losses_c, replay_c = rl_train_sharded_ddpg(env_c, agent_c, num_chunks=5, chunk_steps=48, rank=0, world=1,
                                                        gather_steps=8, replay_capacity=4096, seed=3, pipelined=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
    from smartstartcontinuous_amd.sharding import _views_like
    env_d, agent_d = setup()
    replay_d = DeviceReplayBuffer(4096, 2, 1, env_d.device, seed=3)
    gens = {-2: agent_d.actor_flat.clone(), -1: agent_d.actor_flat.clone()}
    eps_gen = {-2: agent_d.decaying_ou_action_noise.epsilon, -1: agent_d.decaying_ou_action_noise.epsilon}
    losses_d = []
    from smartstartcontinuous_amd.vec_env import EpisodeRing
    ring_d, finished = EpisodeRing(1 << 16, env_d.device), 0.0
    for j in range(5):
        pol = agent_d.as_policy()
        pol.weights = _views_like(gens[j - 2], agent_d.weights)
        pol.ou_epsilon = float(eps_gen[j - 2])          # epsilon is part of the generation (rides in the broadcast)
        chunk = env_d.rollout(48, pol, ring=ring_d)
        tail = ssc.TransitionChunk.from_columns(chunk.obs[:, -8:], chunk.act[-8:], chunk.rew[-8:], chunk.obs2[:, -8:], chunk.done[-8:])
        replay_d.append_chunk(tail, reward_scale=agent_d.reward_scale)
        losses_d.append(agent_d.train_from(replay_d, None))
        gens[j] = agent_d.actor_flat.clone()
        (_, lens, _), _ = ring_d.drain()
        finished += len(lens) / float(env_d.n)          # epsilon decays once per episode per env, like the loop under test
        while finished >= 1.0:
            agent_d.decaying_ou_action_noise.reduce_epsilon()
            finished -= 1.0
        eps_gen[j] = agent_d.decaying_ou_action_noise.epsilon
    assert agent_c.decaying_ou_action_noise.epsilon == agent_d.decaying_ou_action_noise.epsilon < 1.0
    assert torch.equal(agent_c.actor_flat, agent_d.actor_flat) and torch.equal(agent_c.critic_flat, agent_d.critic_flat)
    assert len(losses_c) == 5 and all(torch.equal(x, y) for x, y in zip(losses_c, losses_d))
    assert torch.equal(replay_c.s, replay_d.s) and torch.equal(env_c.s0, env_d.s0)
    assert torch.equal(agent_a.actor_flat, agent_b.actor_flat) and torch.equal(agent_a.critic_flat, agent_b.critic_flat)
    assert len(losses_a) == len(losses_b) == 4 and all(torch.equal(x, y) for x, y in zip(losses_a, losses_b))
    assert sorted(s_a.episodes) == sorted(s_b.episodes) and len(s_a) > 0      # ring order is atomics order
    assert replay_a.count == replay_b.count == 4 * 8 * 256 and torch.equal(replay_a.s, replay_b.s)
    assert not torch.equal(agent_b.actor_flat, setup()[1].actor_flat)          # it did learn something


def test_sharded_pipelined_loop_clips_observations_like_the_learner(ssc):
    """rl_train_sharded_ddpg(pipelined=True) on Pendulum-v1 with every env started at |theta-dot| = 7: the actors must
    see clip(obs, -5, 5) exactly as the learner does (ddpg_editted.py:106-109) -- a hand replay of the pipelined
    schedule with the agent's own policy (observation clip included) has to reproduce it bit for bit, and the same
    replay WITHOUT the clip must not (the case binds)."""
    import dataclasses
    import os
    import torch.distributed as dist
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
    from smartstartcontinuous_amd.sharding import _views_like, rl_train_sharded_ddpg
    from smartstartcontinuous_amd.vec_env import EpisodeRing

    def setup():
        env = ssc.VecEnv("Pendulum-v1", 256, seed=13, max_episode_steps=40)
        env.reset()
        env.s1.copy_(torch.where(env.s1 >= 0, 7.0, -7.0))
        one = ssc.SingleEnvView(ssc.VecEnv("Pendulum-v1", 1, seed=13))
        agent = DDPG_Baselines_agent(one, None, batch_size=64, num_train_iterations=5, actor_h1=64, actor_h2=32,
                                     critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=4)
        return env, agent
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29537"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        env_c, agent_c = setup()
        assert agent_c.as_policy().obs_clip == 5.0
        _, losses_c, replay_c = rl_train_sharded_ddpg(env_c, agent_c, num_chunks=5, chunk_steps=24, rank=0, world=1,
                                                      gather_steps=8, replay_capacity=4096, seed=3, pipelined=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()

    def hand(clip):
        env_d, agent_d = setup()
        replay_d = DeviceReplayBuffer(4096, 3, 1, env_d.device, seed=3)
        gens = {-2: agent_d.actor_flat.clone(), -1: agent_d.actor_flat.clone()}
        eps_gen = {-2: agent_d.decaying_ou_action_noise.epsilon, -1: agent_d.decaying_ou_action_noise.epsilon}
        ring_d, finished = EpisodeRing(1 << 16, env_d.device), 0.0
        for j in range(5):
            pol = dataclasses.replace(agent_d.as_policy(), weights=_views_like(gens[j - 2], agent_d.weights),
                                      ou_epsilon=float(eps_gen[j - 2]))
            if not clip:
                pol.obs_clip = 0.0
            chunk = env_d.rollout(24, pol, ring=ring_d)
            tail = ssc.TransitionChunk.from_columns(chunk.obs[:, -8:], chunk.act[-8:], chunk.rew[-8:], chunk.obs2[:, -8:], chunk.done[-8:])
            replay_d.append_chunk(tail, reward_scale=agent_d.reward_scale)
            agent_d.train_from(replay_d, None)
            gens[j] = agent_d.actor_flat.clone()
            (_, lens, _), _ = ring_d.drain()
            finished += len(lens) / float(env_d.n)
            while finished >= 1.0:
                agent_d.decaying_ou_action_noise.reduce_epsilon()
                finished -= 1.0
            eps_gen[j] = agent_d.decaying_ou_action_noise.epsilon
        return env_d, agent_d, replay_d
    env_d, agent_d, replay_d = hand(True)
    assert torch.equal(replay_c.s, replay_d.s) and torch.equal(replay_c.a, replay_d.a)
    assert torch.equal(agent_c.actor_flat, agent_d.actor_flat) and torch.equal(env_c.s0, env_d.s0)
    assert float(replay_d.s[:, 2].abs().max()) > 5.0                       # the clip had something to do
    _, _, replay_u = hand(False)
    assert not torch.equal(replay_c.a, replay_u.a)


def test_navigator_loads_and_saves_training_data_like_the_reference(ssc, golden_dir, tmp_path):
    """load_existing_training_data / save_training_data (NND_MB_agent.py:203-213, :289-296): the directory layout of
    the reference (<models>/NND_MB_agent/<name>/training_data/*.npy), so its shipped data sets load unchanged."""
    from smartstartcontinuous_amd.agents import NND_MB_agent
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    d = tmp_path / "default" / "training_data"
    d.mkdir(parents=True)
    for k in ("dataX", "dataY", "dataZ", "states_val", "controls_val"):
        np.save(d / (k + ".npy"), g[k])
    env = ssc.make("MountainCarContinuous-v0", seed=2)
    agent = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", model_root=str(tmp_path),
                         load_existing_training_data=True, load_dir_name="default", save_training_data=True,
                         save_dir_name="copy")
    nm = NND_MB_agent.normalisation_from_data(g["dataX"], g["dataY"], g["dataZ"])
    assert np.allclose([agent.dyn_model.norm.std_x[i] for i in range(2)], nm["std_x"], rtol=1e-6)
    assert agent._train_inputs.shape == (8300, 3) and agent.states_val.shape == (20, 333, 2)
    for k in ("dataX", "dataY", "dataZ", "states_val", "controls_val"):
        assert np.array_equal(np.load(tmp_path / "copy" / "training_data" / (k + ".npy")), g[k])
    # a self-collected data set is written in the same format
    own = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", model_root=str(tmp_path),
                       save_training_data=True, save_dir_name="own", num_rollouts_train=5, num_rollouts_val=3,
                       steps_per_rollout_train=50, steps_per_rollout_val=40)
    X = np.load(tmp_path / "own" / "training_data" / "dataX.npy")
    assert X.shape == (5 * 49, 2) and X.dtype == np.float64 and np.array_equal(X, own.dataX.double().cpu().numpy())
    assert np.load(tmp_path / "own" / "training_data" / "states_val.npy").shape == (3, 40, 2)
    again = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", model_root=str(tmp_path),
                         load_existing_training_data=True, load_dir_name="own")
    assert np.allclose(again.dyn_model.norm.mean_x[0], own.dyn_model.norm.mean_x[0], rtol=1e-6)
    # the trained model is kept (save_resulting_dynamics_model) and restored (load_existing_dynamics_model, :463-468)
    own.save_resulting_dynamics_model = True
    own.train_dynamics_model(nEpoch=2, rng=np.random.RandomState(0))
    assert (tmp_path / "own" / "models" / "finalModel.npz").exists() and (tmp_path / "own" / "models" / "model_numTrain1.npz").exists()
    restored = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", model_root=str(tmp_path),
                            load_existing_training_data=True, load_existing_dynamics_model=True, load_dir_name="own", seed=99)
    assert not torch.equal(restored.dyn_model.W[0], own.dyn_model.W[0])
    path = np.load(tmp_path / "own" / "training_data" / "states_val.npy")[0]
    restored.start_new_episode_plan(path[0], path)               # first plan "trains" = restores the saved model
    assert all(torch.equal(a, b) for a, b in zip(restored.dyn_model.W + restored.dyn_model.b, own.dyn_model.W + own.dyn_model.b))
    assert bytes(restored.dyn_model.norm) == bytes(own.dyn_model.norm)


def test_rl_train_vec_summary(ssc):
    env = ssc.VecEnv("MountainCarContinuous-v0", 512, seed=9, max_episode_steps=50)
    seen = []
    summ = ssc.rl_train_vec(env, ssc.RandomPolicy(), num_chunks=3, chunk_steps=40,
                            on_chunk=lambda chunk, e: seen.append(int(chunk.done.sum().item())))
    stats = env.stats.cpu().numpy()
    assert len(summ) == int(stats[3]) == sum(seen) and summ.dropped_episode_records == 0
    assert len(summ) == 512 * 2                                 # 120 steps with a 50-step limit
    assert all(l == 50 or r > 0 for l, r in summ.episodes)      # timed out, or ended by the +100 goal reward
    assert abs(sum(r for _, r in summ.episodes) + float(env.ep_ret.sum()) - stats[0]) < 1.0


def test_replay_buffer_ingests_chunk(ssc):
    from smartstartcontinuous_amd.replay_buffer import ReplayBuffer
    env = ssc.VecEnv("MountainCarContinuous-v0", 8, seed=4, max_episode_steps=10)
    chunk = env.rollout(25, ssc.RandomPolicy())
    owner = object()
    rb = ReplayBuffer(owner, 1000)
    rb.start_new_episode(owner)
    rb.add_chunk(owner, chunk, env_index=5)
    assert len(rb) == 25 and list(rb.episode_starting_indices) == [0, 10, 20]
    s, a, r, t, s2 = rb.all_batch()
    assert np.array_equal(np.nonzero(t)[0], [9, 19])
    assert np.allclose(s[:, 0], chunk.obs[0, :, 5].cpu().numpy()) and np.allclose(a[:, 0], chunk.act[:, 5].cpu().numpy())
    path = np.asarray(rb.get_episodic_path_to_buffer_index(14))
    assert path.shape == (6, 2) and np.allclose(path[0], s[10])


def test_critic_kde_ucb_kernels(ssc):
    from smartstartcontinuous_amd import smartstart as SS
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    rng = np.random.default_rng(4)
    env = ssc.make("MountainCarContinuous-v0")
    for (h1, h2, llt) in [(64, 32, True), (200, 100, False)]:
        agent = DDPG_Baselines_agent(env, None, actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2, lastLayerTanh=llt, seed=h1)
        cw = {k: (v.cpu().numpy() + (0.05 * rng.normal(size=tuple(v.shape))).astype(np.float32)) for k, v in agent.critic_weights.items()}
        agent.set_critic_weights(cw)
        obs = rng.uniform(-1, 1, size=(777, 2)).astype(np.float32)
        act = rng.uniform(-1, 1, size=(777, 1)).astype(np.float32)
        q = agent.critic(obs, act).cpu().numpy()
        ref = O.critic_forward(obs, act, **cw, last_layer_tanh=llt)[:, 0]
        assert np.max(np.abs(q - ref)) <= 1e-5 * max(1.0, np.abs(ref).max())
        aw = {k: v.cpu().numpy() for k, v in agent.weights.items()}
        v = agent.get_state_value(obs)
        ref_v = O.critic_forward(obs, O.actor_forward(obs, **aw, last_layer_tanh=llt), **cw, last_layer_tanh=llt)
        assert v.shape == (777, 1) and np.max(np.abs(v - ref_v)) <= 2e-5 * max(1.0, np.abs(ref_v).max())
        assert agent.get_state_value(obs[0]).shape == (1,)
        # DDPG_editted clips network inputs to observation_range = (-5, 5) (ddpg_editted.py:106-109)
        big = (obs * 9.0).astype(np.float32)
        assert np.abs(big).max() > 5.0
        qb = agent.critic(big, act).cpu().numpy()
        ref_b = O.critic_forward(big, act, **cw, last_layer_tanh=llt, obs_clip=5.0)[:, 0]
        assert np.max(np.abs(qb - ref_b)) <= 1e-5 * max(1.0, np.abs(ref_b).max())
        assert np.max(np.abs(qb - O.critic_forward(big, act, **cw, last_layer_tanh=llt)[:, 0])) > 1e-3   # the clip matters
        ab = agent.actor(big).cpu().numpy()
        ref_ab = O.actor_forward(big, **aw, last_layer_tanh=llt, obs_clip=5.0)
        assert np.max(np.abs(ab - ref_ab)) <= 2e-5
    # KDE (Scott) + UCB against the oracle (itself pinned to scipy.stats.gaussian_kde on the CPU)
    # d = 1, 2, 3 run the dimension-templated kernel, d = 5 the guarded generic one; n = 4099 leaves a ragged last trip
    for n, d, m in [(5000, 2, 2000), (100000, 2, 2000), (3000, 3, 77), (4099, 1, 9), (2500, 5, 130)]:
        data = (np.cumsum(rng.normal(size=(n, d)) * 0.01, axis=0) % 1.0).astype(np.float32)
        pts = data[rng.integers(0, n, m)]
        dt, pt = torch.as_tensor(data, device="cuda"), torch.as_tensor(pts, device="cuda")
        wh, norm = SS.kde_scott_bandwidth(dt)
        cov, wh_ref, norm_ref = O.kde_scott(data)
        assert np.allclose(wh, wh_ref, rtol=1e-5) and np.isclose(norm, norm_ref, rtol=1e-6)
        pdf = SS.kde_evaluate(dt, pt, wh, norm).cpu().numpy()
        ref = O.kde_evaluate(data, pts, wh_ref, norm_ref)
        assert np.max(np.abs(pdf / ref - 1)) <= 2e-4
        values = rng.normal(size=m).astype(np.float32)
        ucb, best = SS.ucb_argmax(torch.as_tensor(values, device="cuda"), torch.as_tensor(pdf, device="cuda"), n, 0.003, 1.0, 2.0)
        ref_ucb, ref_best = O.smart_start_ucb(values, pdf, n, 0.003)
        assert np.max(np.abs(ucb.cpu().numpy() - ref_ucb)) <= 1e-4 * np.abs(ref_ucb).max()
        assert ref_ucb[int(best.item())] >= ref_ucb.max() - 1e-4 * abs(ref_ucb.max())


def test_rltrain_with_smartstart_agent(ssc, golden_dir):
    """The SmartStart example flow (examples/continuous/SmartStart_DDPG_Baselines_example.py:29-130):
    base DDPG agent wrapped by SmartStartContinuous, driven by rlTrain, sharing one replay buffer."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    env = ssc.make("MountainCarContinuous-v0", seed=5)
    env.vec.params.max_episode_steps = 40
    base = DDPG_Baselines_agent(env, None, buffer_size=100000, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                                ou_sigma=0.6, lastLayerTanh=True, seed=1)
    agent = ssc.SmartStartContinuous(base, env, None, eta=1.0, eta_decay_factor=0.99, n_ss=200, print_ss_stuff=False,
                                     nnd_mb_horizon=4, nnd_mb_num_control_samples=500, nnd_mb_num_fc_layers=1,
                                     nnd_mb_depth_fc_layers=32, nnd_mb_precision="f32",
                                     nnd_mb_training_data=dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"]))
    assert agent.replay_buffer is base.replay_buffer and base.replay_buffer.main_agent is agent
    assert agent.get_summary_name() == "SmartStartC_DDPG_Baselines_agent"
    np.random.seed(1)
    import random
    random.seed(1)
    summary = ssc.rlTrain(agent, env, print_results=False, print_steps=False, num_episodes=4, max_steps=100)
    assert len(summary) == 4 and len(agent.replay_buffer) == sum(e[0] for e in summary.episodes)
    assert len(summary.smart_start_episodes) >= 1                 # eta = 1: from the 2nd episode on a path exists
    assert abs(agent.eta - 0.99 ** 4) < 1e-12 and len(agent.times_for_smart_start) >= 3
    # selection against the oracle on the final buffer
    idx = agent.replay_buffer.get_possible_smart_start_indices(150)
    ucb, best = agent.smart_start_scores(idx)
    states = agent.replay_buffer.get_all_states()
    _, wh, norm = O.kde_scott(states)
    s2 = agent.replay_buffer._gather(idx)[4]
    aw = {k: v.cpu().numpy() for k, v in base.weights.items()}
    cw = {k: v.cpu().numpy() for k, v in base.critic_weights.items()}
    vals = O.critic_forward(s2, O.actor_forward(s2, **aw), **cw)[:, 0]
    vol = O.hyperellipsoid_volume(agent.nnd_mb_agent.radii) if agent.nnd_mb_agent.radii is not None else 1
    ref_ucb, ref_best = O.smart_start_ucb(vals, O.kde_evaluate(states, s2, wh, norm), len(agent.replay_buffer), vol)
    assert np.max(np.abs(ucb.cpu().numpy() - ref_ucb)) <= 2e-3 * np.abs(ref_ucb).max()
    assert ref_ucb[best] >= ref_ucb.max() - 2e-3 * abs(ref_ucb.max())
    path = agent.get_smart_start_path()
    assert len(path) >= 2 and np.asarray(path).shape[1] == 2


class _BoxEnv:
    """Just the spaces a DDPG agent reads: obs_dim-dimensional observations, one action in [-1, 1]."""

    def __init__(self, obs_dim):
        from smartstartcontinuous_amd import spaces
        self.observation_space = spaces.Box(low=-np.ones(obs_dim, np.float32), high=np.ones(obs_dim, np.float32))
        self.action_space = spaces.Box(low=np.array([-1.0], np.float32), high=np.array([1.0], np.float32))


def _ddpg_kernel_vs_oracle(ssc, obs_dim, h1, h2, B=64, ch1=None, ch2=None, n_iters=6, llts=(True, False), cap=1000, layer_norm=False,
                           critic_l2_reg=0.0, clip_norm=None):
    """ssc_ddpg_train_ws against the fp64 restatement of ddpg_editted.py:287-339 (itself cross-checked against torch
    autograd on the CPU): parameters, targets, Adam moments, losses after ``n_iters`` iterations on batches of ``B``."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    ch1, ch2 = ch1 or h1, ch2 or h2
    rng = np.random.default_rng(11)
    env = ssc.make("MountainCarContinuous-v0") if obs_dim == 2 else _BoxEnv(obs_dim)
    for llt in llts:
        agent = DDPG_Baselines_agent(env, None, actor_h1=h1, actor_h2=h2, critic_h1=ch1, critic_h2=ch2, lastLayerTanh=llt,
                                     actor_lr=1e-3, critic_lr=1e-3, gamma=0.99, tau=0.001, batch_size=B, seed=5,
                                     training=False, layer_norm=layer_norm, critic_l2_reg=critic_l2_reg, clip_norm=clip_norm)
        # non-trivial starting point: perturb every parameter (biases and the 3e-3 output layers included)
        aw = {k: v.cpu().numpy() + (0.05 * rng.normal(size=tuple(v.shape))).astype(np.float32) for k, v in agent.weights.items()}
        cw = {k: v.cpu().numpy() + (0.05 * rng.normal(size=tuple(v.shape))).astype(np.float32) for k, v in agent.critic_weights.items()}
        agent.set_weights(aw)
        agent.set_critic_weights(cw)
        agent.target_actor_flat += 0.01
        agent.target_critic_flat -= 0.01
        cap = max(cap, B)
        s = rng.uniform(-1.2, 0.6, (cap, obs_dim)).astype(np.float32)
        if obs_dim == 3:
            s[:, 2] = rng.uniform(-8, 8, cap)            # Pendulum's theta-dot: observation_range (-5, 5) clips it
        a = rng.uniform(-1, 1, (cap, 1)).astype(np.float32)
        r = (rng.normal(size=cap) * 0.5).astype(np.float32)
        t = (rng.random(cap) < 0.1)
        s2 = (s + rng.normal(size=(cap, obs_dim)) * 0.01).astype(np.float32)
        idx = np.stack([rng.permutation(cap)[:B] for _ in range(n_iters)]).astype(np.int32)
        # oracle state (fp64 copies of what the device holds)
        o_a = {k: v.astype(np.float64) for k, v in aw.items()}
        o_c = {k: v.astype(np.float64) for k, v in cw.items()}
        o_ta = O.unflatten_params(agent.target_actor_flat.cpu().numpy().astype(np.float64), o_a)
        o_tc = O.unflatten_params(agent.target_critic_flat.cpu().numpy().astype(np.float64), o_c)
        na, nc = agent.actor_flat.numel(), agent.critic_flat.numel()
        adam = dict(m_actor=np.zeros(na), v_actor=np.zeros(na), t_actor=0, m_critic=np.zeros(nc), v_critic=np.zeros(nc), t_critic=0)
        ref_losses = []
        for it in range(n_iters):
            bi = idx[it]
            o_a, o_c, o_ta, o_tc, adam, cl, al = O.ddpg_train_step(
                o_a, o_c, o_ta, o_tc, adam, (s[bi], a[bi], r[bi], t[bi], s2[bi]), gamma=0.99, tau=0.001,
                actor_lr=1e-3, critic_lr=1e-3, last_layer_tanh=llt, obs_clip=5.0, critic_l2_reg=critic_l2_reg, clip_norm=clip_norm)
            ref_losses.append((cl, al))
        dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
        losses = agent.train_on(dev(s, torch.float32), dev(a, torch.float32), dev(r, torch.float32), dev(t, torch.uint8),
                                dev(s2, torch.float32), dev(idx, torch.int32), n_iters)
        torch.cuda.synchronize()
        assert agent._adam_t.cpu().tolist() == [n_iters, n_iters]
        got_l = losses.cpu().numpy()
        assert np.allclose(got_l, np.asarray(ref_losses), rtol=2e-4, atol=1e-6), (got_l, ref_losses)
        tol = 5e-6     # 6 Adam steps of size ~1e-3 each; fp32 kernel vs fp64 oracle
        assert np.max(np.abs(agent.actor_flat.cpu().numpy() - O.flatten_params(o_a))) <= tol
        assert np.max(np.abs(agent.critic_flat.cpu().numpy() - O.flatten_params(o_c))) <= tol
        assert np.max(np.abs(agent.target_actor_flat.cpu().numpy() - O.flatten_params(o_ta))) <= tol
        assert np.max(np.abs(agent.target_critic_flat.cpu().numpy() - O.flatten_params(o_tc))) <= tol
        assert np.allclose(agent._adam_actor[0].cpu().numpy(), adam["m_actor"], rtol=1e-3, atol=1e-7)
        assert np.allclose(agent._adam_critic[1].cpu().numpy(), adam["v_critic"], rtol=2e-3, atol=1e-9)
        # the weight VIEWS used by the forward kernels see the update
        assert np.max(np.abs(agent.weights["W2"].cpu().numpy() - o_a["W2"])) <= tol


@pytest.mark.parametrize("obs_dim,h1,h2,interpreter", [(2, 64, 32, False), (3, 64, 32, False), (2, 64, 32, True), (3, 64, 32, True),
                                                       (8, 64, 32, False), (2, 24, 20, False)])
def test_ddpg_train_kernel_vs_oracle(ssc, obs_dim, h1, h2, interpreter, monkeypatch):
    """The single-workgroup learner kernels (batch 64, layers <= 64).  obs_dim 3 is the Pendulum layout, 8 the widest
    the ABI admits (the LDS carve must hold), 24-20 exercises ragged unit groups.  The shipped 64-32 shape with a
    2-d / 3-d observation runs the shape-specialised kernel (ddpg_train_fixed.hip), everything else -- and 64-32 again
    with SSC_DDPG_INTERPRETER=1 -- the step interpreter (ddpg_train.hip)."""
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    if interpreter:
        monkeypatch.setenv("SSC_DDPG_INTERPRETER", "1")
    else:
        monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    _ddpg_kernel_vs_oracle(ssc, obs_dim, h1, h2)


@pytest.mark.parametrize("obs_dim,h1,h2,B", [(2, 128, 64, 64), (2, 128, 64, 256), (2, 128, 64, 1024),
                                             (2, 200, 100, 64), (2, 200, 100, 256), (2, 200, 100, 1024),
                                             (3, 200, 100, 64), (3, 200, 100, 256), (3, 200, 100, 1024),
                                             (2, 64, 32, 32), (2, 64, 32, 4096), (3, 64, 32, 50), (8, 37, 19, 77)])
def test_ddpg_train_wide_kernel_vs_oracle(ssc, obs_dim, h1, h2, B, monkeypatch):
    """The multi-workgroup learner (ddpg_train_wide.hip): the reference's own network grid -- actor / critic 64-32,
    128-64, 200-100 (data/ddpg_baselines_summaries/hidden_layer_size_experiment/, DDPG_Baselines_agent.py:86-92) -- at
    batch sizes 32 .. 4096, batches that do not fill the last 16-row tile (50, 77) and layer sizes off the 16-unit
    tiles (37-19), at the tolerance of the single-workgroup kernels."""
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    _ddpg_kernel_vs_oracle(ssc, obs_dim, h1, h2, B=B, llts=(True, False) if B <= 256 else (True,), cap=5000)


@pytest.mark.parametrize("obs_dim,h1,h2,B,ln,l2,clip", [(2, 64, 32, 64, False, 1e-2, None), (2, 64, 32, 64, False, 0.0, 0.05),
                                                        (2, 64, 64, 256, True, 1e-2, 0.05), (3, 200, 100, 1024, False, 1e-2, 0.5),
                                                        (8, 37, 19, 77, True, 0.3, 0.02), (3, 128, 64, 64, True, 1e-2, 5.0)])
def test_ddpg_train_l2_regularisation_and_gradient_clipping(ssc, obs_dim, h1, h2, B, ln, l2, clip, monkeypatch):
    """critic_l2_reg (ddpg_editted.py:183-191) and clip_norm (:175, :197) on the multi-workgroup learner -- alone and
    together, with and without LayerNorm, clip thresholds that bind for every variable (0.02), for some, and for none (5.0)
    -- against the oracle (checked against torch autograd in tests/test_oracle_networks.py); the reported critic loss
    carries the regularisation term."""
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    _ddpg_kernel_vs_oracle(ssc, obs_dim, h1, h2, B=B, llts=(True, False) if B <= 256 else (True,), cap=3000, layer_norm=ln,
                           critic_l2_reg=l2, clip_norm=clip)


@pytest.mark.parametrize("obs_dim,B,l2,clip", [(2, 128, 0.0, None), (2, 1024, 0.0, None), (3, 192, 0.0, None), (3, 1024, 0.0, None),
                                               (2, 256, 1e-2, 0.05), (3, 4096, 0.0, None)])
def test_ddpg_train_shipped_networks_on_batches_of_64_row_tiles(ssc, obs_dim, B, l2, clip, monkeypatch):
    """The 64-32 networks at a batch that is a multiple of 64 (the vectorised actor-learner loop trains on 1024): the
    straight-line kernel of the shipped shape runs one 64-row tile per workgroup up to the gradients
    (ddpg_train_fixed_kernel<.., TILED>) and the multi-workgroup apply pass sums the tiles -- same oracle, same tolerance;
    SSC_DDPG_WIDE=1 keeps the 16-row-tile kernel reachable for the same shape (the two agree to rounding)."""
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    llts = (True, False) if B <= 256 else (True,)
    _ddpg_kernel_vs_oracle(ssc, obs_dim, 64, 32, B=B, llts=llts, cap=5000, critic_l2_reg=l2, clip_norm=clip)
    if B == 1024:
        monkeypatch.setenv("SSC_DDPG_WIDE", "1")
        _ddpg_kernel_vs_oracle(ssc, obs_dim, 64, 32, B=B, llts=(True,), cap=5000)


def test_ddpg_train_wide_on_the_shipped_shape_and_mixed_sizes(ssc, monkeypatch):
    """SSC_DDPG_WIDE=1 sends the shipped 64-32 / batch-64 shape through the multi-workgroup kernels as well (same
    oracle, same tolerance); actor and critic of different sizes; and the workspace-less entry point says which call
    serves a wide shape."""
    import ctypes
    from smartstartcontinuous_amd import _ffi
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    monkeypatch.setenv("SSC_DDPG_WIDE", "1")
    _ddpg_kernel_vs_oracle(ssc, 2, 64, 32)
    _ddpg_kernel_vs_oracle(ssc, 3, 64, 32, llts=(True,))
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    _ddpg_kernel_vs_oracle(ssc, 2, 128, 64, ch1=200, ch2=100, B=128, llts=(True,))
    _ddpg_kernel_vs_oracle(ssc, 2, 200, 100, ch1=64, ch2=32, B=64, llts=(False,))
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=200, actor_h2=100, critic_h1=200,
                                 critic_h2=100, lastLayerTanh=True, seed=1, training=False)
    d = agent.ddpg_desc()
    z = torch.zeros(64, 2, device="cuda")
    rv = _ffi.ReplayView(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 64)
    idx = torch.zeros((1, 64), dtype=torch.int32, device="cuda")
    rc = _ffi.lib().ssc_ddpg_train(ctypes.byref(d), ctypes.byref(rv), _ffi.ptr(idx), 1, None, None)
    assert rc == _ffi.SSC_EUNSUPPORTED and b"ssc_ddpg_train_ws" in _ffi.lib().ssc_last_error()
    # a width whose 16-row tile does not fit the LDS is refused with the byte count
    big = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=512, actor_h2=512, critic_h1=512,
                               critic_h2=512, lastLayerTanh=True, seed=1, training=False)
    with pytest.raises(_ffi.SscError) as ei:
        big.train_on(z, z[:, :1].contiguous(), z[:, 0].contiguous(), torch.zeros(64, dtype=torch.uint8, device="cuda"), z, idx, 1)
    assert ei.value.code == _ffi.SSC_EUNSUPPORTED and "LDS" in str(ei.value)


def test_ddpg_training_reduces_critic_loss(ssc):
    """Functional check: repeated train iterations on one fixed data set drive the TD error down."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    rng = np.random.default_rng(2)
    env = ssc.make("MountainCarContinuous-v0")
    agent = DDPG_Baselines_agent(env, None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True,
                                 actor_lr=1e-4, critic_lr=1e-3, seed=1, training=False)
    cap = 256
    s = rng.uniform(-1.2, 0.6, (cap, 2)).astype(np.float32)
    a = rng.uniform(-1, 1, (cap, 1)).astype(np.float32)
    r = (-0.1 * a[:, 0] ** 2 + s[:, 0]).astype(np.float32)
    t = np.ones(cap, bool)                       # terminal everywhere: target_Q = r, a pure regression problem
    dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
    idx = torch.as_tensor(np.stack([rng.permutation(cap)[:64] for _ in range(600)]).astype(np.int32), device="cuda")
    losses = agent.train_on(dev(s, torch.float32), dev(a, torch.float32), dev(r, torch.float32), dev(t, torch.uint8),
                            dev(s, torch.float32), idx, 600).cpu().numpy()
    assert losses[-50:, 0].mean() < 0.2 * losses[:20, 0].mean()


@pytest.mark.parametrize("obs_dim", [2, 3])
def test_ddpg_fixed_kernel_tracks_interpreter_over_launches(ssc, obs_dim, monkeypatch):
    """The shape-specialised learner (ddpg_train_fixed.hip: Adam moments in registers, target update fused, padded LDS
    weight image) against the step interpreter over 3 launches x 100 iterations from the same start: parameters,
    targets, moments and step counters must survive the launch boundary through the global arrays, and the two fp32
    summation orders may not drift apart (both are within 5e-6 of the fp64 oracle after 6 steps; here 300)."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    rng = np.random.default_rng(4)
    env = ssc.make("MountainCarContinuous-v0") if obs_dim == 2 else _BoxEnv(obs_dim)
    cap, B = 2000, 64
    s = rng.uniform(-1.0, 1.0, (cap, obs_dim)).astype(np.float32)
    a = rng.uniform(-1, 1, (cap, 1)).astype(np.float32)
    r = (rng.normal(size=cap) * 0.5).astype(np.float32)
    t = (rng.random(cap) < 0.1)
    s2 = (s + rng.normal(size=(cap, obs_dim)) * 0.02).astype(np.float32)
    dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
    data = (dev(s, torch.float32), dev(a, torch.float32), dev(r, torch.float32), dev(t, torch.uint8), dev(s2, torch.float32))
    idx = [dev(np.stack([rng.permutation(cap)[:B] for _ in range(100)]).astype(np.int32), torch.int32) for _ in range(3)]
    out = {}
    for mode in ("fixed", "interpreter"):
        if mode == "interpreter":
            monkeypatch.setenv("SSC_DDPG_INTERPRETER", "1")
        else:
            monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
        agent = DDPG_Baselines_agent(env, None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True,
                                     actor_lr=1e-3, critic_lr=1e-3, gamma=0.99, tau=0.01, batch_size=64, seed=9, training=False)
        losses = [agent.train_on(*data, ix, 100).cpu().numpy() for ix in idx]
        torch.cuda.synchronize()
        out[mode] = dict(actor=agent.actor_flat.cpu().numpy(), critic=agent.critic_flat.cpu().numpy(),
                         ta=agent.target_actor_flat.cpu().numpy(), tc=agent.target_critic_flat.cpu().numpy(),
                         m=agent._adam_critic[0].cpu().numpy(), v=agent._adam_actor[1].cpu().numpy(),
                         t=agent._adam_t.cpu().tolist(), losses=np.concatenate(losses))
    f, g = out["fixed"], out["interpreter"]
    assert f["t"] == g["t"] == [300, 300]
    assert np.isfinite(f["losses"]).all()
    # Adam normalises the step: an element whose gradient is rounding noise moves by +-lr per step whatever the size of
    # the gradient, so a handful of elements may differ by a few steps' worth; everything else agrees to fp32 rounding
    for k in ("actor", "critic", "ta", "tc"):
        diff = np.abs(f[k] - g[k])
        assert np.quantile(diff, 0.99) <= 2e-5 and diff.max() <= 3e-3, (k, np.quantile(diff, 0.99), diff.max())
    assert np.allclose(f["losses"], g["losses"], rtol=2e-3, atol=1e-5)
    assert np.allclose(f["m"], g["m"], rtol=5e-2, atol=1e-6) and np.allclose(f["v"], g["v"], rtol=5e-2, atol=1e-9)


def test_device_replay_ring_matches_oracle(ssc):
    """ssc_replay_append / ssc_replay_sample against the restatement: ring contents bit-exact (wrap-around,
    chunk larger than the ring, last_steps), indices bit-exact."""
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
    for (N, K, cap, last) in [(50, 7, 1000, None), (64, 33, 500, None), (300, 5, 256, None), (40, 20, 333, 6)]:
        env = ssc.VecEnv("MountainCarContinuous-v0", N, seed=11)
        env.reset()
        replay = DeviceReplayBuffer(cap, 2, 1, seed=5)
        ring = dict(s=np.zeros((cap, 2), np.float32), a=np.zeros((cap, 1), np.float32), r=np.zeros(cap, np.float32),
                    t=np.zeros(cap, np.uint8), s2=np.zeros((cap, 2), np.float32))
        count = 0
        for c in range(3):
            chunk = env.rollout(K, ssc.RandomPolicy())
            replay.append_chunk(chunk, reward_scale=0.25, last_steps=last)
            s, a, r, t, s2 = (x.cpu().numpy() for x in chunk.records())
            first = 0 if last is None else (K - last) * N
            count = O.replay_append(ring, count, s[first:], a[first:], r[first:], t[first:].astype(np.uint8), s2[first:],
                                    reward_scale=0.25)
            assert replay.count == count and len(replay) == min(count, cap)
            live = np.arange(min(count, cap)) if count <= cap else np.arange(cap)
            for key in ("s", "a", "r", "t", "s2"):
                assert np.array_equal(getattr(replay, key).cpu().numpy()[live], ring[key][live]), (N, K, cap, c, key)
        idx = replay.sample_indices(9, 64).cpu().numpy()
        assert np.array_equal(idx, O.replay_sample_indices(5, 0, len(replay), 9, 64))
        idx2 = replay.sample_indices(2, 17).cpu().numpy()                  # the batch counter runs on
        assert np.array_equal(idx2, O.replay_sample_indices(5, 9, len(replay), 2, 17))
        # batches of more than 64 (the multi-workgroup learner's sizes): one workgroup per batch, same rule
        drawn = 11
        for nb, B in ((3, 65), (2, 256), (2, 1000), (1, 4096)):
            if B > len(replay):
                continue
            big = replay.sample_indices(nb, B).cpu().numpy()
            assert np.array_equal(big, O.replay_sample_indices(5, drawn, len(replay), nb, B)), (nb, B)
            assert all(len(set(row.tolist())) == B for row in big)
            drawn += nb
    from smartstartcontinuous_amd import _ffi
    for size, nb, B in ((10000, 2, 4096), (4096, 1, 4096), (5000, 3, 2049)):      # up to a dense draw of the largest batch
        out = torch.empty((nb, B), dtype=torch.int32, device="cuda")
        _ffi.check(_ffi.lib().ssc_replay_sample(9, 4, size, nb, B, _ffi.ptr(out), None))
        assert np.array_equal(out.cpu().numpy(), O.replay_sample_indices(9, 4, size, nb, B)), (size, nb, B)
    tiny = DeviceReplayBuffer(64, 2, 1, seed=1)
    with pytest.raises(ValueError):
        tiny.sample_indices(1, 64)


def test_rl_train_vec_ddpg_loop_in_hbm(ssc):
    """rollout (actor + OU) -> device replay -> DDPG iterations, repeated: the learner's parameters move,
    the replay holds what the rollouts produced, the losses are finite, episodes are recorded."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    N, K = 256, 40
    env = ssc.VecEnv("MountainCarContinuous-v0", N, seed=2, max_episode_steps=50)
    env.reset()
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=64, num_train_iterations=5,
                                 actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3,
                                 reward_scale=0.5)
    before = agent.actor_flat.clone()
    summary, losses, replay = ssc.rl_train_vec_ddpg(env, agent, num_chunks=3, chunk_steps=K, replay_capacity=20000, seed=9)
    assert replay.count == 3 * K * N and len(replay) == 20000
    assert len(losses) == 3 and all(l.shape == (5, 2) and torch.isfinite(l).all() for l in losses)
    assert not torch.equal(agent.actor_flat, before)
    assert len(summary.episodes) == 2 * N and all(e[0] == 50 for e in summary.episodes)   # 120 steps, TimeLimit 50
    assert abs(agent.decaying_ou_action_noise.epsilon - 0.99 ** 2) < 1e-12               # one decay per env-generation
    # the newest records in the ring are the last rollout's: rewards scaled, dynamics consistent
    r = replay.r.cpu().numpy(); a = replay.a.cpu().numpy()[:, 0]; t = replay.t.cpu().numpy()
    assert np.allclose(r[t == 0], 0.5 * (-0.1 * a[t == 0] ** 2), atol=1e-6)


def test_rl_train_vec_ddpg_overlapped_rollout(ssc):
    """``overlap=True``: chunk i+1 is rolled on a second stream under a snapshot of the actor while the learner works on
    chunk i.  (1) With both learning rates at zero the weights never move, so the overlapped loop must reproduce the
    synchronous one bit for bit (episodes, replay contents, losses) -- every cross-stream dependency (chunk buffers,
    episode ring, epsilon decay, replay append) is then exercised against a known answer.  (2) With learning on, the
    loop is deterministic (two runs agree bit for bit) and differs from the synchronous one only through the one-chunk
    staleness: chunk 0 (rolled with the initial weights in both modes) is identical, chunk 1 is not."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    N, K = 512, 40

    def run(overlap, lr_scale, chunks=5):
        env = ssc.VecEnv("MountainCarContinuous-v0", N, seed=2, max_episode_steps=50)
        env.reset()
        agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=64, num_train_iterations=7,
                                     actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3,
                                     actor_lr=1e-2 * lr_scale, critic_lr=1e-2 * lr_scale)
        summary, losses, replay = ssc.rl_train_vec_ddpg(env, agent, num_chunks=chunks, chunk_steps=K, replay_capacity=1 << 17,
                                                        seed=9, overlap=overlap)
        torch.cuda.synchronize()
        return (summary, torch.stack(losses).cpu(), replay.s.cpu(), replay.a.cpu(), replay.r.cpu(), replay.t.cpu(),
                agent.actor_flat.cpu(), agent.decaying_ou_action_noise.epsilon, env.stats.cpu())

    a, b = run(False, 0.0), run(True, 0.0)
    # (records of one chunk arrive in the order the waves claimed ring slots: compare as multisets)
    assert sorted(a[0].episodes) == sorted(b[0].episodes) and len(a[0].episodes) == 4 * N
    for x, y in zip(a[1:7], b[1:7]):
        assert torch.equal(x, y)
    assert a[7] == b[7] and torch.equal(a[8], b[8])

    c, d, e = run(True, 1.0), run(True, 1.0), run(False, 1.0)
    assert sorted(c[0].episodes) == sorted(d[0].episodes)
    for x, y in zip(c[1:7], d[1:7]):
        assert torch.equal(x, y)
    assert not torch.equal(c[6], a[6])                                    # the learner did move the actor
    one = K * N                                                           # records of chunk 0 (ring not yet wrapped)
    assert torch.equal(c[3][:one], e[3][:one])                            # chunk 0: the initial weights in both modes
    assert not torch.equal(c[3][one:2 * one], e[3][one:2 * one])          # chunk 1: initial (stale) vs once-trained weights


def _fake_chunk(ssc, K, n, done, step0, base=0.0):
    """A TransitionChunk with recognisable content: obs = (global step, env), obs2 = (global step + 1, env)."""
    chunk = ssc.TransitionChunk(2, K, n, "cuda")
    k = torch.arange(K, dtype=torch.float32, device="cuda")[:, None] + float(step0) + base
    e = torch.arange(n, dtype=torch.float32, device="cuda")[None, :].expand(K, n)
    chunk.obs[0].copy_(k.expand(K, n)); chunk.obs[1].copy_(e)
    chunk.obs2[0].copy_(k.expand(K, n) + 1.0); chunk.obs2[1].copy_(e)
    chunk.act.zero_(); chunk.rew.zero_()
    chunk.done.copy_(torch.as_tensor(done, dtype=torch.uint8, device="cuda"))
    chunk.step0 = step0
    return chunk


def test_device_replay_episode_index_matches_reference_trace(ssc, golden_dir):
    """The device ring's episode index against the REFERENCE ReplayBuffer's own trace (tests/golden/replay_buffer_kats.npz:
    12 episodes of 3-30 steps through a 50-record buffer, generated by running smartstart/RLAgents/replay_buffer.py):
    same first smart-start index, same episodic paths (replay_buffer.py:154-176), same get_all_states (:102-103).  n = 1
    is the reference's setting; the ring is fed in three chunks so that the per-env step count carries across appends."""
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
    g = np.load(f"{golden_dir}/replay_buffer_kats.npz")
    total = int(g["ep_lens"].sum())
    done = np.zeros((total, 1), np.uint8)
    done[np.cumsum(g["ep_lens"]) - 1, 0] = 1                     # an episode ends where the trace starts the next one
    replay = DeviceReplayBuffer(50, 2, 1, seed=1, track_episodes=True, n_envs=1, max_path_len=64)
    k0 = 0
    for K in (100, 100, total - 200):
        replay.append_chunk(_fake_chunk(ssc, K, 1, done[k0:k0 + K], k0))
        k0 += K
    assert len(replay) == 50 and replay.count == total
    eps = replay.ep_steps.cpu().numpy()
    valid = O.smart_start_valid(eps, 50, total, 1)
    assert int(np.argmax(valid)) == int(g["final_first_index"]) and valid[int(g["final_first_index"]):].all()
    for j in range(3):
        path = replay.get_episodic_path_to_buffer_index(int(g[f"path_idx_{j}"])).double().cpu().numpy()
        assert np.array_equal(path, g[f"path_to_{j}"]), j
    with pytest.raises(ValueError):
        replay.get_episodic_path_to_buffer_index(0)             # its episode start was evicted
    assert np.array_equal(replay.get_all_states().double().cpu().numpy(), g["all_states"])
    idx = replay.get_possible_smart_start_indices(10).cpu().numpy()
    assert len(idx) == 10 and len(set(idx.tolist())) == 10 and idx.min() >= int(g["final_first_index"])
    assert np.array_equal(idx, O.smart_start_indices(valid, 10, 1 ^ 0x5353, 0))
    # more requested than valid records exist: every valid index once
    many = replay.get_possible_smart_start_indices(64)
    assert sorted(many.cpu().tolist()) == np.nonzero(valid)[0].tolist()


@pytest.mark.parametrize("n,K,cap,chunks", [(37, 23, 1000, 3), (64, 40, 1500, 4), (300, 9, 4096, 5)])
def test_device_replay_episode_index_many_envs(ssc, n, K, cap, chunks):
    """n interleaved envs with random episode ends: ep_steps, the valid set, the Philox-keyed sample and the recovered
    paths against the restatement; each path must also equal what the HOST ReplayBuffer (the reference's
    bookkeeping, pinned by the reference trace) recovers for the same env fed alone."""
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer, ReplayBuffer
    rng = np.random.default_rng(n)
    replay = DeviceReplayBuffer(cap, 2, 1, seed=9, track_episodes=True, n_envs=n, max_path_len=4096)
    dones, run, k0 = [], None, 0
    for c in range(chunks):
        d = (rng.random((K, n)) < 0.08).astype(np.uint8)
        replay.append_chunk(_fake_chunk(ssc, K, n, d, k0))
        dones.append(d)
        k0 += K
    done = np.concatenate(dones)
    steps, _ = O.replay_episode_steps(done)
    count = chunks * K * n
    size = min(count, cap)
    ring_steps = np.zeros(cap, np.int64)
    rec = np.arange(count - size, count)
    ring_steps[rec % cap] = steps.reshape(-1)[rec]
    assert np.array_equal(replay.ep_steps.cpu().numpy()[rec % cap], ring_steps[rec % cap])
    valid = O.smart_start_valid(ring_steps, cap, count, n)
    got = replay.get_possible_smart_start_indices(200).cpu().numpy()
    want = O.smart_start_indices(valid, 200, 9 ^ 0x5353, 0)
    assert np.array_equal(got, want[want >= 0]) and valid[got].all() and len(set(got.tolist())) == len(got)
    s_ring, s2_ring = replay.s.cpu().numpy(), replay.s2.cpu().numpy()
    for bi in got[:12]:
        path = replay.get_episodic_path_to_buffer_index(int(bi)).cpu().numpy()
        assert np.array_equal(path, O.replay_episode_path(s_ring, s2_ring, ring_steps, cap, count, n, int(bi), 4096))
        # the same env alone through the host buffer (reference semantics): identical path
        r = count - size + int(bi)
        k_rec, e = r // n, r % n
        host, agent = None, object()
        host = ReplayBuffer(agent, 10 ** 6)
        host.start_new_episode(agent)
        for k in range(k_rec + 1):
            host.add(agent, np.array([k, e], float), [0.0], 0.0, bool(done[k, e]), np.array([k + 1, e], float))
            if done[k, e] and k < k_rec:
                host.start_new_episode(agent)
        assert np.array_equal(path, np.asarray(host.get_episodic_path_to_buffer_index(k_rec)))
    # a gap between two appends restarts every env's running episode
    replay.append_chunk(_fake_chunk(ssc, 3, n, np.zeros((3, n), np.uint8), k0 + 50))
    newest = (replay.count - 1 - np.arange(n)) % cap
    assert (replay.ep_steps.cpu().numpy()[newest] == 3).all()


def test_device_smart_start_selection_without_host_replay(ssc):
    """get_smart_start_path (smartexplorationcontinuous.py:223-305) over the device ring: candidates, V(s), KDE, UCB1 and
    the winner's episodic path never leave the GPU; checked against the restatement fed the ring's contents."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
    from smartstartcontinuous_amd.smartstart import device_smart_start_path
    n, K = 128, 60
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=21, max_episode_steps=25)
    one = ssc.SingleEnvView(ssc.VecEnv("MountainCarContinuous-v0", 1, seed=21))
    agent = DDPG_Baselines_agent(one, None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=2)
    replay = DeviceReplayBuffer(5000, 2, 1, seed=4, track_episodes=True, n_envs=n)
    for _ in range(2):
        replay.append_chunk(env.rollout(K, ssc.RandomPolicy()))
    radii = np.array([0.01, 0.002])
    path, chosen = device_smart_start_path(replay, agent, radii, n_ss=300)
    torch.cuda.synchronize()
    count, cap = replay.count, replay.capacity
    size = min(count, cap)
    eps = replay.ep_steps.cpu().numpy()
    valid = O.smart_start_valid(eps, cap, count, n)
    idx = O.smart_start_indices(valid, 300, 4 ^ 0x5353, 0)
    idx = idx[idx >= 0]
    phys = (idx + count - size) % cap
    s_ring, s2_ring = replay.s.cpu().numpy().astype(np.float64), replay.s2.cpu().numpy().astype(np.float64)
    order = (np.arange(size) + count - size) % cap
    all_states = np.concatenate([s_ring[order], s2_ring[(count - 1) % cap][None]])
    cand = s2_ring[phys]
    _cov, wh, norm = O.kde_scott(all_states)
    pdf = O.kde_evaluate(all_states, cand, wh, norm)
    cw = {k: v.cpu().numpy().astype(np.float64) for k, v in agent.critic_weights.items()}
    aw = {k: v.cpu().numpy().astype(np.float64) for k, v in agent.weights.items()}
    values = O.critic_forward(cand, O.actor_forward(cand, **aw), **cw)[:, 0]
    ucb, best = O.smart_start_ucb(values, pdf, size, O.hyperellipsoid_volume(radii))
    # fp32 kernels vs fp64 restatement: the device winner must be (near-)optimal under the restatement's scores
    pos = int(np.nonzero(idx == int(chosen.item()))[0][0])
    assert ucb[pos] >= ucb[best] - 1e-3 * abs(ucb[best])
    want = O.replay_episode_path(replay.s.cpu().numpy(), replay.s2.cpu().numpy(), eps, cap, count, n, int(chosen.item()), 1001)
    assert np.array_equal(path.cpu().numpy(), want) and path.shape[0] >= 2
