"""layer_norm=True -- the default of Actor_Editted / Critic_Editted (models_editted.py:23, 45-46, 50-51, 85-86, 91-92; no
shipped run turns it on): forward kernels, the fused fp32 rollout policy and the multi-workgroup learner against the fp64
oracle (oracle.actor_forward / critic_forward(layer_norm=...), oracle.ddpg_train_step -- itself checked against torch
autograd through torch.nn.functional.layer_norm in tests/test_oracle_networks.py) at the tolerances of the networks
without LayerNorm."""
import numpy as np
import pytest

from oracle import ssc_oracle as O
from tests.test_gpu_agents import _BoxEnv, _ddpg_kernel_vs_oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def _agent(ssc, obs_dim, h1, h2, llt, seed=3, ch1=None, ch2=None):
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    env = ssc.make("MountainCarContinuous-v0") if obs_dim == 2 else _BoxEnv(obs_dim)
    agent = DDPG_Baselines_agent(env, None, actor_h1=h1, actor_h2=h2, critic_h1=ch1 or h1, critic_h2=ch2 or h2, lastLayerTanh=llt,
                                 layer_norm=True, seed=seed, training=False)
    rng = np.random.default_rng(seed)
    # gamma / beta away from their initial 1 / 0, output layers away from 3e-3
    perturb = lambda w: {k: v.cpu().numpy() + (0.1 * rng.normal(size=tuple(v.shape))).astype(np.float32) for k, v in w.items()}
    aw, cw = perturb(agent.weights), perturb(agent.critic_weights)
    agent.set_weights(aw)
    agent.set_critic_weights(cw)
    return agent, aw, cw


def _ln(w):
    return ((w["ln1_g"], w["ln1_b"]), (w["ln2_g"], w["ln2_b"]))


def _core(w):
    return {k: w[k] for k in O.ACTOR_KEYS}


@pytest.mark.parametrize("obs_dim,h1,h2,llt", [(2, 64, 32, True), (2, 64, 64, False), (3, 200, 100, True), (8, 24, 12, True)])
def test_layer_norm_actor_and_critic_forward(ssc, obs_dim, h1, h2, llt):
    agent, aw, cw = _agent(ssc, obs_dim, h1, h2, llt)
    assert list(agent.weights) == list(O.LN_KEYS) and agent.precision == "f32"
    assert agent.actor_flat.numel() == O.flatten_params(aw).size
    rng = np.random.default_rng(1)
    for m in (1, 7, 1000):                                      # the row kernel (m <= 32) and the row-per-lane kernel
        obs = rng.normal(size=(m, obs_dim)).astype(np.float32) * 2.0
        if obs_dim >= 3:
            obs[:, 2] *= 4.0                                    # beyond observation_range: the clip acts
        act = rng.uniform(-1, 1, (m, 1)).astype(np.float32)
        ref = O.actor_forward(obs, **_core(aw), last_layer_tanh=llt, layer_norm=_ln(aw), obs_clip=5.0)
        got = agent.actor(obs).cpu().numpy()
        assert np.max(np.abs(got - ref)) <= 1e-5, (m, np.max(np.abs(got - ref)))
        refq = O.critic_forward(obs, act, **_core(cw), last_layer_tanh=llt, layer_norm=_ln(cw), obs_clip=5.0)
        gotq = agent.critic(obs, act).cpu().numpy()
        assert np.max(np.abs(gotq - refq[:, 0])) <= 2e-5 * max(1.0, np.abs(refq).max())
    # LayerNorm does something: the same weights without it give another action
    plain = O.actor_forward(obs, **_core(aw), last_layer_tanh=llt, obs_clip=5.0)
    assert np.max(np.abs(plain - ref)) > 1e-3
    # the MFMA kernels do not carry LayerNorm: asked for by hand, the call says so
    from smartstartcontinuous_amd import _ffi
    agent._desc.precision = _ffi.SSC_PREC_BF16_MFMA
    with pytest.raises(_ffi.SscError) as ei:
        agent.actor(obs)
    assert ei.value.code == _ffi.SSC_EUNSUPPORTED and "LayerNorm" in str(ei.value)


@pytest.mark.parametrize("env_name,h2", [("MountainCarContinuous-v0", 32), ("MountainCarContinuous-v0", 64), ("Pendulum-v1", 32)])
def test_layer_norm_actor_in_the_fused_rollout(ssc, env_name, h2):
    """rollout(K, actor policy) with a LayerNorm actor (fp32 policy, 64-32 and the class-default 64-64): every logged action is
    the oracle's actor output on the logged observation (noise off), scaled to the env's action bounds."""
    obs_dim = 2 if env_name.startswith("Mountain") else 3
    agent, aw, _ = _agent(ssc, obs_dim, 64, h2, True, seed=8)
    agent.decaying_ou_action_noise.epsilon = 0.0
    env = ssc.VecEnv(env_name, 300, seed=4, max_episode_steps=25)
    pol = agent.as_policy()
    assert pol.precision == "f32" and "ln1_g" in pol.weights
    chunk = env.rollout(40, pol)
    torch.cuda.synchronize()
    o = chunk.obs.cpu().numpy().transpose(1, 2, 0).reshape(-1, obs_dim)
    ref = np.clip(O.actor_forward(o, **_core(aw), last_layer_tanh=True, layer_norm=_ln(aw), obs_clip=5.0)[:, 0], -1, 1)
    hi = float(env.action_space.high[0])
    expect = hi * np.clip(hi * ref, -1, 1)                      # scale(scale(a)) (DDPG_Baselines_agent.py:236-240) for bounds +-hi
    assert np.max(np.abs(chunk.act.cpu().numpy().reshape(-1) - expect)) <= 2e-5 * hi
    assert float(chunk.done.sum()) >= 300                       # episodes ended and restarted inside the chunk


@pytest.mark.parametrize("obs_dim,h1,h2,B,llts", [(2, 64, 32, 64, (True, False)), (2, 64, 64, 256, (True,)), (3, 128, 64, 64, (True,)),
                                                   (3, 128, 64, 1024, (True,)), (8, 37, 19, 77, (False,))])
def test_layer_norm_ddpg_train_kernel_vs_oracle(ssc, obs_dim, h1, h2, B, llts, monkeypatch):
    """ssc_ddpg_train_ws with ssc_ddpg_desc.layer_norm: parameters in TF order [W1|b1|beta1|gamma1|W2|b2|beta2|gamma2|W3|b3],
    LayerNorm forward in all four networks, backward through the critic (both losses) and the actor, gamma / beta updated by
    MpiAdam and tracked by the targets -- parameters, targets, moments and losses after 6 iterations against the fp64 oracle."""
    monkeypatch.delenv("SSC_DDPG_WIDE", raising=False)
    monkeypatch.delenv("SSC_DDPG_INTERPRETER", raising=False)
    _ddpg_kernel_vs_oracle(ssc, obs_dim, h1, h2, B=B, llts=llts, cap=3000, layer_norm=True)


def test_layer_norm_learner_says_when_a_tile_does_not_fit(ssc):
    """The x-hat rows the LayerNorm backward pass keeps make a 16-row tile of the 200-100 networks 181 KB: refused with the
    byte count (the shapes with LayerNorm that fit go up to 128-64; the reference's LayerNorm default is 64-64)."""
    from smartstartcontinuous_amd import _ffi
    agent, _, _ = _agent(ssc, 2, 200, 100, True)
    z = torch.zeros(64, 2, device="cuda")
    idx = torch.zeros((1, 64), dtype=torch.int32, device="cuda")
    with pytest.raises(_ffi.SscError) as ei:
        agent.train_on(z, z[:, :1].contiguous(), z[:, 0].contiguous(), torch.zeros(64, dtype=torch.uint8, device="cuda"), z, idx, 1)
    assert ei.value.code == _ffi.SSC_EUNSUPPORTED and "LDS" in str(ei.value)


def test_layer_norm_agent_runs_rltrain_and_the_vector_loop(ssc):
    """DDPG_Baselines_agent(layer_norm=True) no longer raises: the scalar rlTrain loop (get_action / observe / train) and the
    vectorised actor-learner loop both run on it; the LayerNorm parameters move."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    env = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(1.0, max_episode_steps=40)
    agent = DDPG_Baselines_agent(env, None, batch_size=64, num_train_iterations=2, num_steps_before_train=1, actor_h1=64, actor_h2=64,
                                 critic_h1=64, critic_h2=64, layer_norm=True, lastLayerTanh=False, seed=2)
    g0 = agent.weights["ln1_g"].clone()
    np.random.seed(0)
    summary = ssc.rlTrain(agent, env, print_results=False, print_steps=False, num_episodes=3, max_steps=1000)
    assert [e[0] for e in summary.episodes] == [40, 40, 40] and int(agent._adam_t[0].item()) > 20
    assert not torch.equal(agent.weights["ln1_g"], g0) and bool(torch.isfinite(agent.actor_flat).all())
    venv = ssc.VecEnv("MountainCarContinuous-v0", 512, seed=1, max_episode_steps=50)
    venv.reset()
    s, losses, replay = ssc.rl_train_vec_ddpg(venv, agent, num_chunks=4, chunk_steps=32, replay_capacity=1 << 15, train_iters=3)
    assert len(losses) == 4 and all(bool(torch.isfinite(l).all()) for l in losses) and len(s.episodes) >= 512
