"""Helpers shared by the -m gpu tests."""
import numpy as np


def mc_log(chunk):
    return dict(s_pos=chunk.obs[0].cpu().numpy(), s_vel=chunk.obs[1].cpu().numpy(), act=chunk.act.cpu().numpy(),
                rew=chunk.rew.cpu().numpy(), done=chunk.done.cpu().numpy(),
                s2_pos=chunk.obs2[0].cpu().numpy(), s2_vel=chunk.obs2[1].cpu().numpy())


def assert_replay_clean(res, tol_pos=2.4e-7, tol_vel=1e-8, tol_rew=1e-4):
    assert res["start_mismatch"] == 0, res
    assert res["act_mismatch"] == 0, res
    assert res["done_mismatch"] == 0, res
    assert res["continuity_mismatch"] == 0, res
    assert res["reset_mismatch"] == 0, res
    assert res["max_dpos"] <= tol_pos, res
    assert res["max_dvel"] <= tol_vel, res
    assert res["max_drew"] <= tol_rew, res


def actor_weights(obs_dim=2, h1=64, h2=32, seed=1234, w3_scale=3e-3):
    """glorot-uniform / U(+-3e-3) init like tf.layers.dense in models_editted.py:44-59."""
    rng = np.random.default_rng(seed)

    def glorot(i, o):
        lim = np.sqrt(6.0 / (i + o))
        return rng.uniform(-lim, lim, size=(i, o)).astype(np.float32)
    return dict(W1=glorot(obs_dim, h1), b1=(0.1 * rng.normal(size=h1)).astype(np.float32),
                W2=glorot(h1, h2), b2=(0.1 * rng.normal(size=h2)).astype(np.float32),
                W3=rng.uniform(-w3_scale, w3_scale, size=(h2, 1)).astype(np.float32),
                b3=(0.05 * rng.normal(size=1)).astype(np.float32))
