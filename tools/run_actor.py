"""Run the fused actor (bf16 MFMA) rollout a few times (profiling target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import VecEnv, ActorPolicy, TransitionChunk
from smartstartcontinuous_amd.agents import init_actor_weights
n, K = 65536, 256
w = init_actor_weights(2, 64, 32, 1, torch.Generator().manual_seed(1234))
env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
env.reset()
chunk = TransitionChunk(2, K, n, env.device)
pd = env.policy_desc(ActorPolicy(w, precision="bf16_mfma", ou_mu=0.4, ou_sigma=0.6, ou_theta=0.15))
for _ in range(6):
    env.rollout(K, out=chunk, policy_desc=pd)
torch.cuda.synchronize()
