"""Why are the K timed launches of bench.py (bracketed by device synchronisations) slower than the same launches in a
long back-to-back series?  Pattern: settle 300 | 10 x [synchronize, 20 launches] | 400 back to back | 10 x [synchronize,
sleep 2 ms, 20 launches] | 10 x [synchronize, 100 launches].  Prints per-window means and the per-position mean."""
import sys, os, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv

n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
env.reset()
chunks = [TransitionChunk(env.obs_dim, K, n, env.device) for _ in range(2)]
pd = env.policy_desc(RandomPolicy())
cnt = [0]

def launches(m):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(m)]
    return evs

def run(evs):
    for a, b in evs:
        a.record(); env.rollout(K, out=chunks[cnt[0] & 1], policy_desc=pd); b.record(); cnt[0] += 1

def times(evs):
    return [a.elapsed_time(b) * 1e3 for a, b in evs]

e = launches(300); run(e); torch.cuda.synchronize()
t = times(e); print("settle 300: by 50:", " ".join("%.1f" % st.mean(t[j:j + 50]) for j in range(0, 300, 50)))
for label, m, nap in (("sync + 20", 20, 0.0), ("back to back 400", 400, None), ("sync + 2 ms idle + 20", 20, 0.002), ("sync + 100", 100, 0.0),
                      ("sync + 20 again", 20, 0.0)):
    if nap is None:
        e = launches(m); run(e); torch.cuda.synchronize(); t = times(e)
        print("%s: by 50:" % label, " ".join("%.1f" % st.mean(t[j:j + 50]) for j in range(0, m, 50)), "median %.1f" % st.median(t))
        continue
    wins = []
    for rep in range(10):
        e = launches(m)
        torch.cuda.synchronize()
        if nap: time.sleep(nap)
        run(e)
        torch.cuda.synchronize()
        wins.append(times(e))
    print("%s: window means:" % label, " ".join("%.1f" % st.mean(w) for w in wins))
    print("   mean by position:", " ".join("%.0f" % st.mean(w[p] for w in wins) for p in range(0, m, max(1, m // 20))))
