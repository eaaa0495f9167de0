"""Per-phase cycles of the DDPG learner kernels (diagnostic build -DSSC_DDPG_DIAG in tools/_build/libssc_ddpgdiag.so):
the levels of the shape-specialised kernel (ddpg_train_fixed.hip) or, with SSC_DDPG_INTERPRETER=1, the steps of the
interpreter (ddpg_train.hip)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = os.path.join(ROOT, "tools/_build/libssc_ddpgdiag.so")
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
interp = os.environ.get("SSC_DDPG_INTERPRETER", "0") == "1"
rng = np.random.default_rng(0)
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                             lastLayerTanh=True, seed=1, training=False)
cap, n_it = 100000, 200
dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
s = dev(rng.uniform(-1.2, 0.6, (cap, 2)), torch.float32); a = dev(rng.uniform(-1, 1, (cap, 1)), torch.float32)
r = dev(rng.normal(size=cap), torch.float32); t = dev(rng.random(cap) < 0.01, torch.uint8)
idx = torch.randint(0, cap, (n_it, 64), dtype=torch.int32, device="cuda")
rv = F.ReplayView(s.data_ptr(), a.data_ptr(), r.data_ptr(), t.data_ptr(), s.data_ptr(), cap)
KMAX = 44 if interp else 16
out = torch.zeros((n_it, KMAX), dtype=torch.float32, device="cuda")
d = agent.ddpg_desc()
for _ in range(2):
    F.check(agent.lib.ssc_ddpg_train(ctypes.byref(d), ctypes.byref(rv), F.ptr(idx), n_it, F.ptr(out),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
m = np.median(out.cpu().numpy()[20:], axis=0)
if interp:
    names = (["gather"] + ["tgt fwd a1", "tgt fwd c1", "tgt fwd a2", "tgt fwd a3", "tgt fwd c2", "tgt fwd c3 + y target"]
             + ["fwd c1", "fwd a1", "fwd c2", "fwd a2", "fwd c3 + closs", "fwd a3", "bwd c3 (+deriv)", "fwd c2(s,pi)", "bwd c2->dz1 (+relu)",
                "fwd c3(s,pi) + aloss + dq", "bwd c3 (pi)", "bwd c2->da", "bwd a3", "bwd a2",
                "losses+adam cfg", "wgrad c1", "wgrad c2", "wgrad c3", "wgrad a1", "wgrad a2", "wgrad a3", "target update"])
else:
    names = ["L0 gather + adam cfg", "L1 layer 1 x4", "L2 layer 2: 4 contractions (MFMA)", "L3 output layers x3", "L4 action rows onto the critic heads",
             "L5 y, loss terms, dzb2", "L6 dz2, d action, loss sums", "L7 bwd c2 (MFMA), dz2a", "L8 bwd a2 + critic grads/Adam (MFMA)",
             "L9 actor grads/Adam (MFMA)"]
    m[8] = m[8] + m[10] + m[11] + m[12] + m[13]     # L8 carries marks of its own (10..13) in front of its barrier mark
for k, nme in enumerate(names):
    print("%2d %-40s %7.0f cycles" % (k, nme, m[k]))
print("total %.0f cycles per iteration" % m[:len(names)].sum())
if interp:
    print("marks of step %s (cycles since step start): %s" % (os.environ.get("SSC_DIAG_STEP_NAME", "10"), " ".join("%.0f" % x for x in m[32:40])))
if not interp:
    print("inside L8 (thread 0): bwd a2 %.0f | wgrad c2 tile %.0f | Adam x4 %.0f | small element %.0f (each incl. ~150 cycles of mark overhead)" % tuple(m[[10, 11, 12, 13]]))
