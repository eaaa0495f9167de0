"""Where does a wave-step of the fused actor rollout (BASELINE config 3) go?  Times the same launch with parts of the
step body ablated (diagnostic builds, `make -C smartstartcontinuous_amd/csrc actor_abl`; their results are wrong by
design) and the candidate restructurings, one subprocess per library, with the shader clock sampled by the
one-wave probe of tools/clock_probe.hip beside it.

    python tools/exp_actor_abl.py            # parent: runs every variant
"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [("base", "smartstartcontinuous_amd/libssc.so"), ("plain_stores", "tools/_build/libssc_act_plain.so"),
            ("stagger", "tools/_build/libssc_act_stagger.so"), ("plain+stagger", "tools/_build/libssc_act_plainstagger.so"),
            ("abl1_no_tanh_layer", "tools/_build/libssc_act_abl1.so"), ("abl3_no_noise_gen", "tools/_build/libssc_act_abl3.so")]
if os.environ.get("SSC_ABL_ONLY"):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["SSC_ABL_ONLY"].split(",")]


def child(lib, log):
    import smartstartcontinuous_amd._ffi as F
    F.LIB_PATH = os.path.join(ROOT, lib)
    import numpy as np
    import torch
    from smartstartcontinuous_amd import ActorPolicy, TransitionChunk, VecEnv
    from smartstartcontinuous_amd.agents import init_actor_weights
    n, K = 65536, 256
    w = init_actor_weights(2, 64, 32, 1, torch.Generator().manual_seed(1234))
    env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
    env.reset()
    chunk = TransitionChunk(2, K, n, env.device) if log else None
    pd = env.policy_desc(ActorPolicy(w, precision="bf16_mfma", ou_mu=0.4, ou_sigma=0.6, ou_theta=0.15))
    probe = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libclockprobe.so"))
    probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for _ in range(300):
        env.rollout(K, out=chunk, policy_desc=pd, log=log)
    torch.cuda.synchronize()
    reps = 400
    side = torch.cuda.Stream()
    ns = int(reps * 0.0003 / 50e-6) + 400
    pbuf = torch.zeros(2 * ns, dtype=torch.int64, device="cuda")
    with torch.cuda.stream(side):
        probe.clock_probe_launch(ctypes.c_void_p(pbuf.data_ptr()), ns, 12, ctypes.c_void_p(side.cuda_stream))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); env.rollout(K, out=chunk, policy_desc=pd, log=log); b.record()
    torch.cuda.synchronize()
    d = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    p = pbuf.cpu().numpy().reshape(-1, 2)
    p = p[p[:, 0] != 0]
    h = len(p) // 2
    clk = float((p[h, 1] - p[10, 1]) / (p[h, 0] - p[10, 0]) * 100.0)       # MHz while the launches run
    med = float(np.median(d))
    print(json.dumps({"us_median": med, "us_min": float(d.min()), "us_p90": float(np.percentile(d, 90)), "shader_MHz": clk,
                      "cycles_per_wave_step": med * clk / K, "env_steps_per_s": n * K / med * 1e6}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2] == "1")
    else:
        for name, lib in VARIANTS:
            if not os.path.exists(os.path.join(ROOT, lib)):
                print(name, "missing", lib)
                continue
            for log in (1, 0):
                r = subprocess.run([sys.executable, __file__, lib, str(log)], capture_output=True, text=True, timeout=120)
                line = [x for x in r.stdout.splitlines() if x.startswith("{")]
                print("%-22s log=%d  %s" % (name, log, line[0] if line else ("FAILED " + r.stderr[-300:])), flush=True)
