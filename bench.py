#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused random-policy rollout on 65 536 parallel
MountainCarContinuous envs per GPU (BASELINE.json metric; config[1] at N=1, config[4] sharding
for N>1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      # no launcher: starts the N ranks itself (child processes)

One "step" = one ssc_rollout launch: every env of the rank advances CHUNK (=1024) env-steps
and the whole transition log (25 B per env-step, SoA [K][n]) is written to HBM.  Inputs
(env state) are resident in HBM before the timed region.  With --gpus N each rank owns
65 536 envs of one global id space (weak scaling) and, after every chunk, takes part in the
bounded RCCL transition gather + 4-scalar stats all-reduce described in DESIGN.md.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ENVS_PER_GPU = 65536
CHUNK = 1024
BYTES_PER_STEP = 25           # s[2] a r t(u8) s2[2]
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
GATHER_RECORDS = 1 << 20      # bounded gather: last G steps with G * N_total <= 2^20 records


def cpu_baseline(budget_s=24.0):
    """The CPU oracle timed on this box's host cores on a bounded sample of the same workload (kind "port": the
    reference's Python/TF cannot travel).  SURVEY.md section 8(d): `value` = P independent scalar
    rlTrain processes, P = the cores this process may use (stated as `cores`); the numpy-vectorised and C
    restatements on 1 and P cores ride along as the "strong CPU" lines."""
    from oracle import cpu_baseline as cb
    return cb.run(budget_s, N_ENVS_PER_GPU)


def dist_stats(ms):
    """min / median / p90 / mean of a list of per-launch durations (ms)."""
    v = sorted(ms)
    n = len(v)
    return {"n": n, "min": v[0], "median": v[n // 2], "p90": v[min(n - 1, int(0.9 * n))], "max": v[-1], "mean": sum(v) / n}


def source_sha():
    """sha256 over the sources of the headline kernel; profiles/*/traffic.json carries the same stamp, so a PMC
    figure measured on other code is reported as stale instead of silently riding along."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rollout.hip", "ssc_device.h"):
        h.update(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def source_sha_of(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


C3_SOURCES = ("rollout.hip", "ssc_device.h", "actor_device.h")


def profiled_valu_issue():
    """The roof that actually binds BASELINE configs[2] (fused actor rollout): VALU / transcendental ISSUE at one wave per
    SIMD.  `frac` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the rollout kernel (share of the resident wave's cycles in which
    a VALU instruction was issuing), from the committed PMC pass of this same command (profiles/*/valu_issue.json, written
    by tools/summarize_r03.py); like `traffic` it is reported only for the kernel sources it was measured on."""
    import glob
    sha, best = source_sha_of(C3_SOURCES), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "valu_issue.json")) +
                    glob.glob(os.path.join(ROOT, "profiles", "*", "*", "valu_issue.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        rel = os.path.relpath(f, ROOT)
        if d.get("source_sha") == sha:
            best = dict(d, source=rel)
        elif best is None or "stale" in best:
            best = {"stale": "%s was measured on kernel sources %s, this build is %s" % (rel, d.get("source_sha"), sha)}
    return best


C4_SOURCES = ("dyn_mfma.hip",)


def profiled_pipe_busy():
    """Matrix-pipe busy fraction of the forward-simulation kernel (BASELINE configs[3]): SQ_VALU_MFMA_BUSY_CYCLES per SIMD over
    the launch's own length (SQ_BUSY_CYCLES per shader engine), both from the committed PMC pass of this same command
    (profiles/*/c4/mfma_busy.json, written by tools/summarize_c4_busy.py) -- clock-independent, and reported only for the
    kernel source it was measured on."""
    import glob
    sha, best = source_sha_of(C4_SOURCES), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "c4", "mfma_busy.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        rel = os.path.relpath(f, ROOT)
        if d.get("source_sha") == sha:
            best = dict(d, source=rel)
        elif best is None or "stale" in best:
            best = {"stale": "%s was measured on kernel source %s, this build is %s" % (rel, d.get("source_sha"), sha)}
    return best


def bench_config3(args, torch, emit=True):
    """BASELINE configs[2]: 65 536 MountainCar envs + DDPG actor 64-32 (bf16 MFMA) + OU noise, fused."""
    import numpy as np

    from smartstartcontinuous_amd import ActorPolicy, TransitionChunk, VecEnv
    from smartstartcontinuous_amd.agents import init_actor_weights
    # the env stream of SURVEY.md section 8(d) config 2 ("same env stream"): K = --chunk env-steps per launch (1024).  Rounds 1-2 used
    # 256-step launches here; every launch boundary costs ~20 us of stream time, which is 9 % of a 256-step launch of this kernel
    n, K = args.envs_per_gpu, args.chunk
    w = init_actor_weights(2, 64, 32, 1, torch.Generator().manual_seed(1234))
    env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
    env.reset()
    chunk = TransitionChunk(2, K, n, env.device)
    pd = env.policy_desc(ActorPolicy(w, precision="bf16_mfma", ou_mu=0.4, ou_sigma=0.6, ou_theta=0.15))
    for _ in range(args.warmup + args.settle_launches):     # settling: see main()
        env.rollout(K, out=chunk, policy_desc=pd)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_first.record()                   # one bracketing pair in the timed region, per-launch pairs in a second window: see main()
    for _ in range(args.steps):
        env.rollout(K, out=chunk, policy_desc=pd)
    e_last.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kms = e_first.elapsed_time(e_last) / args.steps
    for a, b in evs:
        a.record(); env.rollout(K, out=chunk, policy_desc=pd); b.record()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in evs]
    rate = n * K * args.steps / el
    mfma_flops = 2.0 * 64 * 32 + 2.0 * 2 * 64          # hidden GEMM (bf16 MFMA) + layer 1 (fp32 MFMA) per env-step
    res = {"metric": "env-steps/sec, 65 536 MountainCar envs + DDPG actor 64-32 fwd (bf16 MFMA) + OU noise", "value": rate,
           "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "settle_launches": args.settle_launches,
           "ms_per_step": el / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (hidden GEMM), f32 elsewhere",
           "data": "synthetic", "config": {"workload": "BASELINE configs[2]: MountainCarContinuous-v0, %d envs, actor 64-32 "
                                          "lastLayerTanh, OU mu0.4 sigma0.6 theta0.15, %d env-steps per launch, full log" % (n, K)},
           "roofline": {"bound": "hbm", "achieved": 25.0 * n * K / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": 25.0 * n * K / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kms,
                        "kernel_ms_dist": dist_stats(per),
                        "mfma_tflops": mfma_flops * n * K / (kms * 1e-3) / 1e12,
                        "mfma_frac_of_bf16_peak": mfma_flops * n * K / (kms * 1e-3) / 1e12 / 2500.0,
                        "note": "neither contract roof binds: the kernel is VALU/transcendental-ISSUE-bound (33 tanh + 96 "
                                "relu/convert + OU Box-Muller per env-step); reported against the HBM roof of its 25 B/step "
                                "log (the only HBM traffic), with the MFMA-eligible rate beside it"}}
    vi = profiled_valu_issue()
    if vi is not None:
        res["roofline"]["valu_issue"] = vi
    if not args.no_cpu_baseline:
        from oracle import ssc_oracle as O               # the checker, timed as the CPU baseline only
        wn = {k: v.numpy() for k, v in w.items()}
        obs = np.random.default_rng(0).uniform(-1, 1, (n, 2)).astype(np.float32)
        pos, vel = obs[:, 0].astype(np.float64) * 0.5 - 0.5, obs[:, 1].astype(np.float64) * 0.05
        x = np.zeros(n)
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < args.cpu_budget / 2:
            a = O.actor_forward(np.stack([pos, vel], 1), **wn)[:, 0]
            x = O.ou_step(x, np.random.standard_normal(n), 0.4, 0.6)
            pos, vel, r, d = O.mc_step(pos, vel, O.ddpg_action(a, x, 1.0))
            k += 1
        res["cpu_baseline"] = {"value": n * k / (time.perf_counter() - t0), "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "numpy fp64 oracle (actor + OU + step), %d envs x %d steps" % (n, k)}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_config4(args, torch, emit=True):
    """BASELINE configs[3]: NND_MB dynamics MLP 2x500 (Pendulum: in 4, out 3), M = 65 536 rows, H = 4:
    MPC sampling + forward simulation (bf16 MFMA) + trajectory scoring."""
    import numpy as np

    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd.agents import init_dynamics_weights
    from smartstartcontinuous_amd import RandomPolicy, VecEnv
    from smartstartcontinuous_amd import numerical as num
    P, N, H, d, a = 16, 4096, 4, 3, 1
    M = P * N
    # data set exactly as the reference builds it (NND_MB_agent.py:75-76,230-242): 25 random-policy
    # rollouts of 333 steps -- produced by the Pendulum rollout kernel; z-score statistics :302-315
    from smartstartcontinuous_amd import collect_samples as cs
    denv = VecEnv("Pendulum-v0", 25, seed=1234)
    dchunk = denv.rollout(333, RandomPolicy())
    ts = cs.dataset_from_chunk(dchunk)                                   # data_manipulation.py:58-88 on the device
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    Ws, bs = init_dynamics_weights(d + a, d, 2, 500, torch.Generator().manual_seed(1234))
    model = nav.DynamicsModel(Ws, bs, norm, d, a, precision="bf16_mfma")
    # waypoints = a recorded 200-state path (env 0 of the data set), radii / distances_left as in
    # start_new_episode_plan (NND_MB_agent.py:375-423, no shortcutting)
    path = dchunk.obs[:, :200, 0].t().double().cpu().numpy()
    stds, means = num.path_deltas_stds_and_means_per_dim(path)
    rad = num.radii_calc(means, stds, 1, 1, 1)
    dist = num.elliptical_euclidean_distance_function_generator(rad)
    wps = [path] * P
    radii = [rad] * P
    lefts = [num.distances_left(path, dist)] * P
    ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P, theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5)
    s0 = torch.as_tensor(np.repeat(np.stack([w[0] for w in wps]), N, axis=0), dtype=torch.float32, device="cuda")
    s0_p = torch.as_tensor(np.stack([w[0] for w in wps]), dtype=torch.float32, device="cuda")     # one start state per problem

    def _ws_bytes(P_, N_, H_):
        from smartstartcontinuous_amd import _ffi
        return _ffi.lib().ssc_mpc_score_workspace_bytes(P_, N_, H_)
    S = torch.empty((H + 1, M, d), device="cuda")
    rng = np.random.default_rng(0)

    sel_out = dict(scores=torch.empty(M, device="cuda"), best=torch.empty(P, dtype=torch.int32, device="cuda"),
                   best_score=torch.empty(P, device="cuda"), action=torch.empty((P, a), device="cuda"),
                   ws=torch.empty(_ws_bytes(P, N, H), dtype=torch.uint8, device="cuda"))

    def step(t):
        # THREE launches per MPC step: the forward simulation draws its own candidate sequences (no sample launch, no
        # [M][H][a] matrix), scoring pass A, scoring pass B whose last block also selects the action
        sp = nav.mpc_sampling(N, [-2.0], [2.0], 1234, 0, t)
        model.do_forward_sim_sampled(s0_p, sp, M, H, out=S)
        return nav.mpc_score_select(ps, S, sampling=sp, act_dim=a, noise_amount=0.005, seed=1234, problem_id0=0, t=t,
                                    want_path=False, out=sel_out)[2]

    def sim_only(A):
        model.do_forward_sim(s0, A, out=S)
    for t in range(args.warmup + args.settle_launches):     # settling: see main()
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        step(t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # kernel leg of the roofline: the forward-sim launches alone, replayed from a HIP graph so that the event
    # interval holds back-to-back kernels and not the Python/ctypes launch path (~40 us per call, the same
    # order as the kernel at H = 4)
    A = nav.mpc_sample_actions(P, N, H, [-2.0], [2.0], 1234, 0, 0)
    reps = 10
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sim_only(A)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            sim_only(A)
    graph.replay()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for e0, e1 in evs:
        e0.record(); graph.replay(); e1.record()
    torch.cuda.synchronize()
    kms_b2b = sum(x.elapsed_time(y) for x, y in evs) / (args.steps * reps)
    # ... and the same kernel where the product runs it: inside the MPC step (simulate, score A, score B), one event pair
    # around the simulation launch of every step of a second, untimed pass over the same steps.  This is the figure the
    # roofline uses (it is what rocprofv3 averages for this kernel over the step loop); ten simulations back to back
    # run 7-10 % slower -- the kernel is power-limited and the two scoring launches are its breathing space.
    sim_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for t, (e0, e1) in enumerate(sim_evs):
        sp = nav.mpc_sampling(N, [-2.0], [2.0], 1234, 0, t)
        e0.record(); model.do_forward_sim_sampled(s0_p, sp, M, H, out=S); e1.record()
        nav.mpc_score_select(ps, S, sampling=sp, act_dim=a, noise_amount=0.005, seed=1234, problem_id0=0, t=t, want_path=False, out=sel_out)
    torch.cuda.synchronize()
    kms = sum(x.elapsed_time(y) for x, y in sim_evs) / args.steps
    flop_row = 2.0 * ((d + a) * 500 + 500 * 500 + 500 * d)         # 507 000, SURVEY 8d
    res = {"metric": "row-steps/sec, NND_MB dynamics MLP 2x500 forward sim + MPC scoring, 65 536 rows", "value": M * H * args.steps / el,
           "unit": "row-steps/s", "env_steps_per_s": P * args.steps / el,
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "settle_launches": args.settle_launches,
           "ms_per_step": el / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (all layers on bf16 MFMA, fp32 accumulate; layer 1 split into bf16 head + residual)",
           "data": "synthetic", "config": {"workload": "BASELINE configs[3]: Pendulum dims (in 4, out 3), num_fc_layers 2, depth 500, "
                                          "%d MPC problems x %d samples = %d rows, horizon %d; z-score stats from 25x333 Pendulum random rollouts, "
                                          "200-state recorded path as waypoints; forward sim (in-kernel sampling) + score + select = 3 launches per MPC step" % (P, N, M, H)},
           "roofline": {"bound": "mfma", "achieved": flop_row * M * H / (kms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": flop_row * M * H / (kms * 1e-3) / 1e12 / 2500.0, "traffic": None, "kernel_ms": kms,
                        "kernel": "ssc::dyn_mfma_sim_kernel<16,2,true,4> (weight image prepared once; one event pair around the simulation launch of every MPC step)",
                        "kernel_ms_back_to_back": kms_b2b, "kernel_ms_back_to_back_note": "HIP-graph replay of 10 simulation launches with nothing in between",
                        "algorithmic_flop_per_launch": flop_row * M * H}}
    pb = profiled_pipe_busy()
    if pb is not None:
        res["roofline"]["pipe_busy"] = pb
    # the box's own matrix ceiling at this shape ("confirm with a hipBLASLt [65536,512] x [512,512] probe"):
    # a plain bf16 library GEMM through torch.matmul, median of 10 launches -- a measurement aid, not part of the product path
    try:
        ga = torch.randn((M, 512), device="cuda", dtype=torch.bfloat16)
        gb = torch.randn((512, 512), device="cuda", dtype=torch.bfloat16)
        gc = torch.empty((M, 512), device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            torch.matmul(ga, gb, out=gc)
        ge = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in ge:
            e0.record(); torch.matmul(ga, gb, out=gc); e1.record()
        torch.cuda.synchronize()
        gms = sorted(x.elapsed_time(y) for x, y in ge)[len(ge) // 2]
        gtf = 2.0 * M * 512 * 512 / (gms * 1e-3) / 1e12
        res["roofline"]["gemm_probe"] = {"achieved": gtf, "unit": "TFLOP/s", "ms": gms,
                                         "kernel": "torch.matmul bf16 [%d,512] x [512,512] (library GEMM)" % M,
                                         "sim_over_probe": res["roofline"]["achieved"] / gtf}
        del ga, gb, gc
    except Exception as e:           # noqa: BLE001
        res["roofline"]["gemm_probe"] = {"error": repr(e)}
    # SURVEY.md section 8(d) names H = 20 (the class default horizon, NND_MB_agent.py:62) beside H = 4: the same launch, 20 steps
    if getattr(args, "no_h20", False):
        if emit:
            print(json.dumps(res), flush=True)
        return res
    H20 = 20
    S20 = torch.empty((H20 + 1, M, d), device="cuda")
    sp20 = nav.mpc_sampling(N, [-2.0], [2.0], 1234, 0, 0)
    def timed20(n=10):
        e20 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for e0, e1 in e20:
            e0.record(); model.do_forward_sim_sampled(s0_p, sp20, M, H20, out=S20); e1.record()
        torch.cuda.synchronize()
        return sorted(x.elapsed_time(y) for x, y in e20)[n // 2]
    for _ in range(3):
        model.do_forward_sim_sampled(s0_p, sp20, M, H20, out=S20)
    k20_cold = timed20()        # rounds 1-3 reported THIS: 3 + 10 launches = 6 ms, all of it inside the post-idle clock transient
    if args.settle_launches > 0:                                  # ~0.4 s of the same launch, like the headline's settle phase
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.4:
            for _ in range(20):
                model.do_forward_sim_sampled(s0_p, sp20, M, H20, out=S20)
            torch.cuda.synchronize()
    k20 = timed20()
    res["h20"] = {"kernel_ms": k20, "row_steps_per_s": M * H20 / (k20 * 1e-3), "achieved": flop_row * M * H20 / (k20 * 1e-3) / 1e12,
                  "unit": "TFLOP/s", "frac": flop_row * M * H20 / (k20 * 1e-3) / 1e12 / 2500.0,
                  "after_warmup_only": {"kernel_ms": k20_cold, "frac": flop_row * M * H20 / (k20_cold * 1e-3) / 1e12 / 2500.0},
                  "note": "the forward simulation alone at horizon 20, median of 10 launches back to back after 0.4 s of the same launch "
                          "(`after_warmup_only`: the same 10 launches right after 3 warm-ups -- 6 ms that lie inside the post-idle clock "
                          "transient, DESIGN.md section 6; in-kernel stamps: profiles/r04/c4/clk_h20.txt, 42.2 k cycles per step at H = 4 and H = 20 alike)"}
    del S20
    if not args.no_cpu_baseline:
        from oracle import ssc_oracle as O               # the checker, timed as the CPU baseline only
        Wn, bn = [w.numpy() for w in Ws], [b.numpy() for b in bs]
        m_cpu = 2048
        An = rng.uniform(-2, 2, (m_cpu, H, a))
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < args.cpu_budget / 2:
            Sn = O.dyn_forward_sim(np.zeros((m_cpu, d)), An, norm, Wn, bn)
            O.mpc_scores_add_delta(Sn, wps[0], lefts[0], radii[0], 0)
            k += 1
        res["cpu_baseline"] = {"value": m_cpu * H * k / (time.perf_counter() - t0), "unit": "row-steps/s", "cores": 1, "kind": "port",
                               "sample": "numpy fp64 oracle forward sim + scoring, %d rows x H=%d x %d repeats" % (m_cpu, H, k)}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_config4_envs(args, torch, emit=True, P=N_ENVS_PER_GPU, N=16, H=4, K=40):
    """BASELINE configs[3] read literally: Pendulum-v1, 65 536 ENVS, every env navigated by its own MPC problem
    (NND_MB 2x500, P = 65 536 problems x N candidate sequences = P*N simulated rows per env-step) through
    VecEnv.rollout(K, MpcPolicy): per env-step 3 graph-replayed launches (forward simulation with in-kernel sampling,
    one-launch scoring, fused action / env.step / log / waypoint bookkeeping)."""
    import numpy as np

    from smartstartcontinuous_amd import MpcPolicy, RandomPolicy, TransitionChunk, VecEnv
    from smartstartcontinuous_amd import collect_samples as cs
    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd import numerical as num
    from smartstartcontinuous_amd.agents import init_dynamics_weights
    d, a = 3, 1
    denv = VecEnv("Pendulum-v1", 64, seed=1234)
    dchunk = denv.rollout(200, RandomPolicy())
    ts = cs.dataset_from_chunk(dchunk)
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    Ws, bs = init_dynamics_weights(d + a, d, 2, 500, torch.Generator().manual_seed(1234))
    model = nav.DynamicsModel(Ws, bs, norm, d, a, precision="bf16_mfma")
    # plans: the 64 recorded 200-state paths, one per env (env p follows path p mod 64)
    paths = dchunk.obs[:, :200, :].permute(2, 1, 0).double().cpu().numpy()           # [64, 200, 3]
    radii, lefts = [], []
    for pth in paths:
        stds, means = num.path_deltas_stds_and_means_per_dim(pth)
        r = num.radii_calc(means, stds, 1, 1, 1)
        radii.append(r)
        lefts.append(num.distances_left(pth, num.elliptical_euclidean_distance_function_generator(r)))
    which = np.arange(P) % 64
    W = paths.shape[1]
    ps = nav.MpcProblemSet.from_packed(paths[which].reshape(P * W, d), np.stack(lefts)[which].reshape(-1),
                                       (np.arange(P + 1) * W).astype(np.int32), np.stack(radii)[which], np.zeros(P, np.int32))
    env = VecEnv("Pendulum-v1", P, seed=1234)
    env.reset()
    start = paths[which, 0]
    env.s0.copy_(torch.as_tensor(np.arctan2(start[:, 1], start[:, 0]), dtype=torch.float32))
    env.s1.copy_(torch.as_tensor(start[:, 2], dtype=torch.float32))
    batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, action_low=[-2.0], action_high=[2.0], seed=1234)
    pol = MpcPolicy(batch)
    chunk = TransitionChunk(d, K, P, env.device)
    env.rollout(K, pol, out=chunk)                         # captures the graph, warms up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, args.steps // 4)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        env.rollout(K, pol, out=chunk)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms_step = el / (reps * K) * 1e3
    flop_row = 2.0 * ((d + a) * 500 + 500 * 500 + 500 * d)
    # the simulation launch of this leg alone (P * N rows = 4096 row tiles on 256 CUs: the WALKING instantiation of the
    # kernel, round 4), settled: median of 20 launches after 0.4 s of the same launch
    sp = nav.mpc_sampling(N, [-2.0], [2.0], 1234, 0, 0)
    s0p = torch.as_tensor(start, dtype=torch.float32, device="cuda").contiguous()
    Ssim = torch.empty((H + 1, P * N, d), device="cuda")
    sim = lambda: model.do_forward_sim_sampled(s0p, sp, P * N, H, out=Ssim)
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < (0.4 if args.settle_launches > 0 else 0.0):
        for _ in range(10):
            sim()
        torch.cuda.synchronize()
    sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for x, y in sev:
        x.record(); sim(); y.record()
    torch.cuda.synchronize()
    sim_ms = sorted(x.elapsed_time(y) for x, y in sev)[len(sev) // 2]
    del Ssim
    res = {"metric": "env-steps/sec, Pendulum-v1 + NND_MB 2x500 MPC navigation, 65 536 envs (one navigator per env)",
           "value": P / (ms_step * 1e-3), "unit": "env-steps/s", "row_steps_per_s": P * N * H / (ms_step * 1e-3),
           "n_gpus": 1, "steps": reps * K, "ms_per_step": ms_step, "gpu_ms_per_step": e0.elapsed_time(e1) / (reps * K),
           "higher_is_better": True, "dtype": "bf16 (MFMA), fp32 accumulate", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3] as written: Pendulum-v1, %d envs x %d candidates = %d rows, horizon %d, "
                                  "num_fc_layers 2, depth 500; rollout(K=%d, 'mpc'), 3 launches per env-step from a HIP graph; "
                                  "200-waypoint plans" % (P, N, P * N, H, K)},
           "roofline": {"bound": "mfma", "achieved": flop_row * P * N * H / (ms_step * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": flop_row * P * N * H / (ms_step * 1e-3) / 1e12 / 2500.0, "traffic": None,
                        "note": "whole env-step (simulate + score + act/step/log) against the bf16 MFMA roof of its simulated rows",
                        "sim_kernel": {"kernel_ms": sim_ms, "achieved": flop_row * P * N * H / (sim_ms * 1e-3) / 1e12,
                                       "frac": flop_row * P * N * H / (sim_ms * 1e-3) / 1e12 / 2500.0,
                                       "kernel": "ssc::dyn_mfma_sim_kernel<16,2,true,4,true,1,true> (256 blocks walking over 4096 row tiles)",
                                       "note": "the simulation launch alone, median of 20 launches after 0.4 s of the same launch"}}}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_smartstart_vec(args, torch, emit=True, P=N_ENVS_PER_GPU, N=16, H=4, K=64):
    """The vectorised SmartStart step (beyond SURVEY.md section 8; smartexplorationcontinuous.py:307-376 for all envs at once):
    65 536 MountainCar envs, each either navigating to a smart start (NND_MB 1 x 32, the reference's shipped navigator
    shape, N candidates, horizon H) or acting with the DDPG actor (64-32, bf16 MFMA) + OU noise; five launches per step
    from a HIP graph (actor forward, forward simulation drawing its candidates, scoring, fused step with the per-env mode,
    hand-over and new-episode logic).  Plans: recorded random-rollout paths published in the plan pool; eta 0.5."""
    import numpy as np

    from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv, VecSmartStart, make
    from smartstartcontinuous_amd import collect_samples as cs
    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent, init_dynamics_weights
    d, a = 2, 1
    denv = VecEnv("MountainCarContinuous-v0", 64, seed=1234)
    dchunk = denv.rollout(150, RandomPolicy())
    ts = cs.dataset_from_chunk(dchunk)
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    Ws, bs = init_dynamics_weights(d + a, d, 1, 32, torch.Generator().manual_seed(1234))
    model = nav.DynamicsModel(Ws, bs, norm, d, a, precision="f32")      # one small hidden layer: the fused fp32 kernel (one launch)
    env = VecEnv("MountainCarContinuous-v0", P, seed=1234, max_episode_steps=300)
    env.reset()
    agent = DDPG_Baselines_agent(make("MountainCarContinuous-v0"), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                                 lastLayerTanh=True, seed=1, training=False, ou_mu=0.4, ou_sigma=0.6, precision="bf16_mfma")
    smart = VecSmartStart(env, agent, model, eta=0.5, n_plans=8, num_control_samples=N, horizon=H, chunk_steps=K, seed=1234,
                          log_modes=True, path_shortcutting=False)
    paths = dchunk.obs[:, :150, :8].permute(2, 1, 0).double().cpu().numpy()          # 8 recorded 150-state paths
    smart.pool.publish([smart.plan_from_path(pth) for pth in paths])
    chunk = TransitionChunk(d, K, P, env.device)
    for _ in range(5):                                     # graph capture; 5 x 64 steps fill the envs' modes (300-step episodes)
        smart.rollout(K, chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps, nav_steps = max(1, args.steps // 4), 0
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        smart.rollout(K, chunk)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nav_frac = float(smart.mode_log.float().mean().item())
    ms_step = el / (reps * K) * 1e3
    res = {"metric": "env-steps/sec, vectorised SmartStart step (per-env navigate / explore mode), 65 536 MountainCar envs",
           "value": P / (ms_step * 1e-3), "unit": "env-steps/s", "n_gpus": 1, "steps": reps * K, "ms_per_step": ms_step,
           "gpu_ms_per_step": e0.elapsed_time(e1) / (reps * K), "navigated_fraction_last_chunk": nav_frac,
           "higher_is_better": True, "dtype": "f32 (navigator, fused VALU kernel), bf16 MFMA (actor)", "data": "synthetic",
           "config": {"workload": "MountainCarContinuous-v0, %d envs, eta 0.5, 8 plans on offer (recorded 150-state paths), NND_MB 1x32 x %d "
                                  "candidates x horizon %d, DDPG actor 64-32 + OU noise; 5 launches per env-step from a HIP graph; the "
                                  "per-chunk smart-start selection is not part of this figure" % (P, N, H)}}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_mpc_reference_shape(args, torch, emit=True, H=4):
    """The MPC step in the shape every shipped SmartStart run uses (examples/continuous/SmartStart_DDPG_Baselines_example.py:94-95;
    class default NND_MB_agent.py:62): N = 5000 candidate sequences per problem (and 500), horizon 4, the 1 x 32 dynamics model,
    MountainCar dims -- P problems at once with P * N ~ 1 Mi simulated rows per step.  Three launches per step: the fused fp32
    simulation of the small network (``dyn_small_sim_pair_kernel``: two rows per lane on packed fp32 FMAs, weights as scalar-load
    operands, candidates drawn in the kernel), and the two-pass scorer (generate_scores_add_delta's batch-global projection scalars, NND_MB_agent.py:566-628; pass B
    selects).  VALU-bound: 2 * (3 * 32 + 32 * 2) = 320 flop per row-step against the 157.3 TFLOP/s fp32 vector peak says little --
    how busy the vector ALU is over the launch (``valu_busy``, from the committed PMC pass) is the roof that binds: 46 % of the
    issued vector instructions are the network's FMAs, the rest the z-score divisions, Philox and the ReLU."""
    import numpy as np

    from smartstartcontinuous_amd import RandomPolicy, VecEnv
    from smartstartcontinuous_amd import collect_samples as cs
    from smartstartcontinuous_amd import navigator as nav
    from smartstartcontinuous_amd import numerical as num
    from smartstartcontinuous_amd import _ffi
    from smartstartcontinuous_amd.agents import init_dynamics_weights
    d, a = 2, 1
    denv = VecEnv("MountainCarContinuous-v0", 64, seed=1234)
    dchunk = denv.rollout(150, RandomPolicy())
    ts = cs.dataset_from_chunk(dchunk)
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    Ws, bs = init_dynamics_weights(d + a, d, 1, 32, torch.Generator().manual_seed(1234))
    model = nav.DynamicsModel(Ws, bs, norm, d, a, precision="f32")
    path = dchunk.obs[:, :150, 0].t().double().cpu().numpy()
    stds, means = num.path_deltas_stds_and_means_per_dim(path)
    rad = num.radii_calc(means, stds, 1, 1, 1)
    left = num.distances_left(path, num.elliptical_euclidean_distance_function_generator(rad))
    out = {}
    for N in (5000, 500):
        P = max(1, (1 << 20) // N)
        M = P * N
        ps = nav.MpcProblemSet([path] * P, [left] * P, [rad] * P, [0] * P, theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5)
        s0_p = torch.as_tensor(np.repeat(path[:1], P, axis=0), dtype=torch.float32, device="cuda")
        S = torch.empty((H + 1, M, d), device="cuda")
        sel = dict(scores=torch.empty(M, device="cuda"), best=torch.empty(P, dtype=torch.int32, device="cuda"),
                   best_score=torch.empty(P, device="cuda"), action=torch.empty((P, a), device="cuda"),
                   ws=torch.empty(_ffi.lib().ssc_mpc_score_workspace_bytes(P, N, H), dtype=torch.uint8, device="cuda"))

        def step(t, ev=None):
            sp = nav.mpc_sampling(N, [-1.0], [1.0], 1234, 0, t)
            if ev is not None:
                ev[0].record()
            model.do_forward_sim_sampled(s0_p, sp, M, H, out=S)
            if ev is not None:
                ev[1].record()
            nav.mpc_score_select(ps, S, sampling=sp, act_dim=a, noise_amount=0.005, seed=1234, problem_id0=0, t=t, want_path=False, out=sel)
        for t in range(args.warmup + min(args.settle_launches, 300)):
            step(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(args.steps):
            step(t)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for t, ev in enumerate(evs):
            step(t, ev)
        torch.cuda.synchronize()
        kms = sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2]
        flop_row = 2.0 * ((d + a) * 32 + 32 * d)
        out["N%d" % N] = {"problems": P, "rows": M, "ms_per_mpc_step": el / args.steps * 1e3, "row_steps_per_s": M * H * args.steps / el,
                          "mpc_problems_per_s": P * args.steps / el, "sim_kernel_ms": kms, "sim_row_steps_per_s": M * H / (kms * 1e-3),
                          "sim_tflops": flop_row * M * H / (kms * 1e-3) / 1e12}
    r5 = out["N5000"]
    res = {"metric": "row-steps/sec, MPC step in the reference's shipped shape (N = 5000 candidates, H = 4, NND_MB 1x32)",
           "value": r5["row_steps_per_s"], "unit": "row-steps/s", "n_gpus": 1, "steps": args.steps, "ms_per_step": r5["ms_per_mpc_step"],
           "higher_is_better": True, "dtype": "f32 (VALU)", "data": "synthetic", "by_candidates": out,
           "config": {"workload": "MountainCar dims (in 3, out 2), num_fc_layers 1, depth 32; N = 5000 x %d problems (and 500 x %d) ~ 1 Mi rows, "
                                  "horizon %d; simulate (in-kernel sampling) + score A + score B/select = 3 launches per MPC step"
                                  % (out["N5000"]["problems"], out["N500"]["problems"], H)},
           "roofline": {"bound": "valu", "achieved": r5["sim_tflops"], "peak": 157.3, "unit": "TFLOP/s", "frac": r5["sim_tflops"] / 157.3,
                        "traffic": None, "kernel": "ssc::dyn_small_sim_pair_kernel", "kernel_ms": r5["sim_kernel_ms"],
                        "note": "fp32 vector flop of the network alone against the packed-FMA vector peak; the z-score divisions, Philox "
                                "sampling and ReLU around them issue on the same pipe (valu_busy: its busy share over the launch)"}}
    vi = profiled_small_sim_issue()
    if vi is not None:
        res["roofline"]["valu_busy"] = vi
    if emit:
        print(json.dumps(res), flush=True)
    return res


SMALL_SIM_SOURCES = ("dyn_model.hip",)


def profiled_small_sim_issue():
    """Vector-ALU busy share of the small simulation kernel's launch (SQ_ACTIVE_INST_VALU x 4 per SIMD over SQ_BUSY_CYCLES per
    shader engine) from the committed PMC pass of `bench.py --config 5` (profiles/*/mpc_ref/valu_busy.json), reported only for
    the kernel source it was measured on."""
    import glob
    sha, best = source_sha_of(SMALL_SIM_SOURCES), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "mpc_ref", "valu_busy.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        rel = os.path.relpath(f, ROOT)
        if d.get("source_sha") == sha:
            best = dict(d, source=rel)
        elif best is None or "stale" in best:
            best = {"stale": "%s was measured on kernel source %s, this build is %s" % (rel, d.get("source_sha"), sha)}
    return best


def profiled_traffic():
    """HBM bytes per launch of the rollout kernel from the committed rocprofv3 PMC passes
    (profiles/<tag>/traffic.json: WRITE_SIZE*1024 + 2*FETCH_SIZE*1024, gfx950 correction) -- the counters cannot
    be read from inside the process, so the latest committed profile of this same command is reported, but ONLY
    when it was taken on the same kernel sources (`source_sha`); otherwise (None, "stale: ...")."""
    import glob
    best = None
    sha = source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            d = json.load(open(f))
            rel = os.path.relpath(f, ROOT)
            if d.get("source_sha") == sha:
                best = (d["write_bytes"] + d["fetch_bytes_corrected"], rel)
            elif best is None or best[0] is None:
                best = (None, "stale: %s was measured on kernel sources %s, this build is %s"
                        % (rel, d.get("source_sha", "<unstamped>"), sha))
        except Exception:
            pass
    return best


def next_rows(torch):
    """SURVEY.md section 8(f) rows that have a kernel of their own, one figure each, for the default line's
    `other_configs`: the DDPG learner (batch 64, 64-32 nets, 1000 iterations in one launch) and the SmartStart KDE
    (2000 candidates x 100 000 buffer states)."""
    import numpy as np

    import smartstartcontinuous_amd as ssc
    from smartstartcontinuous_amd import smartstart as SS
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    rng = np.random.default_rng(0)
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                                 lastLayerTanh=True, seed=1, training=False)
    cap, n_it = 100000, 1000
    dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
    s, a = dev(rng.uniform(-1.2, 0.6, (cap, 2)), torch.float32), dev(rng.uniform(-1, 1, (cap, 1)), torch.float32)
    r, t = dev(rng.normal(size=cap), torch.float32), dev(rng.random(cap) < 0.01, torch.uint8)
    idx = torch.randint(0, cap, (n_it, 64), dtype=torch.int32, device="cuda")

    def timed(fn, warm=2):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    learner_ms = timed(lambda: agent.train_on(s, a, r, t, s, idx, n_it))
    # the multi-workgroup learner (csrc/ddpg_train_wide.hip) over the reference's own network grid
    # (data/ddpg_baselines_summaries/hidden_layer_size_experiment/) at batch sizes that spread over 16..256 workgroups
    wide = {}
    for h1, h2, B in ((128, 64, 256), (200, 100, 256), (200, 100, 1024), (64, 32, 4096)):
        ag = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2,
                                  lastLayerTanh=True, seed=1, training=False, batch_size=B)
        n_w = 100
        idx_w = torch.randint(0, cap, (n_w, B), dtype=torch.int32, device="cuda")
        ms = timed(lambda: ag.train_on(s, a, r, t, s, idx_w, n_w))
        wide["%d-%d/batch%d" % (h1, h2, B)] = {"us_per_iteration": ms / n_w * 1e3, "samples_per_s": B * n_w / (ms * 1e-3)}
    pts = s[torch.randint(0, cap, (2000,), device="cuda")]
    wh, norm = SS.kde_scott_bandwidth(s)
    kde_ms = timed(lambda: SS.kde_evaluate(s, pts, wh, norm), warm=3)
    return {"ddpg_learner_us_per_iteration": learner_ms / n_it * 1e3,
            "ddpg_learner_note": "ssc_ddpg_train, batch 64, actor/critic 64-32, %d iterations in one launch" % n_it,
            "ddpg_learner_wide": wide,
            "ddpg_learner_wide_note": "ssc_ddpg_train_ws on the multi-workgroup path (batch / 16 workgroups, two launches per iteration), 100 iterations per call",
            "kde_2000x100000_ms": kde_ms}


def single_step_api(env, torch, steps=200):
    """the gym-shaped single-step API (`ssc_mc_step`, one launch per env-step,
    actions pre-generated in HBM) on the same 65 536 envs -- launch/L2-bound by construction (1.6 MB per step),
    which is why the fused rollout exists.  -> dict (env-steps/s through the Python/ctypes call path, kernel us)."""
    import ctypes

    from smartstartcontinuous_amd import _ffi
    n = env.n
    acts = torch.rand((64, n), device=env.device) * 2.0 - 1.0            # pre-generated, resident
    rew = torch.empty(n, device=env.device)
    done = torch.empty(n, dtype=torch.uint8, device=env.device)
    pos, vel, st = env.s0.clone(), env.s1.clone(), torch.zeros(n, dtype=torch.int32, device=env.device)
    def call(i):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)      # (the capture stream inside a graph capture)
        _ffi.check(env.lib.ssc_mc_step(ctypes.byref(env.params), n, _ffi.ptr(pos), _ffi.ptr(vel), _ffi.ptr(acts[i & 63]),
                                       _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(st), stream))
    for i in range(20):
        call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        call(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the same launches replayed from a HIP graph: the kernel + launch-boundary cost without the Python call path
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(64):
            call(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us_graph = e0.elapsed_time(e1) * 1e3 / 256
    return {"value": n * steps / el, "unit": "env-steps/s", "us_per_call": el / steps * 1e6,
            "graph_replay_us_per_step": us_graph, "graph_replay_env_steps_per_s": n / (us_graph * 1e-6),
            "hbm_frac_at_graph_rate": 25.0 * n / (us_graph * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "note": "ssc_mc_step, one launch per env-step, actions pre-generated in HBM; launch/L2-bound (1.6 MB per step)"}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N ...` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>` as a child process and
    return its exit code.  Nothing in THIS process has initialised a GPU (or imported torch) at this point."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=N_ENVS_PER_GPU)
    ap.add_argument("--chunk", type=int, default=CHUNK)
    ap.add_argument("--gather", choices=["bounded", "full", "none"], default="bounded")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: exchange the newest transitions every M-th chunk (M x as many steps per message: the "
                         "same records per second in fewer, larger RCCL messages)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gather path even at world size 1 "
                         "(single-GPU rehearsal of the N>1 code path)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (the measured configuration); gloo = the "
                         "packed payload staged through pinned host memory -- a rehearsal of the N > 1 control flow "
                         "(rank -> env_id0, barriers, MAX-reduced timings, rank-0 JSON) that also runs with several ranks "
                         "on ONE GPU; its numbers are not RCCL numbers")
    ap.add_argument("--cpu-budget", type=float, default=10.0,
                    help="seconds of CPU work in the cpu_baseline leg (a bounded sample of the same workload)")
    ap.add_argument("--steady-launches", type=int, default=400,
                    help="extra untimed-for-`value` launches after the timed region whose per-launch distribution is "
                         "reported as roofline.steady (N = 1 only; 0 disables)")
    ap.add_argument("--settle-launches", type=int, default=1500,
                    help="untimed launches of the step between the W warm-up steps and the timed region, so that the "
                         "timed K steps run past the post-idle power/clock transient (0 disables)")
    ap.add_argument("--series-out", default=None,
                    help="diagnostic: write (phase, start, duration) of every launch of the run to this CSV")
    ap.add_argument("--no-single-step", action="store_true", help="skip the single-step-API (ssc_mc_step) line")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="config 2, one GPU: skip the BASELINE configs[2] / configs[3] lines attached as `other_configs`")
    ap.add_argument("--no-per-env", action="store_true", help="--config 4: skip the one-navigator-per-env leg")
    ap.add_argument("--per-env-only", action="store_true", help="--config 4: ONLY the one-navigator-per-env leg (profiling runs)")
    ap.add_argument("--no-h20", action="store_true", help="--config 4: skip the horizon-20 leg (profiling runs: its launches carry the same kernel name)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config (1-based): 2 = headline random-policy rollout (default), "
                         "3 = + DDPG actor MFMA, 4 = NND_MB 2x500 forward sim + MPC; 5 = (not a BASELINE config) the MPC step in "
                         "the reference's shipped shape, N = 5000 / 500 candidates on the 1x32 model")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch rehearsal: parse, (self-)spawn the ranks, rendezvous (process group + barrier), print one "
                         "JSON line on rank 0 and exit before anything touches a GPU -- what the CPU suite runs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started as plain `python bench.py --gpus N` (the way the N = 1 line is started): become the launcher.  The N
        # ranks are CHILD processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1), started before this
        # process has imported torch or touched a GPU; rank 0's JSON line passes through on stdout and the exit code
        # is the launcher's.
        return spawn_ranks(args.gpus, sys.argv[1:])

    import torch
    import torch.distributed as dist

    if args.dry_run:
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            seen = torch.tensor([1.0])
            dist.all_reduce(seen)
            dist.barrier()
            assert int(seen.item()) == world
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "requested_gpus": args.gpus, "steps": args.steps,
                              "warmup": args.warmup, "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}), flush=True)
        return 0

    if args.config != 2:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.exit("--config 3/4 are single-GPU measurements")
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if args.config == 5:
            return bench_mpc_reference_shape(args, torch) and 0
        if args.config == 4:
            if args.per_env_only:
                return bench_config4_envs(args, torch) and 0
            bench_config4(args, torch)
            if args.no_per_env:          # profiling runs: keep the kernel statistics to the 65 536-row workload
                return 0
            return bench_config4_envs(args, torch) and 0
        return bench_config3(args, torch) and 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world          # the launcher's world size wins over a stale --gpus
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.backend != "gloo":
        sys.exit("bench.py: rank %d has no GPU of its own (%d visible); only --backend gloo may share a GPU" % (rank, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where the timing reductions live

    from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv
    from smartstartcontinuous_amd.sharding import TransitionGather

    n, K = args.envs_per_gpu, args.chunk
    env = VecEnv("MountainCarContinuous-v0", n, device=dev, seed=1234, env_id0=rank * n)
    env.reset()
    chunks = [TransitionChunk(env.obs_dim, K, n, dev) for _ in range(2)]   # double buffer
    pd = env.policy_desc(RandomPolicy())
    gather = None
    if use_dist and args.gather != "none":
        # bounded gather: the learner ingests GATHER_RECORDS records per chunk on average.  They travel as ONE message
        # every --gather-every chunks (the last M * G steps of that chunk): an exchange costs the rollout stream ~35 us
        # of queue time whatever its size (event packets + the side stream's launches, tools/exp_gather_timeline.py),
        # so fewer, larger messages -- same records per second -- keep that off the per-chunk critical path
        M = max(1, args.gather_every)
        g_steps = K if args.gather == "full" else max(1, min(K, M * max(1, GATHER_RECORDS // (n * world))))
        gather = TransitionGather(env.obs_dim, g_steps, n, world, rank, dev)

    gather_every = max(1, args.gather_every) if args.gather != "full" else 1

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(i, events=None):
        chunk = chunks[i & 1]
        if events is not None:
            events[0].record()
        env.rollout(K, out=chunk, policy_desc=pd)
        if events is not None:
            events[1].record()
        if gather is not None and (i + 1) % gather_every == 0:
            gather.submit(chunk, i & 1, env.stats)

    series = [] if args.series_out else None     # diagnostic: (phase, event pair) of EVERY launch

    def ev_pair(phase):
        if series is None:
            return None
        series.append((phase, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        return series[-1][1:]

    for i in range(args.warmup):
        one_step(i, ev_pair("warmup"))
    # Settling: the first ~10 ms of sustained load after an idle GPU are a power/clock transient (the shader clock dips
    # to ~1.7 GHz: the same launch takes 260 -> 340 -> 287 us over its first 25 repetitions, profiles/r02/drift), and
    # the launch time keeps creeping down for ~0.3 s of sustained load after that (262 -> 252 -> 243 us,
    # profiles/r02/drift/window.txt).  W = 5 warm-up launches end in the middle of the first, so
    # W is followed by --settle-launches further untimed launches of the same step (1500 = 0.4 s): the timed K steps
    # then measure the state a rollout engine actually runs in.  The count is reported in the JSON line.
    # For the record, the SAME K-step window is also timed right after the W warm-up launches, before any settling
    # (`after_warmup_only` in the JSON line): that is the number a reader gets who wants W and nothing else.
    cold_elapsed, cold_steps = None, 0
    if args.settle_launches > 0:
        cold_steps = args.steps
        barrier()
        t0c = time.perf_counter()
        for i in range(cold_steps):
            one_step(args.warmup + i, ev_pair("after warm-up only"))
        if gather is not None:
            gather.finish()
        barrier()
        cold_elapsed = time.perf_counter() - t0c
    for i in range(args.settle_launches):
        one_step(args.warmup + cold_steps + i, ev_pair("settle"))
    warm_total = args.warmup + cold_steps + args.settle_launches
    # The timed region carries ONE pair of HIP events (on the launch stream) around all K launches: an event pair around
    # every launch costs ~7 us of queue time per step (tools/exp_event_cost.py: 0.249 vs 0.2415 ms per step), which
    # would be charged to `value`.  The per-launch distribution comes from a second window of K launches right after
    # the timed one (`kernel_ms_dist`, same shape, per-launch events) and from the steady series below.
    e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    e_first.record()
    for i in range(args.steps):
        one_step(warm_total + i)
    e_last.record()
    if gather is not None:
        gather.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = e_first.elapsed_time(e_last) / max(1, args.steps)
    for i in range(args.steps):                                   # the per-launch window (untimed)
        one_step(warm_total + args.steps + i, evs[i])
    if gather is not None:
        gather.finish()
    barrier()
    per_launch = [a.elapsed_time(b) for a, b in evs]
    steps_done = warm_total + 2 * args.steps
    if gather is not None:          # one last exchange (untimed), so that the statistics check below sees every step
        gather.submit(chunks[(steps_done - 1) & 1], (steps_done - 1) & 1, env.stats)
        gather.finish()
        barrier()

    # Steady-state leg (outside the timed region, N = 1): the launches of a short timed window start from an idle
    # power state and slow down as the chip settles; `steady` is the distribution over
    # --steady-launches further back-to-back launches of the same kernel.
    steady = None
    if world == 1 and args.steady_launches > 0 and gather is None:
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steady_launches)]
        for i, ev in enumerate(sev):
            one_step(steps_done + i, ev)
        torch.cuda.synchronize()
        sl = [a.elapsed_time(b) for a, b in sev]
        if series is not None and rank == 0:
            t_first = series[0][1]
            rows = [(ph, t_first.elapsed_time(a), a.elapsed_time(b)) for ph, a, b in series]
            rows += [("timed (bracketing events only)", t_first.elapsed_time(e_first), e_first.elapsed_time(e_last))]
            rows += [("per-launch window", t_first.elapsed_time(a), a.elapsed_time(b)) for a, b in evs]
            rows += [("steady", t_first.elapsed_time(a), a.elapsed_time(b)) for a, b in sev]
            with open(args.series_out, "w") as f:
                f.write("phase,start_ms,kernel_ms\n")
                for r in rows:
                    f.write("%s,%.4f,%.5f\n" % r)
        steady = dist_stats(sl[len(sl) // 4:])            # the settled three quarters
        steady["first_quarter_mean"] = sum(sl[:len(sl) // 4]) / max(1, len(sl) // 4)
        steady["launches"] = args.steady_launches

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        kernel_ms = float(kms.item())
        if cold_elapsed is not None:
            tc = torch.tensor([cold_elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cold_elapsed = float(tc.item())

    if rank == 0:
        total_env_steps = float(n) * K * args.steps * world
        value = total_env_steps / elapsed
        per_launch_bytes = float(n) * K * BYTES_PER_STEP
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "env-steps/sec (whole node), 65 536 parallel MountainCarContinuous envs",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_launches": args.settle_launches,
            "ms_per_step": elapsed / args.steps * 1e3,
            "after_warmup_only": (None if cold_elapsed is None else
                                  {"value": float(n) * K * cold_steps * world / cold_elapsed,
                                   "ms_per_step": cold_elapsed / cold_steps * 1e3,
                                   "note": "the same K-step window timed right after the W warm-up launches, before the "
                                           "settle launches (post-idle clock transient, DESIGN.md section 6b)"}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("MountainCarContinuous-v0, %d batched envs per MI355X, random-policy fused rollout, "
                             "%d env-steps per launch, full transition log to HBM" % (n, K)),
                "envs_per_gpu": n, "chunk_steps": K, "global_envs": n * world,
                "settle_launches": args.settle_launches,
                "parallelism": "env-sharded x%d" % world,
                "gather": (args.gather if use_dist else "none"),
                "backend": (("rccl" if args.backend == "nccl" else "gloo (host-staged rehearsal)") if use_dist else "none"),
                "gather_steps_per_message": (gather.g_steps if gather is not None else 0),
                "gather_every_chunks": (gather_every if gather is not None else 0),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "ssc::rollout_kernel<McEnv, RandomPolicy<2>>",
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": per_launch_bytes,
                "kernel_ms_note": "HIP events bracketing the K timed launches on the launch stream, / K",
                "kernel_ms_dist": dist_stats(per_launch),
                "kernel_ms_dist_note": "a second window of K launches right after the timed one, one event pair per launch "
                                       "(the pairs cost ~7 us of queue time per launch, so they stay out of the timed region)",
            },
        }
        if steady is not None:
            steady["frac_at_median"] = per_launch_bytes / (steady["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            steady["frac_at_min"] = per_launch_bytes / (steady["min"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            steady["frac_at_p90"] = per_launch_bytes / (steady["p90"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            result["roofline"]["steady"] = steady
        if world == 1:
            # the write ceiling of THIS box, measured beside the kernel ("use the measured ceiling alongside
            # nominal"): a plain fill of a buffer of the chunk's size, median of 10 launches
            try:
                fill = torch.empty(int(per_launch_bytes) // 4, dtype=torch.float32, device=dev)
                for _ in range(3):
                    fill.zero_()
                fe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for e0, e1 in fe:
                    e0.record(); fill.zero_(); e1.record()
                torch.cuda.synchronize()
                fms = sorted(x.elapsed_time(y) for x, y in fe)[len(fe) // 2]
                result["roofline"]["fill_ceiling"] = {"achieved": fill.numel() * 4 / (fms * 1e-3) / 1e9, "unit": "GB/s", "ms": fms,
                                                      "bytes": fill.numel() * 4, "kernel": "torch fill (zero_) of a buffer of the chunk's size",
                                                      "rollout_over_fill": (per_launch_bytes / (kernel_ms * 1e-3)) / (fill.numel() * 4 / (fms * 1e-3))}
                del fill
            except Exception as e:           # noqa: BLE001
                result["roofline"]["fill_ceiling"] = {"error": repr(e)}
        tr = profiled_traffic()
        if tr is not None and n == N_ENVS_PER_GPU and K == CHUNK and gather is None:   # the PMC pass is the N = 1 kernel alone:
            # a run that also packs and ships transitions moves more bytes than it measured
            result["roofline"]["traffic"] = tr[0]
            result["roofline"]["traffic_source"] = tr[1]
        if world == 1 and not args.no_single_step and n == N_ENVS_PER_GPU:
            result["single_step_api"] = single_step_api(env, torch)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        if world == 1 and not use_dist and not args.no_other_configs and n == N_ENVS_PER_GPU:
            # BASELINE configs[2] (fused actor rollout) and configs[3] (forward simulation + MPC scoring) measured in the
            # same process AFTER the headline's timed region: the same functions `--config 3` / `--config 4` print as
            # lines of their own (without their CPU legs), so that the driver's run carries all three
            import copy
            import gc
            chunks.clear()                               # give the headline's buffers back first: the other configurations
            del env                                      # then start from the allocator state their own runs start from
            gc.collect()
            torch.cuda.empty_cache()
            a2 = copy.copy(args)
            a2.no_cpu_baseline = True
            other = {}
            # (a failure in one of these legs must not cost the headline line: it is reported in place of the leg)
            for name, fn in (("config3", bench_config3), ("config4", bench_config4), ("config4_per_env", bench_config4_envs),
                             ("smartstart_vec", bench_smartstart_vec), ("mpc_reference_shape", bench_mpc_reference_shape)):
                try:
                    r = fn(a2, torch, emit=False)
                    other[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "roofline", "h20",
                                                     "navigated_fraction_last_chunk", "by_candidates") if k in r}
                    other[name]["workload"] = r["config"]["workload"]
                except Exception as e:           # noqa: BLE001
                    other[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                other["next_rows"] = next_rows(torch)
            except Exception as e:               # noqa: BLE001
                other["next_rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
            result["other_configs"] = other
        print(json.dumps(result), flush=True)
    if use_dist:
        if gather is not None and rank == 0:
            # the learner rank really received every rank's last steps
            obs, act, rew, obs2, done = gather.unpack(world - 1)
            assert obs.shape == (env.obs_dim, gather.g_steps, n) and bool(torch.isfinite(act).all())
            # ... and the statistics that rode in the payloads add up to every rank's env-steps so far
            st = gather.global_stats.cpu().numpy()
            assert st[2] == float(n) * K * world * (warm_total + 2 * args.steps), st
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
