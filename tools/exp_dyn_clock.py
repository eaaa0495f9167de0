"""In-kernel clock of dyn_mfma_sim_kernel (diagnostic build tools/variants/dyn_mfma_clk.hip -> tools/_build/libssc_clk.so, `make -C tools`):
after ~2 s of back-to-back launches, Delta s_memtime / Delta s_memrealtime x 100 MHz per block."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "tools/_build/libssc_clk.so")
import torch, numpy as np, time
from exp_nav import make
dims, M, H = (4, 500, 500, 3), (int(sys.argv[4]) if len(sys.argv) > 4 else 65536), int(sys.argv[2]) if len(sys.argv) > 2 else 20
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
TILES = M // 256
BLOCKS = TILES if TILES <= N_CU else N_CU          # more tiles than CUs: the blocks walk (dyn_mfma.hip launch_sim)
HT = H * -(-TILES // BLOCKS)                         # steps a block runs over all its row tiles
model, d, a = make(dims)
A = torch.rand((M, H, a), device="cuda") * 2 - 1
s0 = torch.randn((M, d), device="cuda") * 0.3
S = torch.empty((H + 1, M, d), device="cuda")
SAMPLE = len(sys.argv) > 3 and sys.argv[3] == "sample"      # the MPC step's form: candidate actions drawn in the kernel
if SAMPLE:
    from smartstartcontinuous_amd import navigator as nav
    sp = nav.mpc_sampling(4096 if M == 65536 else 16, [-1.0] * a, [1.0] * a, 1234, 0, 0)
    s0p = s0[:(16 if M == 65536 else M // 16)].contiguous()
    run = lambda: model.do_forward_sim_sampled(s0p, sp, M, H, precision="bf16_mfma", out=S)
else:
    run = lambda: model.do_forward_sim(s0, A, precision="bf16_mfma", out=S)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
raw = S.view(torch.int32).flatten()[: 48 * BLOCKS].cpu().numpy().astype(np.uint32).reshape(-1, 2, 12, 2).astype(np.uint64)
v = raw[..., 0] | (raw[..., 1] << np.uint64(32))       # [block][group][dc, dr, input, layer 1, hidden tiles, tail, entry_rt, loop_rt]
dc, dr = v[:, 0, 0], v[:, 0, 1]
clk = dc / dr * 100.0
H_ONE, H = H, HT
mf = 560 * 2 * 32 * H   # MFMA pipe cycles per SIMD for the step loop (2 waves x 17920 cyc of MFMAs per step)
print("launch %.4f ms; blocks %d; step-loop cycles median %.0f (%.0f per step); realtime median %.1f us; clock median %.0f MHz (min %.0f max %.0f); MFMA floor %.0f cyc -> pipe busy %.1f %%"
      % (e0.elapsed_time(e1), len(dc), np.median(dc), np.median(dc) / H, np.median(dr) / 100.0, np.median(clk), clk.min(), clk.max(), mf, 100.0 * mf / np.median(dc)))
for gi in range(2):
    ph = np.median(v[:, gi, 2:6].astype(np.int64), axis=0) / H
    bw = np.median(v[:, gi, 8:10].astype(np.int64), axis=0) / H
    print("group %d per step: input code %.0f, layer 1 %.0f (of which phase barrier %.0f), hidden tiles %.0f (%.0f each; barrier waits %.0f in all), step tail %.0f cycles" % (gi, ph[0], ph[1], bw[0], ph[2], ph[2] / 16, bw[1], ph[3]))
entry, loop0 = v[:, 0, 6].astype(np.int64), v[:, 0, 7].astype(np.int64)
exit_ = loop0 + v[:, 0, 1].astype(np.int64)
# a block whose stamp dwords were overwritten by another block's late S[0] rows shows absurd values: drop it
ok = (np.abs(entry - np.median(entry)) < 10_000_000) & (np.abs(loop0 - np.median(loop0)) < 10_000_000) & (dr < 4 * np.median(dr))
entry, loop0, exit_ = entry[ok], loop0[ok], exit_[ok]
t0 = entry.min()
print("realtime (us): kernel span first entry -> last exit %.1f; block entry stagger median %.1f max %.1f; entry -> step loop median %.1f max %.1f; step loop median %.1f max %.1f; last exit - median exit %.1f"
      % ((exit_.max() - t0) / 100.0, np.median(entry - t0) / 100.0, (entry - t0).max() / 100.0, np.median(loop0 - entry) / 100.0, (loop0 - entry).max() / 100.0,
         np.median(exit_ - loop0) / 100.0, (exit_ - loop0).max() / 100.0, (exit_.max() - np.median(exit_)) / 100.0))
bad = np.where(~ok)[0]
if len(bad):
    print("blocks with implausible stamps:", bad[:16].tolist(), "raw:", [hex(int(x)) for x in v[bad[0], 0]])
