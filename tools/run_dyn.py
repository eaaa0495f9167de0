"""Run the MFMA forward sim a few times (profiling target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.exp_nav import make
M, H = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 4
model, d, a = make((4, 500, 500, 3))
A = torch.rand((M, H, a), device="cuda") * 2 - 1
s0 = torch.randn((M, d), device="cuda") * 0.3
S = torch.empty((H + 1, M, d), device="cuda")
for _ in range(6):
    model.do_forward_sim(s0, A, precision="bf16_mfma", out=S)
torch.cuda.synchronize()
