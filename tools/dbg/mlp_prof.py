"""rocprofv3 target: the Compressor's fused MLP kernel alone, 2 M rows (a decoder level of 1024 clouds), plain (affine LayerNorm) form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
C, M = int(os.environ.get("CH", 128)), int(os.environ.get("ROWS", 1024 * 2048))
torch.manual_seed(0)
x = torch.randn(M, C, device="cuda")
w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
for _ in range(int(os.environ.get("REPS", 5))): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, ln_w=lw, ln_b=lb)
torch.cuda.synchronize()
