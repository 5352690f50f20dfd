import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_GELU_BF16, EPI_RESID_F32
C, M = 128, int(os.environ.get("ROWS", 128 * 2048))
torch.manual_seed(0)
x = torch.randn(M, C, device="cuda")
w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
def fused(): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, ln_w=lw, ln_b=lb)
def unfused():
    h = ops.layernorm_modulate(x, w=lw, b=lb)
    u = ops.gemm_bf16(h, w_up, b_up, EPI_GELU_BF16)
    ops.gemm_bf16(u, w_dn, b_dn, EPI_RESID_F32, out=x, resid=x)
for fn in (fused, unfused):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("%s: %.1f us  %.0f TFLOP/s  x-traffic %.2f TB/s" % (fn.__name__, us, 2.0 * M * C * 4 * C * 2 / us / 1e6, M * C * 8 / us / 1e6), flush=True)
