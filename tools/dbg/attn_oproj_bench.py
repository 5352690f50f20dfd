import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_RESID_F32
B, H, Nq, Nk, dh = 128, 4, 2048, 256, 32
C = H * dh
torch.manual_seed(0)
q = torch.randn(B * Nq, C, device="cuda").to(torch.bfloat16); kv = torch.randn(B * Nk, 2 * C, device="cuda").to(torch.bfloat16)
wo = (torch.randn(C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); bo = torch.randn(C, device="cuda"); x = torch.randn(B * Nq, C, device="cuda")
def fused(): ops.attention_oproj_resid_(q, kv[:, :C], kv[:, C:], B, H, Nq, Nk, dh, wo, bo, x)
def unfused():
    o = ops.attention_fwd(q, kv[:, :C], kv[:, C:], B, H, Nq, Nk, dh)
    ops.gemm_bf16(o.view(B * Nq, C), wo, bo, EPI_RESID_F32, out=x, resid=x)
def attn_only(): ops.attention_fwd(q, kv[:, :C], kv[:, C:], B, H, Nq, Nk, dh)
for fn in (fused, unfused, attn_only):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us" % (fn.__name__, e0.elapsed_time(e1) / 10 * 1e3), flush=True)
