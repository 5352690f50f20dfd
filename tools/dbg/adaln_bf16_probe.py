"""Probe (round 5): the per-step conditional AdaLN rows of BASELINE configs[4] (M = 32 samples, N = 149,504 modulation columns, K = 1024) as a
bf16-weight GEMM through the existing kernels against the shipped fp32 SGEMM (604 MB of fp32 weights per step): time and error of the rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_F32
torch.manual_seed(0)
M, N, K = int(os.environ.get("M", 32)), 149504, 1024
c = torch.nn.functional.silu(torch.randn(M, K, device="cuda") * 0.5)
w = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
b = (torch.rand(N, device="cuda") * 2 - 1) / K ** 0.5
wb = w.to(torch.bfloat16)
ref = (c.double() @ w.double().T + b.double())
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rel = lambda a: float(((a.double() - ref) ** 2).sum() / (ref ** 2).sum())
out = torch.empty(M, N, device="cuda")
us = t(lambda: ops.sgemm(c, w, b, out=out))
print("fp32 SGEMM (shipped): %.1f us, %.2f TB/s of weights, rel-MSE %.2e" % (us, N * K * 4 / us / 1e6, rel(out)))
for Mp in (M, 64, 128):
    cb = torch.zeros(Mp, K, device="cuda", dtype=torch.bfloat16); cb[:M] = c.to(torch.bfloat16)
    try:
        o = ops.gemm_bf16(cb, wb, b, EPI_F32)
        us = t(lambda: ops.gemm_bf16(cb, wb, b, EPI_F32))
        print("bf16 x bf16 (rows padded to %d): %.1f us, %.2f TB/s of weights, rel-MSE %.2e" % (Mp, us, N * K * 2 / us / 1e6, rel(o[:M])))
    except Exception as e:
        print("M=%d: %s" % (Mp, e))
# hi + lo split of the activations stacked as 2M rows: W read once, the two row blocks added
hi = c.to(torch.bfloat16); lo = (c - hi.float()).to(torch.bfloat16)
cb = torch.cat([hi, lo]).contiguous()
o = ops.gemm_bf16(cb, wb, None, EPI_F32)
print("bf16 weights, activations hi + lo: rel-MSE %.2e" % rel(o[:M] + o[M:] + b))
print("weights rounded only (fp64 product): rel-MSE %.2e" % rel(c.double() @ wb.double().T + b.double()))
