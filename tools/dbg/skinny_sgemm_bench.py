import os, sys; sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import ACT_SILU
torch.manual_seed(0)
M, K, N = 32, 1024, 24 * 6144 + 2048
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(3): ops.sgemm(a, w, b, act_in=ACT_SILU, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.sgemm(a, w, b, act_in=ACT_SILU, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("skinny sgemm M=%d N=%d K=%d: %.1f us  %.2f TB/s" % (M, N, K, us, N * K * 4 / us / 1e6))
ref = torch.nn.functional.silu(a.double()) @ w.double().T + b.double()
print("rel err", float(((out.double() - ref) ** 2).sum() / (ref ** 2).sum()))
