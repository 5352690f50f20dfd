"""tools/dbg: the per-step conditional AdaLN rows of BASELINE configs[4] — one skinny fp32 GEMM, M = batch share, weights streamed once.
SKINNY_PAD=<floats> pads the weight rows (ldb = K + pad: breaks the 4-KiB row stride); LDT_SGEMM_SKINNY picks the kernel form."""
import os, sys; sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import ACT_SILU
torch.manual_seed(0)
M, K, N = int(os.environ.get("SKINNY_M", "32")), 1024, int(os.environ.get("SKINNY_N", str(24 * 6144 + 2048)))
pad = int(os.environ.get("SKINNY_PAD", "0"))
a = torch.randn(M, K, device="cuda"); wfull = torch.randn(N, K + pad, device="cuda") * 0.03; w = wfull[:, :K]; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
pre = os.environ.get("SKINNY_PRE", "1") == "1"          # SiLU applied to A beforehand (what the sampling loop does since round 4)
act = 0 if pre else ACT_SILU
a_in = torch.nn.functional.silu(a) if pre else a
for _ in range(3): ops.sgemm(a_in, w, b, act_in=act, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.sgemm(a_in, w, b, act_in=act, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
ref = (a_in.double() if pre else torch.nn.functional.silu(a.double())) @ w.double().T + b.double()
print("skinny sgemm M=%d N=%d K=%d pad=%d form=%s: %.1f us  %.2f TB/s  rel err %.1e" % (M, N, K, pad, os.environ.get("LDT_SGEMM_SKINNY", "-"), us, N * K * 4 / us / 1e6,
      float(((out.double() - ref) ** 2).sum() / (ref ** 2).sum())))
