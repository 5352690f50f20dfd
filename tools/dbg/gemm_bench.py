"""A/B of the GEMM variants on the Score shapes (random data), in one process."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
M = int(os.environ.get("M", 16384))
shapes = [("qkv", 3072, 1024, EPI_BF16), ("o", 1024, 1024, EPI_RESID_F32), ("up", 4096, 1024, EPI_GELU_BF16), ("dn", 1024, 4096, EPI_RESID_F32)]
if os.environ.get("DBG"):
    shapes = [("up-discard", 4096, 1024, 5), ("up-bf16", 4096, 1024, EPI_BF16), ("up-gelu", 4096, 1024, EPI_GELU_BF16), ("dn-discard", 1024, 4096, 5), ("dn-resid", 1024, 4096, EPI_RESID_F32), ("o-discard", 1024, 1024, 5), ("o-resid", 1024, 1024, EPI_RESID_F32), ("k8192-discard", 1024, 8192, 5), ("k512-discard", 1024, 512, 5), ("k64-discard", 1024, 64, 5)]
torch.manual_seed(0)
for name, N, K, epi in shapes:
    x = (torch.randn(M, K, device="cuda")).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); gate = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi in (EPI_RESID_F32, 5) else torch.bfloat16)
    kw = dict(out=out)
    if epi == EPI_RESID_F32: kw.update(resid=out, gate=gate, rows_per_sample=256)
    for _ in range(3): ops.gemm_bf16(x, w, b, epi, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n): ops.gemm_bf16(x, w, b, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%s M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
