import os, sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
M = 16384
for name, N, K in (("dn", 1024, 4096), ("up", 4096, 1024)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    for _ in range(3): ops.gemm_bf16(x, w, None, 5, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_bf16(x, w, None, 5, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s %s: %.1f us %.0f TF" % (os.environ.get("LDT_HIP_LIB", "product")[-14:], name, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
