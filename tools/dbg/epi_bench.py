import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_RESID_F32
M = 16384
torch.manual_seed(0)
for name, N, K, epi in (("o-discard", 1024, 1024, 5), ("o-resid", 1024, 1024, EPI_RESID_F32), ("dn-discard", 1024, 4096, 5), ("dn-resid", 1024, 4096, EPI_RESID_F32)):
    x = (torch.randn(M, K, device="cuda") * 0.05).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); gate = torch.randn(64, N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    kw = dict(out=out)
    if epi == EPI_RESID_F32: kw.update(resid=out, gate=gate, gate_sample_stride=N, rows_per_sample=256)
    for _ in range(3): ops.gemm_bf16(x, w, b, epi, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_bf16(x, w, b, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
