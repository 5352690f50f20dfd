"""Race screen for the 3-stage 64x64-tile GEMM (and the 2-stage shapes): same inputs 40 times beside a memory-bound stream,
fp32 outputs must be bit-identical and match an fp64 reference."""
import sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_F32
noise_stream = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
bad = 0
for (M, N, K) in ((2048, 1024, 4096), (2048, 1024, 1024), (2048, 3072, 1024), (640, 120, 1024), (300, 200, 64), (2048, 1024, 128)):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g)
    refd = x.double() @ w.double().T + b.double()
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    ref = None
    for it in range(40):
        with torch.cuda.stream(noise_stream):
            junk.add_(1.0)
        out = ops.gemm_bf16(xd, wd, bd, EPI_F32)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
            err = float(((ref.cpu().double() - refd) ** 2).sum() / (refd ** 2).sum())
            assert err < 1e-9, (M, N, K, err)
        elif not torch.equal(ref, out):
            bad += 1; print("shape", (M, N, K), "iteration", it, "differs in", int((ref != out).sum()))
print("gemm soak: %d mismatches" % bad)
sys.exit(1 if bad else 0)
