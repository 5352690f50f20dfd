"""Race screen for the LN-folded GEMM pair at the bench shape: the same inputs 40 times, outputs must be bit-identical
(a mis-ordered LDS-DMA / statistics hand-off shows up as run-to-run differences), under a concurrent memory-bound stream."""
import sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16
M, D = 16384, 1024
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 1024, generator=g).cuda().to(torch.bfloat16); wo = (torch.randn(D, 1024, generator=g) / 32).cuda().to(torch.bfloat16)
bo = torch.randn(D, generator=g).cuda(); x0 = (torch.randn(M, D, generator=g) + 0.5).cuda(); gate = torch.randn(D, generator=g).cuda()
sc = (0.3 * torch.randn(D, generator=g)).cuda()
w2 = (torch.randn(4096, D, generator=g) / 32).cuda().to(torch.bfloat16); S = torch.randn(4096, generator=g).cuda(); C = torch.randn(4096, generator=g).cuda()
w3 = (torch.randn(3072, D, generator=g) / 32).cuda().to(torch.bfloat16)
noise_stream = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
ref = None
bad = 0
for it in range(40):
    with torch.cuda.stream(noise_stream):
        junk.add_(1.0)                                   # uneven extra load on the memory system
    x = x0.clone()
    xs, st = ops.gemm_resid_lnstats(a, wo, bo, x, sc, gate=gate, rows_per_sample=256)
    y1 = ops.gemm_lnfold(xs, w2, st, S, C, EPI_GELU_BF16)
    y2 = ops.gemm_lnfold(xs, w3, st, S[:3072].contiguous(), C[:3072].contiguous(), EPI_BF16)
    torch.cuda.synchronize()
    cur = (x.clone(), xs.clone(), st.clone(), y1.clone(), y2.clone())
    if ref is None:
        ref = cur
        assert all(bool(torch.isfinite(t.float()).all()) for t in cur)
    else:
        for i, (p, q) in enumerate(zip(ref, cur)):
            if not torch.equal(p, q):
                bad += 1
                print("iteration %d: output %d differs in %d elements" % (it, i, int((p != q).sum())))
print("soak: %d mismatching outputs over 39 repeats" % bad)
sys.exit(1 if bad else 0)
