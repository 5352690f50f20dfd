"""Race screen for the attention kernels: same inputs 40 times beside a memory-bound stream, outputs must be bit-identical
(Score shape B=64,H=16,T=256,Dh=64; ragged Nk; Compressor shapes Dh=32)."""
import sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
noise_stream = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
bad = 0
for (B, H, Nq, Nk, dh) in ((64, 16, 256, 256, 64), (8, 16, 200, 177, 64), (32, 4, 2048, 256, 32), (32, 4, 256, 2048, 32)):
    C = H * dh
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B * Nq, C, generator=g).cuda().to(torch.bfloat16); kv = (torch.randn(B * Nk, 2 * C, generator=g) * 1.5).cuda().to(torch.bfloat16)
    ref = None
    for it in range(40):
        with torch.cuda.stream(noise_stream):
            junk.add_(1.0)
        out = ops.attention_fwd(q, kv[:, :C], kv[:, C:], B, H, Nq, Nk, dh)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone(); assert bool(torch.isfinite(ref.float()).all())
        elif not torch.equal(ref, out):
            bad += 1; print("shape", (B, H, Nq, Nk, dh), "iteration", it, "differs in", int((ref != out).sum()))
print("attention soak: %d mismatches" % bad)
sys.exit(1 if bad else 0)
