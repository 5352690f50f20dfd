"""Race screen for the small-batch kernels (csrc/gemm_mid.hip: loader / compute waves, LN folding through the loader waves, QKV + self-attention
and q + cross-attention in one launch): whole-loop reproducibility of `sample()` at the shipped 32-token config (B = 64) and at BASELINE configs[4]'s
per-GPU share (B = 32, ViPC condition) — three calls with the same seed and x0 beside a memory-bound stream on a second HIP stream must return
bit-identical latents and points."""
import sys
sys.path.insert(0, '.')
import torch, ldt_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
noise_stream = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
ok = True
for name, B, cond in (("shipped T=32, B=64", 64, None), ("configs[4] share, B=32, ViPC", 32, "vipc")):
    g = torch.Generator().manual_seed(3 + B)
    x0 = torch.randn(B, 32, cfg.score.z_dim, generator=g)
    condition = None
    if cond:
        condition = (torch.randn(B, cfg.score.hidden_size, 32, generator=g).cuda(), torch.randn(B, cfg.score.t_dim, generator=g).cuda())
    outs = []
    for rep in range(3):
        with torch.cuda.stream(noise_stream):
            for _ in range(50): junk.add_(1.0)
        pts, eps = tr.sample(B, x0=x0, seed=99, condition=condition)
        torch.cuda.synchronize()
        outs.append((pts.clone(), eps.clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    fin = bool(torch.isfinite(outs[0][1]).all())
    print("%s, %d steps: three runs bit-identical: %s, finite: %s" % (name, N, same, fin), flush=True)
    ok = ok and same and fin
sys.exit(0 if ok else 1)
