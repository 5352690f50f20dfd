"""Whole-loop reproducibility at the bench shape: two `sample()` calls with the same seed and x0 (N steps, B=64, T=256,
LN folding on) must return bit-identical latents and points; also with two sub-batch streams (equal to rounding)."""
import os, sys
sys.path.insert(0, '.')
import torch, ldt_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
g = torch.Generator().manual_seed(3)
x0 = torch.randn(64, 256, 120, generator=g)
outs = []
for rep in range(2):
    pts, eps = tr.sample(64, x0=x0, seed=99)
    torch.cuda.synchronize()
    outs.append((pts.clone(), eps.clone()))
same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
print("single stream, two runs bit-identical:", same, "finite:", bool(torch.isfinite(outs[0][1]).all()))
os.environ["LDT_STREAMS"] = "2"
pts2, eps2 = tr.sample(64, x0=x0, seed=99)
torch.cuda.synchronize()
d = float(((eps2.double() - outs[0][1].double()) ** 2).sum() / (outs[0][1].double() ** 2).sum())
print("two streams vs one: rel-MSE of the latents %.3e" % d)
sys.exit(0 if same and d < 1e-4 else 1)
