import os, sys, copy, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import to_ns
import ldt_amd
cfg = to_ns(json.load(open(os.path.join(ROOT, "tests/golden/tiny_cfg.json"))))
cfg.score.condition = True
torch.manual_seed(6)
score = ldt_amd.Score(cfg.score).cuda()
B, T = 2, cfg.score.z_scale
g = torch.Generator().manual_seed(1)
pts = torch.randn(B, 96, 3, generator=g); img = torch.randn(B, 3, 64, 64, generator=g)
comp = ldt_amd.Compressor(cfg.compressor)
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
cond = {"img": img.cuda(), "pts": pts.cuda()}
x0 = torch.randn(B, T, cfg.score.z_dim, generator=g)
res = []
for i in range(3):
    res.append(tr.sample(B, condition=cond, x0=x0, seed=5))
    res.append(tr.sample(B, condition=score.c_net(cond), x0=x0, seed=5))
for i, (p, e) in enumerate(res):
    print(i, "e diff", (e - res[0][1]).abs().max().item(), "p diff", (p - res[0][0]).abs().max().item(), "e rms", e.pow(2).mean().sqrt().item(), "finite", torch.isfinite(p).all().item())
c1 = score.c_net(cond); c2 = score.c_net(cond)
print("c_net det:", torch.equal(c1[0], c2[0]), torch.equal(c1[1], c2[1]))
eps = torch.randn(B, T, 120, device="cuda")
d1 = comp.sample((B, 64), given_eps=eps); d2 = comp.sample((B, 64), given_eps=eps)
print("decode det:", torch.equal(d1, d2))
