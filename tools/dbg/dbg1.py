import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, json
from types import SimpleNamespace
from ldt_amd import ops
from ldt_amd.diffusion import DiffusionVPSDE
from oracle import ldt_oracle as O
from conftest import to_ns
cfg = to_ns(json.load(open('tests/golden/tiny_cfg.json')))
N=1000; cfg.sde.sample_N=N
sde = DiffusionVPSDE(cfg.sde); osde=O.VPSDE(cfg.sde)
ts, coef, mode = sde.step_table(N,"ancestral",1e-6)
g=torch.Generator().manual_seed(1)
x=torch.randn(2,8,120,generator=g)*5; p=torch.randn(2,8,120,generator=g); z=torch.randn(2,8,120,generator=g)
for i in (0,500,999):
    t=torch.ones(2)*ts[i]
    idx=(t*(N-1)).long(); beta=osde.betas[idx]
    std=torch.sqrt(osde.var(t))
    print(i, "coef", coef[i].tolist(), "beta", float(beta[0]), "std", float(std[0]), "sq1mb", float(torch.sqrt(1.-beta)[0]), "sqb", float(torch.sqrt(beta)[0]))
    score=-p/std[:,None,None]
    xm=(x+beta[:,None,None]*score)/torch.sqrt(1.-beta)[:,None,None]
    # device with params=p, coef s.t. isolate: mode 0
    xmd=torch.empty_like(x,device='cuda')
    out=ops.sampler_step(x.cuda(),p.cuda(),coef.cuda(),i,0,noise=z.cuda(),x_mean_out=xmd)
    d=(xmd.cpu()-xm)
    print("  mismatches", int((d!=0).sum()), "max abs", float(d.abs().max()), "max rel", float((d.abs()/xm.abs()).max()))
    # emulate in float64 to see which is correctly rounded
    import numpy as np
    sc64 = (-p.double()/std.double()[:,None,None]).float()
    print("  score cpu==fp64-rounded:", bool(torch.equal(score, sc64)))
