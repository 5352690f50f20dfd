import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, json
import ldt_amd
from oracle import ldt_oracle as O
from conftest import to_ns, load_golden, rel_mse
cfg = to_ns(json.load(open('tests/golden/tiny_cfg.json')))
_, ssd = load_golden("score_tiny"); tg, csd = load_golden("trainer_sample_tiny")
score = ldt_amd.Score(cfg.score); score.load_state_dict(ssd["w"]); comp = ldt_amd.Compressor(cfg.compressor); comp.load_state_dict(csd["c"])
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
pts, eps = tr.sample(2, x0=tg["x0"], noise=tg["noises"], use_graph=0)
print("eps rel_mse", rel_mse(eps.cpu(), tg["eps"]), "pts rel_mse", rel_mse(pts.cpu(), tg["points"]))
cd = O.chamfer_cd(pts.cpu(), tg["points"]); r2 = (tg["points"]**2).sum(-1).mean(1)
print("cd/r2", (cd/r2).tolist(), "|eps| rms", float(tg["eps"].pow(2).mean().sqrt()), "|pts| rms", float(tg["points"].pow(2).mean().sqrt()))
# decoder sensitivity: decode the reference eps on GPU, and decode perturbed eps on CPU
p2 = comp.sample((2,64), given_eps=tg["eps"].cuda())
print("decode(ref eps) vs ref pts", rel_mse(p2.cpu(), tg["points"]))
g = torch.Generator().manual_seed(0)
for rel in (1e-3, 3e-3, 1e-2):
    pe = tg["eps"] * (1 + rel*torch.randn(tg["eps"].shape, generator=g))
    pp = O.compressor_decode(csd["c"], cfg.compressor, pe)
    print("cpu decode with eps rel perturbation", rel, "-> eps rel_mse", rel_mse(pe, tg["eps"]), "pts rel_mse", rel_mse(pp, tg["points"]))
# per-step curve
rec=[]
tr.SDE.sample_discrete(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", corrector=None, corrector_steps=1, shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"], record=rec)
print("per-step x rel_mse", ["%.1e"%rel_mse(rec[i][0].cpu(), tg["step_x"][j]) for j,i in enumerate(tg["step_ids"].tolist())])
# sensitivity of the CPU trajectory itself to a 2^-9 relative perturbation of x0
x0p = tg["x0"]*(1+2**-9*torch.randn(tg["x0"].shape, generator=g))
ptsp, epsp = O.trainer_sample(ssd["w"], csd["c"], cfg, x0p, list(tg["noises"]))
print("cpu fp32 with x0 perturbed 2^-9: eps rel_mse", rel_mse(epsp, tg["eps"]), "pts", rel_mse(ptsp, tg["points"]))
