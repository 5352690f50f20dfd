"""tools/dbg: per-step relative MSE of the GPU trajectory against the oracle on the reference-trained tiny fixture (tests/golden/trained_tiny.npz)."""
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from conftest import load_golden, to_ns, rel_mse
import ldt_amd
from oracle import ldt_oracle as O
cfg = to_ns(json.load(open('tests/golden/tiny_cfg.json')))
a, sds = load_golden("trained_tiny")
score = ldt_amd.Score(cfg.score); score.load_state_dict(sds["w"], strict=True)
comp = ldt_amd.Compressor(cfg.compressor); comp.load_state_dict(sds["c"], strict=True)
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
x0, noises = a["x0"], a["noises"]
nl = [noises[i] for i in range(noises.shape[0])]
with torch.no_grad():
    rec = []
    pts_o, eps_o = O.trainer_sample(sds["w"], sds["c"], cfg, x0, nl, record=rec)
for ug in (False, True):
    traj = []
    pts, eps = tr.sample(x0.shape[0], x0=x0, noise=noises, trajectory=traj, use_graph=ug)
    xs = traj[0].cpu()
    curve = [rel_mse(xs[i], rec[i][3]) for i in range(len(rec))]
    print("graph=%s final latents %.3e (vs capture %.3e) | per-step:" % (ug, rel_mse(eps.cpu(), eps_o), rel_mse(eps.cpu(), a["eps"])), " ".join("%.1e" % c for c in curve))
    print("   xs[-1] vs eps returned: %.3e ; oracle rec[-1] vs oracle eps: %.3e" % (rel_mse(xs[-1], eps.cpu()), rel_mse(rec[-1][3], eps_o)))
    # teacher-forced Score error at every step on the ORACLE's states
    sde = O.VPSDE(cfg.sde)
    errs = []
    for i in range(len(rec)):
        xin = x0 if i == 0 else rec[i - 1][3]
        tt = torch.full((x0.shape[0],), float(torch.linspace(1.0, cfg.sde.sample_time_eps, cfg.sde.sample_N)[i]))
        with torch.no_grad():
            ref = O.score_forward(sds["w"], cfg.score, xin, tt)
        out = score(xin.cuda(), tt.cuda()).cpu()
        errs.append(rel_mse(out, ref))
        if i in (0, 25, 49) and not ug:
            d = (out.double() - ref.double())
            print("   step %d: |ref| rms %.3f, err rms %.2e; per-token err / ref: %s" % (i, float(ref.pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt()),
                  " ".join("%.1e" % float((d[0, k] ** 2).sum() / (ref[0, k].double() ** 2).sum()) for k in range(ref.shape[1]))))
    print("   teacher-forced Score error per step:", " ".join("%.1e" % e for e in errs))
print("rec entry layout:", [type(v).__name__ + (str(tuple(v.shape)) if torch.is_tensor(v) else "") for v in rec[0]])
