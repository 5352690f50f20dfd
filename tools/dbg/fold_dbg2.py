#!/usr/bin/env python3
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ldt_amd
from oracle import ldt_oracle as O
N = 20
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N, **{
    "score.hidden_size": 256, "score.num_heads": 4, "score.num_blocks": 3, "score.t_dim": 128,
    "compressor.max_outputs": 256, "compressor.outsize": 256, "data.tr_max_sample_points": 256})
torch.manual_seed(11)
score = ldt_amd.Score(cfg.score)
comp = ldt_amd.Compressor(cfg.compressor)
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
B, T, z = 8, cfg.score.z_scale, cfg.score.z_dim
x0, noises = O.draw_noises(77, B, T, z, N)
for nsteps in (1, 2, 5, 20):
    for mode in ("0", "2"):
        os.environ["LDT_LN_FOLD"] = mode
        kw = dict(score_fn=tr.score_fn, num_samples=B, N=N, predictor="ancestral", corrector=None, corrector_steps=1, shape=(T, z),
                  time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=x0,
                  noise=torch.stack(noises))
        out = tr.SDE.sample_discrete(**kw, use_graph=0)
        print("mode", mode, "finite", bool(torch.isfinite(out).all()), "absmean", float(out.abs().mean()), flush=True)
    break
