#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU's share: ShapeNet-ViPC completion sampling, batch 256 over 8 GPUs = 32 shapes per GPU,
256 latent tokens, image-conditioned AdaLN rows (per sample, per step) + cross-attention to the point-cloud condition
on even blocks (step-invariant K/V).  Synthetic condition tensors stand in for ConditionNet's outputs.
Prints one JSON line: shapes/s per GPU for the conditional and, for reference, the unconditional sampler at the same batch."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldt_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, T, S = 32, 256, 32
cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
g = torch.Generator().manual_seed(5)
pts = torch.randn(B, cfg.score.hidden_size, S, generator=g).cuda(); img = torch.randn(B, cfg.score.t_dim, generator=g).cuda()
out = {"workload": "BASELINE configs[4] per-GPU share: B=%d, T=%d, cond tokens %d, %d of 1000 steps timed" % (B, T, S, N)}
for name, cond in (("unconditional", None), ("vipc_conditioned", (pts, img))):
    tr.sample(B, condition=cond); torch.cuda.synchronize()
    t0 = time.perf_counter(); p, e = tr.sample(B, condition=cond); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert bool(torch.isfinite(e).all())
    out[name] = {"ms_per_sde_step": round(1e3 * dt / N, 3), "shapes_per_s_at_1000_steps": round(B / (dt * 1000 / N), 3)}
print(json.dumps(out))
