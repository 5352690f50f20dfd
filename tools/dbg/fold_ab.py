#!/usr/bin/env python3
"""A/B of LN folding at the bench workload (B=64, T=256): interleaved rounds in one process, N SDE steps each."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ldt_amd
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
B = 64
res = {"0": [], "1": []}
for rnd in range(4):
    for mode in ("0", "1"):
        os.environ["LDT_LN_FOLD"] = mode
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pts, eps = tr.sample(B)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rnd:
            res[mode].append(dt)
        print("round %d fold=%s: %.3f s (%.3f ms/step) finite=%s" % (rnd, mode, dt, 1e3 * dt / N, bool(torch.isfinite(eps).all())), flush=True)
for mode in ("0", "1"):
    os.environ["LDT_LN_FOLD"] = mode
    roof, attn, kernels = bench.roofline_pass(tr, cfg, B)
    print("fold=%s kernels: %s" % (mode, json.dumps(kernels)), flush=True)
print(json.dumps({k: min(v) for k, v in res.items()}))
