#!/usr/bin/env python3
"""How far the host runs ahead of the GPU in back-to-back `Trainer.sample(B)` calls (B = 64, T = 256, N steps): wall time until sample() RETURNS
against wall time until the device has finished — the slack in which the next call's host work (the CPU draw of x0) is hidden.   usage: host_ahead.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
tr.sample(64); torch.cuda.synchronize()
for r in range(3):
    t0 = time.perf_counter(); tr.sample(64); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("call %d: sample() returned after %.3f s, device done after %.3f s (host ahead by %.3f s)" % (r, t1 - t0, t2 - t0, t2 - t1), flush=True)
t0 = time.perf_counter()
for r in range(3): tr.sample(64)
torch.cuda.synchronize()
print("3 calls back to back: %.3f s per call" % ((time.perf_counter() - t0) / 3))
