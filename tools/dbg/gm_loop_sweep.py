#!/usr/bin/env python3
"""Whole-loop sweep of the grouped tile order of the persistent 256-tile kernels (LDT_GEMM_GM = row panels per group: fused QKV + attention and MLP-up;
default 8): ms per SDE step of sample(64) at T = 256 in alternating child processes.   usage: gm_loop_sweep.py [N] [rounds] [gm ...]"""
import os, subprocess, sys
N = sys.argv[1] if len(sys.argv) > 1 else "30"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gms = sys.argv[3:] or ["8", "1", "2", "4", "16"]          # "q:u" sets LDT_QKV_GM=q (fused QKV + attention) and LDT_GEMM_GM=u (MLP-up) separately
child = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import ldt_amd
N = int(sys.argv[1])
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
best = 1e9
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.sample(64); torch.cuda.synchronize()
    if r: best = min(best, time.perf_counter() - t0)
print("%.4f" % (1e3 * best / N))
'''
res = {g: [] for g in gms}
for r in range(rounds):
    for g in gms:
        env = dict(os.environ, LDT_GEMM_GM=g) if ":" not in g else dict(os.environ, LDT_QKV_GM=g.split(":")[0], LDT_GEMM_GM=g.split(":")[1])
        out = subprocess.run([sys.executable, "-c", child, N], env=env, capture_output=True, text=True)
        try:
            res[g].append(float(out.stdout.strip().splitlines()[-1]))
        except (ValueError, IndexError):
            print(out.stdout[-300:], out.stderr[-1200:]); raise
        print("round %d LDT_GEMM_GM=%s: %.4f ms/step" % (r, g, res[g][-1]), flush=True)
print({g: min(v) for g, v in res.items()})
