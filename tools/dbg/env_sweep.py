#!/usr/bin/env python3
"""Whole-loop A/B of arbitrary environment settings: each arm is "VAR=val,VAR2=val2" (or "-" for none); ms per SDE step of sample(B) at the
headline shape (AB_TOKENS / AB_BATCH pick another) in alternating child processes on one box.   usage: env_sweep.py N rounds arm [arm ...]"""
import os, subprocess, sys
N, rounds, arms = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
child = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import ldt_amd
N = int(sys.argv[1]); T = int(os.environ.get("AB_TOKENS", "256")); B = int(os.environ.get("AB_BATCH", "64"))
cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
best = 1e9
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.sample(B); torch.cuda.synchronize()
    if r: best = min(best, time.perf_counter() - t0)
print("%.4f" % (1e3 * best / N))
'''
res = {a: [] for a in arms}
for r in range(rounds):
    for a in arms:
        env = dict(os.environ)
        if a != "-":
            env.update(dict(kv.split("=", 1) for kv in a.split(",")))
        out = subprocess.run([sys.executable, "-c", child, N], env=env, capture_output=True, text=True)
        try:
            res[a].append(float(out.stdout.strip().splitlines()[-1]))
        except (ValueError, IndexError):
            print(out.stdout[-300:], out.stderr[-1200:]); raise
        print("round %d %-40s %.4f ms/step" % (r, a, res[a][-1]), flush=True)
print({a: min(v) for a, v in res.items()})
