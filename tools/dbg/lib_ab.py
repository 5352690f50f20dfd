#!/usr/bin/env python3
"""Whole-loop A/B of two builds of libldt_hip.so on one box: runs `sample()` (N SDE steps, B=64, T=256) in a child
process per library, alternating, and prints ms per step (AB_TOKENS / AB_BATCH in the environment pick another workload).   usage: lib_ab.py N libA.so libB.so [libC.so ...] [rounds]"""
import os, subprocess, sys
N = sys.argv[1]
libs = [a for a in sys.argv[2:] if not a.isdigit()]                 # two or more builds ("product" = the in-tree one)
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 3 else 2
child = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import ldt_amd
N = int(sys.argv[1])
T = int(os.environ.get("AB_TOKENS", "256")); B = int(os.environ.get("AB_BATCH", "64"))
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
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["LDT_HIP_LIB"] = l
        out = subprocess.run([sys.executable, "-c", child, N], env=env, capture_output=True, text=True)
        try:
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
        except (ValueError, IndexError):
            print(out.stdout[-500:], out.stderr[-1500:])
            raise
        print("round %d %s: %.4f ms/step" % (r, l, res[l][-1]), flush=True)
print({l: min(v) for l, v in res.items()})
