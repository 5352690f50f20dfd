"""Whole-loop A/B of an environment switch read once per process: child processes with VAR=1 / VAR=0 alternate on one box and
print ms per SDE step of `sample(B)` (default B=64, T=256; AB_TOKENS / AB_BATCH / AB_VIPC=1 in the environment pick another workload:
AB_TOKENS=32 = the shipped config, AB_TOKENS=32 AB_BATCH=32 AB_VIPC=1 = BASELINE configs[4]'s per-GPU share).   usage: env_ab.py VAR [N_steps] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 40; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, ldt_amd
N = %d
T = int(os.environ.get("AB_TOKENS", "256")); B = int(os.environ.get("AB_BATCH", "64"))
cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
cond = None
if os.environ.get("AB_VIPC") == "1":
    g = torch.Generator().manual_seed(5)
    cond = (torch.randn(B, cfg.score.hidden_size, 32, generator=g).cuda(), torch.randn(B, cfg.score.t_dim, generator=g).cuda())
best = 1e9
for r in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.sample(B, condition=cond); torch.cuda.synchronize()
    if r: best = min(best, time.perf_counter() - t0)
print("%%.4f" %% (best / N * 1e3), flush=True)
''' % (ROOT, N)
res = {"1": [], "0": []}
for rnd in range(rounds):
    for flag in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **{var: flag}), capture_output=True, text=True)
        try:
            res[flag].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(out.stdout[-500:], out.stderr[-1500:]); raise
        print("%s=%s: %.4f ms per SDE step" % (var, flag, res[flag][-1]), flush=True)
for flag in ("1", "0"):
    print("%s=%s: best %.4f  median %.4f" % (var, flag, min(res[flag]), sorted(res[flag])[len(res[flag]) // 2]))
