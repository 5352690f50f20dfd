"""A/B of the ring-landed residual read in the residual GEMMs' epilogue: child processes with LDT_RESID_RING=1/0 alternate
(the switch is read once per process); cold rotating buffers, LN-fold producer at fc_o (K=1024) and mlp.out (K=4096) shapes,
plus the whole forward per SDE step."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from ldt_amd import ops
M, N = 16384, 1024
torch.manual_seed(0)
NB = 6
outs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
gate = torch.randn(N, device="cuda"); sc = torch.randn(N, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
for K in (1024, 4096):
    xs_in = [(torch.randn(M, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(NB)]
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    ts = []
    for rnd in range(3):
        for i in range(NB): ops.gemm_resid_lnstats(xs_in[i], w, b, outs[i], sc, gate=gate, rows_per_sample=256)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(4):
            for i in range(NB): ops.gemm_resid_lnstats(xs_in[i], w, b, outs[i], sc, gate=gate, rows_per_sample=256)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
        for o in outs: o.normal_()
    print("  K=%%d producer: %%s us" %% (K, " ".join("%%.1f" %% t for t in ts)), flush=True)
import ldt_amd
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=40)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
tr.sample(64); torch.cuda.synchronize()
t0 = time.perf_counter(); tr.sample(64); torch.cuda.synchronize()
print("  sample(64), 40 steps: %%.3f ms per SDE step" %% ((time.perf_counter() - t0) / 40 * 1e3), flush=True)
''' % ROOT
for rnd in range(2):
    for flag in ("1", "0"):
        print("LDT_RESID_RING=%s" % flag, flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LDT_RESID_RING=flag), check=True)
