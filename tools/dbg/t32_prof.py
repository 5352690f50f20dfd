"""rocprofv3 target: the shipped airplane config (32 latent tokens), B = 64 shapes, 60 SDE steps, unconditional (fused loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = int(os.environ.get("PROF_STEPS", "60"))
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
for _ in range(2):
    tr.sample(64)
torch.cuda.synchronize()
