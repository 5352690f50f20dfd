"""rocprofv3 target: BASELINE configs[4] per-GPU share (ViPC-conditioned sampling, B = 32, T = 32, 32 condition tokens), 60 SDE steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = int(os.environ.get("PROF_STEPS", "60"))
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
g = torch.Generator().manual_seed(5)
cond = (torch.randn(32, cfg.score.hidden_size, 32, generator=g).cuda(), torch.randn(32, cfg.score.t_dim, generator=g).cuda())
for _ in range(2):
    tr.sample(32, condition=cond)
torch.cuda.synchronize()
