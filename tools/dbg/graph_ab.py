"""sample(64) at the bench shape: stream launches vs one captured hipGraph per SDE step; LDT_STREAMS sub-batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = 60
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
for rnd in range(2):
    for ug in (0, 1):
        tr.sample(64, use_graph=ug); torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.sample(64, use_graph=ug); torch.cuda.synchronize()
        print("use_graph=%d: %.3f ms per SDE step" % (ug, (time.perf_counter() - t0) / N * 1e3), flush=True)
