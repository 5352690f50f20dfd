import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = int(os.environ.get("N", 60)); T = int(os.environ.get("T", 256))
cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
for rnd in range(2):
    for st in (1, 2, 4):
        os.environ["LDT_STREAMS"] = str(st)
        tr.sample(64); torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.sample(64); torch.cuda.synchronize()
        print("LDT_STREAMS=%d: %.3f ms per SDE step" % (st, (time.perf_counter() - t0) / N * 1e3), flush=True)
