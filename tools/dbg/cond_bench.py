import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
B, T, S, N = 32, 32, 32, 200
cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
pts = torch.randn(B, 1024, S).cuda(); img = torch.randn(B, 1024).cuda()
for cond in (None, (pts, img)):
    tr.sample(B, condition=cond); torch.cuda.synchronize()
    t0 = time.time(); tr.sample(B, condition=cond); torch.cuda.synchronize(); dt = time.time() - t0
    print("B=%d T=%d N=%d cond=%s: %.3f s  -> %.2f ms/step, %.2f shapes/s at 1000 steps" % (B, T, N, cond is not None, dt, dt / N * 1e3, B / (dt * 1000 / N)))
