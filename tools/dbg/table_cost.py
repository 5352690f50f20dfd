import sys, time
sys.path.insert(0, '.')
import torch, ldt_amd
cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=1000)
torch.manual_seed(0)
m = ldt_amd.Score(cfg.score).cuda()
ts = torch.linspace(1.0, 1e-6, 1000).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, mod = m.time_table(ts); torch.cuda.synchronize(); t1 = time.perf_counter()
    fold = m.fold_table(mod); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("time_table %.1f ms, fold_table %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
