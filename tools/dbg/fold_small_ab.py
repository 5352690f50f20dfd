"""tools/dbg: shipped 32-token config (B = 64: M = 2048 rows), ms per SDE step with the small-batch LN folding on / off (LDT_LN_FOLD_SMALL)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
N = int(os.environ.get("N", 200)); B = int(os.environ.get("B", 64))
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
outs = {}
for rnd in range(3):
    for flag in ("1", "0"):
        os.environ["LDT_LN_FOLD_SMALL"] = flag
        torch.manual_seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pts, eps = tr.sample(B, seed=3)
        torch.cuda.synchronize()
        outs[flag] = eps
        if rnd: print("LDT_LN_FOLD_SMALL=%s: %.3f ms per SDE step (can_fold=%s)" % (flag, (time.perf_counter() - t0) / N * 1e3, score.can_fold(B, 32)), flush=True)
d = ((outs["0"].double() - outs["1"].double()) ** 2).sum() / (outs["0"].double() ** 2).sum()
print("rel-MSE folded vs LayerNorm kernels after %d steps: %.3e; fold ratio seen %s" % (N, d.item(), getattr(score, "fold_ratio_seen", None)))
