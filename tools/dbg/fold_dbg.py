#!/usr/bin/env python3
"""Debug: one Score forward with and without LN folding (same plan otherwise), several shapes."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ldt_amd
from ldt_amd import ops
from ldt_amd._lib import check, lib

def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum() / (b ** 2).sum())

for (hid, heads, blocks, T, B) in ((256, 4, 1, 32, 8), (256, 4, 2, 32, 8), (256, 4, 3, 32, 8), (512, 8, 2, 256, 4), (1024, 16, 2, 256, 8)):
    cfg = ldt_amd.airplane_config(latent_tokens=T, sample_N=20, **{"score.hidden_size": hid, "score.num_heads": heads,
                                                                   "score.num_blocks": blocks, "score.t_dim": 128})
    torch.manual_seed(11)
    m = ldt_amd.Score(cfg.score).cuda()
    ts = torch.linspace(1.0, 1e-6, 20).cuda()
    _, mod = m.time_table(ts)
    fold = m.fold_table(mod)
    x = torch.randn(B, T, 120, device="cuda")
    outs = []
    step = torch.tensor([3], dtype=torch.int32, device="cuda")
    for f in (None, fold):
        plan = m.plan(B, T, mod, m.n_mod, 0, fold=f)
        out = torch.empty_like(x)
        check(lib().ldt_score_forward(ctypes.byref(plan), x.data_ptr(), out.data_ptr(), step.data_ptr(), ops.stream_ptr()), "fwd")
        torch.cuda.synchronize()
        outs.append(out.clone())
        W = m._workspace(B, T)
        print("  fold=%s finite=%s X finite=%s Hb finite=%s stats finite=%s |X|=%.3f" % (f is not None, bool(torch.isfinite(out).all()),
              bool(torch.isfinite(W["X"]).all()), bool(torch.isfinite(W["Hb"].float()).all()), bool(torch.isfinite(W["stats"]).all()), float(W["X"].abs().mean())))
    print("hid=%d blocks=%d T=%d B=%d: fold finite=%s rel(fold, plain)=%.3e fold-table finite=%s" % (
        hid, blocks, T, B, bool(torch.isfinite(fold).all()), rel(outs[1], outs[0]), bool(torch.isfinite(fold).all())), flush=True)
