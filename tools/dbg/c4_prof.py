"""rocprofv3 target: one Compressor.forward (encode, which also reconstructs) + one decode of 1024 clouds in a single call each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
cfg = ldt_amd.airplane_config(latent_tokens=256)
torch.manual_seed(0)
comp = ldt_amd.Compressor(cfg.compressor).cuda(); comp.init()
g = torch.Generator().manual_seed(2)
pts = torch.randn(1024, 2048, 3, generator=g); pts = pts - pts.mean(1, keepdim=True); pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()
for _ in range(2):
    eps = comp(pts)["all_eps"]
    comp.sample((1024, 2048), given_eps=eps)
torch.cuda.synchronize()
