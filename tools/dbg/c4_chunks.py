"""Compressor encode (BASELINE configs[3], 1024 clouds) as a function of the clouds per forward() call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
cfg = ldt_amd.airplane_config(latent_tokens=256)
torch.manual_seed(0)
comp = ldt_amd.Compressor(cfg.compressor).cuda(); comp.init()
g = torch.Generator().manual_seed(2)
pts = torch.randn(1024, 2048, 3, generator=g); pts = pts - pts.mean(1, keepdim=True); pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()
for chunk in (128, 256, 512, 1024):
    f = lambda: torch.cat([comp(pts[i:i + chunk])["all_eps"] for i in range(0, 1024, chunk)])
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print("encode chunk %4d: %.1f ms = %.0f clouds/s" % (chunk, dt * 1e3, 1024 / dt), flush=True)
for dc in (128, 256, 512):
    comp.decode_chunk = dc
    eps = torch.randn(1024, 256, 120, device="cuda")
    comp.sample((1024, 2048), given_eps=eps); torch.cuda.synchronize()
    t0 = time.perf_counter(); comp.sample((1024, 2048), given_eps=eps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("decode chunk %4d: %.1f ms = %.0f clouds/s" % (dc, dt * 1e3, 1024 / dt), flush=True)
