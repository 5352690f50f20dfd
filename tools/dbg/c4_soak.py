"""Race screen for the Compressor kernels at the C4 shape: encode 1024 clouds three times with the same posterior noise, then decode the
latents three times; outputs must be bit-identical run to run (fused grouper with atomicMax groups, kNN candidate path, one-wave FPS,
resident attention + out-projection, the MLP kernel's next-level projection).  T = 256 and the shipped T = 32 (k = 128 neighbours)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ldt_amd
for T in (256, 32):
    cfg = ldt_amd.airplane_config(latent_tokens=T)
    torch.manual_seed(0)
    comp = ldt_amd.Compressor(cfg.compressor).cuda(); comp.init()
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(1024, 2048, 3, generator=g); pts = pts - pts.mean(1, keepdim=True); pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()
    noise = [torch.randn(1024, T, 20, generator=g) for _ in range(6)]
    outs = [comp(pts, post_noise=noise) for _ in range(3)]
    for k in ("all_eps", "set", "fps_idx"):
        same = all(torch.equal(outs[0][k], o[k]) for o in outs[1:])
        print("T=%d encode %s bit-identical over 3 runs: %s" % (T, k, same), flush=True)
    dec = [comp.sample((1024, 2048), given_eps=outs[0]["all_eps"]) for _ in range(3)]
    print("T=%d decode bit-identical over 3 runs: %s; finite: %s" % (T, all(torch.equal(dec[0], d) for d in dec[1:]), bool(torch.isfinite(dec[0]).all())), flush=True)
