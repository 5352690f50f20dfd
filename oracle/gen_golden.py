"""TEST INFRASTRUCTURE ONLY — captures golden vectors from the imported upstream reference.

Run in the build container (where /root/reference exists):

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

The fixtures are *data* (inputs, weights, expected outputs of the reference's own
code run on CPU); they travel to the GPU box, the reference does not.
All tensors are stored token-major ([B,T,C]); the reference's channels-first
tensors are transposed on capture.  Weights are stored under their reference
state_dict names.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sd_np(sd, prefix="w::"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


class Recorder:
    """Records every torch.randn / randn_like / randperm draw made by the reference."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._o = (torch.randn, torch.randn_like, torch.randperm)
        rec = self.draws

        def randn(*a, **k):
            r = self._o[0](*a, **k); rec.append(("randn", r.clone())); return r

        def randn_like(*a, **k):
            r = self._o[1](*a, **k); rec.append(("randn_like", r.clone())); return r

        def randperm(*a, **k):
            r = self._o[2](*a, **k); rec.append(("randperm", r.clone())); return r

        torch.randn, torch.randn_like, torch.randperm = randn, randn_like, randperm
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.randperm = self._o


def tiny_cfg(N=50):
    cfg = R.load_airplane_cfg(**{
        # sizes the MFMA kernels accept (channels % 64 == 0, head dim 64 / 32 like the shipped config)
        "score.z_scale": 8, "score.hidden_size": 128, "score.num_heads": 2, "score.num_blocks": 2,
        "score.t_dim": 64,
        "compressor.z_scales": 8, "compressor.hidden_dim": 64, "compressor.num_heads": 2, "compressor.p_dim": 32,
        "compressor.n_layers": 3, "compressor.z_dim": 40,
        "compressor.max_outputs": 64, "compressor.outsize": 64, "sde.sample_N": N})
    cfg.data.tr_max_sample_points = 64
    return cfg


def randomize_norm_stats(module, gen):
    """Give BatchNorm running stats / LayerNorm affines / ActNorm non-trivial values so the
    fixtures exercise them (default init is identity)."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
        if isinstance(m, torch.nn.LayerNorm) and m.elementwise_affine:
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    for n, p in module.named_parameters():
        if n.endswith("conv_in.shift") or n.endswith("conv_in.log_scale") or n.endswith("affine_beta"):
            p.data.copy_(torch.randn(p.shape, generator=gen) * 0.2)
        if n.endswith("affine_alpha"):
            p.data.copy_(torch.rand(p.shape, generator=gen) + 0.5)


def main():
    os.makedirs(OUT, exist_ok=True)
    R.setup()
    from model.layers import TimeEmbedding, ResidualBlock, FinalLayer
    from model.scorenet.score import Score
    from model.Compressor.Network import Compressor
    from model.Compressor.layers import LocalGrouper, square_distance, knn_point
    from diffusion.diffusion_continuous import DiffusionVPSDE
    from evaluation.evaluation_metrics import distChamfer
    from trainer.Latent_SDE_Trainer import Trainer
    torch.set_grad_enabled(False)

    # ---- a8 / Q5: time embedding ------------------------------------------------------
    torch.manual_seed(10)
    te = TimeEmbedding(256, 48)
    t = torch.tensor([1.0, 0.5, 1e-3, 1e-6, 0.73519], dtype=torch.float32)
    save("time_embedding", t=t, sinusoid=te.calc_t_emb(t), out=te(t), **sd_np(te.state_dict()))

    # ---- a9-a12 / Q1-Q4: ResidualBlock variants ---------------------------------------
    torch.manual_seed(11)
    C, H, N, M, PD = 64, 2, 8, 16, 24
    g = torch.Generator().manual_seed(5)
    # (i) Score-style: AdaLN, y=None (K/V from modulated x)
    blk = ResidualBlock(C, C, PD, H, norm="layer_norm").eval()
    x = torch.randn(2, C, N); c = torch.randn(2, PD)
    save("resblock_self", x=x.transpose(1, 2), c=c, out=blk(x, None, c).transpose(1, 2), heads=H,
         **sd_np(blk.state_dict()))
    # (ii) Encoder-style: AdaLN, y = raw x
    save("resblock_encoder", x=x.transpose(1, 2), c=c, out=blk(x, x, c).transpose(1, 2), heads=H,
         **sd_np(blk.state_dict()))
    # (iii) ViPC-style cross: AdaLN, y != x (M keys)
    y = torch.randn(2, C, M)
    save("resblock_cross", x=x.transpose(1, 2), y=y.transpose(1, 2), c=c, out=blk(x, y, c).transpose(1, 2),
         heads=H, **sd_np(blk.state_dict()))
    # (iv) Decoder-style: no condition, LayerNorm affine, act=None, y given
    dblk = ResidualBlock(C, C, None, H, norm="layer_norm", act=None).eval()
    randomize_norm_stats(dblk, g)
    save("resblock_decoder", x=x.transpose(1, 2), y=y.transpose(1, 2), out=dblk(x, y, None).transpose(1, 2),
         out_self=dblk(x, None, None).transpose(1, 2), heads=H, **sd_np(dblk.state_dict()))
    # (v) FinalLayer with condition
    fl = FinalLayer(C, 20, PD, "layer_norm").eval()
    save("final_layer", x=x.transpose(1, 2), c=c, out=fl(x, c).transpose(1, 2), **sd_np(fl.state_dict()))

    # ---- a7 / a13: Score tiny (uncond + pts/img conditioned) ---------------------------
    cfg = tiny_cfg()
    torch.manual_seed(0)
    score = Score(cfg.score).eval()
    comp = Compressor(cfg.compressor).eval()
    comp.init()
    randomize_norm_stats(comp, g)
    xs = torch.randn(3, cfg.score.z_scale, cfg.score.z_dim)
    ts = torch.tensor([0.9, 0.31, 1e-6])
    pts_cond = torch.randn(3, cfg.score.hidden_size, 5)           # channels-first as c_net returns it
    img_cond = torch.randn(3, cfg.score.t_dim)
    save("score_tiny", x=xs, t=ts, out=score(xs, ts),
         pts_cond=pts_cond.transpose(1, 2), img_cond=img_cond,
         out_cond=score(xs, ts, condition=(pts_cond, img_cond)),
         hidden_size=cfg.score.hidden_size, num_heads=cfg.score.num_heads, num_blocks=cfg.score.num_blocks,
         t_dim=cfg.score.t_dim, z_dim=cfg.score.z_dim, z_scale=cfg.score.z_scale,
         **sd_np(score.state_dict()))

    # ---- a4-a6 / Q7: VPSDE tables -------------------------------------------------------
    for N in (100, 1000):
        c2 = tiny_cfg(N)
        sde = DiffusionVPSDE(c2.sde)
        tsteps = torch.linspace(1.0, c2.sde.sample_time_eps, N)
        idx = (tsteps * (N - 1) / 1.0).long()
        save("vpsde_tables_N%d" % N, betas=sde.betas, alphas_cump=sde.alphas_cump, timesteps=tsteps, idx=idx,
             var=sde.var(tsteps), std=sde.std(tsteps), g2=sde.g2(tsteps), f=sde.f(tsteps),
             e2int_f=sde.e2int_f(tsteps), beta_start=c2.sde.beta_start, beta_end=c2.sde.beta_end,
             sigma2_0=c2.sde.sigma2_0, time_eps=c2.sde.sample_time_eps)

    # ---- a1-a6 / Q8: Trainer.sample end-to-end + per-step trajectory -------------------
    with R.quiet():
        tr = Trainer(cfg, score, comp, "cpu")
    B, N = 2, cfg.sde.sample_N
    # hook score_fn to capture the per-step (x_in, params)
    steps = []
    orig_model_fwd = score.forward

    def spy(x, t, label=None, condition=None):
        out = orig_model_fwd(x, t, label=label, condition=condition)
        steps.append((x.clone(), t.clone(), out.clone()))
        return out

    score.forward = spy
    torch.manual_seed(1234)
    with Recorder() as rec:
        smp, eps = tr.sample(B)
    score.forward = orig_model_fwd
    kinds = [k for k, _ in rec.draws]
    assert kinds[0] == "randn" and kinds[1:N + 1] == ["randn_like"] * N, kinds[:5]
    assert kinds[N + 1:] == ["randperm"] * B, kinds[N + 1:]      # InitialSet burns B randperms (Q9)
    x0 = rec.draws[0][1]
    noises = torch.stack([d for _, d in rec.draws[1:N + 1]], 0)
    every = 7
    save("trainer_sample_tiny", x0=x0, noises=noises, points=smp, eps=eps, N=N,
         step_ids=np.arange(0, N, every),
         step_x=torch.stack([steps[i][0] for i in range(0, N, every)], 0),
         step_t=torch.stack([steps[i][1] for i in range(0, N, every)], 0),
         step_params=torch.stack([steps[i][2] for i in range(0, N, every)], 0),
         last_x=steps[-1][0], last_params=steps[-1][2],
         **sd_np(comp.state_dict(), "c::"))
    # other predictors on the same draws (a6')
    finals = {}
    for pred in ("reversediffusion", "eulermaruyama", "ddim"):
        it = iter([x0] + list(noises))
        o_randn, o_like = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: next(it)
        torch.randn_like = lambda *a, **k: next(it)
        try:
            finals[pred] = tr.SDE.sample_discrete(
                score_fn=tr.score_fn, N=N, corrector=None, predictor=pred, corrector_steps=1,
                shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, label=None,
                denoise=True, device="cpu", num_samples=B, probability_flow=False, snr=0.01, condition=None)
        finally:
            torch.randn, torch.randn_like = o_randn, o_like
    save("other_predictors", **finals)
    # predictor + AncestralCorrector (corrector_steps=2) and the print_steps trajectory dump, on recorded draws (a6')
    extras = {}
    for tag, kw in (("corr", dict(corrector="ancestral", corrector_steps=2, print_steps=None)),
                    ("print", dict(corrector=None, corrector_steps=1, print_steps=5))):
        torch.manual_seed(4321)
        with Recorder() as rec2:
            out = tr.SDE.sample_discrete(
                score_fn=tr.score_fn, N=N, predictor="ancestral", shape=(cfg.score.z_scale, cfg.score.z_dim),
                time_eps=cfg.sde.sample_time_eps, label=None, denoise=True, device="cpu", num_samples=B,
                probability_flow=False, snr=cfg.sde.snr, condition=None, **kw)
        draws = [d for _, d in rec2.draws]
        extras[tag + "_x0"] = draws[0]
        extras[tag + "_noise"] = torch.stack(draws[1:], 0)
        extras[tag + "_out"] = torch.stack(out, 0) if isinstance(out, list) else out
    save("sampler_extras", snr=cfg.sde.snr, **extras)

    # ---- a14-a19: Compressor decode + encode ---------------------------------------------
    geps = torch.randn(2, cfg.compressor.z_scales, cfg.compressor.n_layers * cfg.compressor.z_dim)
    dec = comp.sample((2, 64), given_eps=geps)
    save("decoder_tiny", given_eps=geps, points=dec)

    pts = torch.randn(2, 64, 3, generator=g)
    pts = pts - pts.mean(1, keepdim=True)
    pts = pts / pts.norm(dim=-1).amax(dim=1)[:, None, None]
    caps = {}
    grp = comp.group
    import model.Compressor.layers as L
    o_cluster = L.cluster

    def cluster_spy(xyz, Ng, k, center=None):
        r = o_cluster(xyz, Ng, k, center)
        caps["fps_idx"], caps["knn_idx"], caps["centers"] = r[1].clone(), r[2].clone(), r[0].clone()
        return r

    L.cluster = cluster_spy
    o_grp_fwd = grp.forward

    def grp_spy(*a, **k):
        r = o_grp_fwd(*a, **k); caps["tokens"] = r[1].clone(); return r

    grp.forward = grp_spy
    torch.manual_seed(77)
    with Recorder() as rec:
        out = comp(pts)
    L.cluster = o_cluster
    grp.forward = o_grp_fwd
    kinds = [k for k, _ in rec.draws]
    assert kinds == ["randperm"] * 2 + ["randn"] * cfg.compressor.n_layers, kinds     # hard part 9
    post_noise = torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0)
    sqd = square_distance(caps["centers"], pts)
    save("compressor_fwd_tiny", pts=pts, post_noise=post_noise, all_eps=out["all_eps"], set=out["set"],
         fps_idx=caps["fps_idx"], knn_idx=caps["knn_idx"], centers=caps["centers"],
         tokens=caps["tokens"].transpose(1, 2), sqdist=sqd,
         mu=torch.stack([p[1].transpose(1, 2) for p in out["posteriors"][1:]], 0),
         logvar=torch.stack([p[2].transpose(1, 2) for p in out["posteriors"][1:]], 0),
         max=out["max"])

    # ---- a20: Chamfer ------------------------------------------------------------------
    a = torch.randn(2, 64, 3, generator=g); b = torch.randn(2, 64, 3, generator=g)
    dl, dr = distChamfer(a, b)
    save("chamfer", a=a, b=b, dl=dl, dr=dr, cd=dl.mean(1) + dr.mean(1))

    # tiny cfg as plain dict for the tests
    import json

    def ns2d(ns):
        return {k: (ns2d(v) if hasattr(v, "__dict__") else v) for k, v in vars(ns).items()}

    d = ns2d(cfg)
    d["log"]["save_path"] = ""
    with open(os.path.join(OUT, "tiny_cfg.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
