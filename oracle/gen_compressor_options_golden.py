"""TEST INFRASTRUCTURE ONLY — goldens for Compressor options no shipped YAML uses, captured from the imported reference:
`norm_input: True` + `pre_group: True` (model/Compressor/Network.py:170-174,188-195) and the mixture-of-Gaussians InitialSet
(`max_outputs: None`, model/Compressor/layers.py:17-24,38-42).

    python oracle/gen_compressor_options_golden.py   # writes tests/golden/compressor_options.npz (+ its weights)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import Recorder, randomize_norm_stats, save, sd_np, tiny_cfg  # noqa: E402


def main():
    R.setup()
    from model.Compressor.Network import Compressor
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(17)
    out = {}
    # ---- (a) norm_input + pre_group: 512 points -> 256 pre-groups of 32 -> 8 tokens of 64 neighbours ----------------
    cfg = tiny_cfg()
    cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1      # (keeps the fixture small: two weight sets are stored)
    cfg.compressor.norm_input, cfg.compressor.pre_group = True, True
    cfg.compressor.max_outputs = cfg.compressor.outsize = 96
    torch.manual_seed(5)
    comp = Compressor(cfg.compressor).eval()
    comp.init()
    randomize_norm_stats(comp, g)
    pts = torch.randn(2, 512, 3, generator=g) * torch.tensor([1.0, 0.5, 2.0]) + torch.tensor([0.3, -0.2, 0.1])
    torch.manual_seed(78)
    with Recorder() as rec:
        res = comp(pts)
    kinds = [k for k, _ in rec.draws]
    assert kinds == ["randperm"] * 2 + ["randn"] * cfg.compressor.n_layers, kinds
    out.update(a_pts=pts, a_post_noise=torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0),
               a_all_eps=res["all_eps"], a_set=res["set"], a_max=res["max"], **sd_np(comp.state_dict(), "a::"))
    # ---- (b) mixture InitialSet (max_outputs None): decode and encode ---------------------------------------------------
    cfg = tiny_cfg()
    cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
    cfg.compressor.max_outputs = None
    torch.manual_seed(6)
    comp = Compressor(cfg.compressor).eval()
    comp.init()
    randomize_norm_stats(comp, g)
    geps = torch.randn(2, cfg.compressor.z_scales, cfg.compressor.n_layers * cfg.compressor.z_dim, generator=g)
    torch.manual_seed(79)
    with Recorder() as rec:
        dec = comp.sample((2, 48), given_eps=geps)
    assert [k for k, _ in rec.draws] == ["randn"], [k for k, _ in rec.draws]
    out.update(b_given_eps=geps, b_seed_eps=rec.draws[0][1], b_points=dec)
    pts = torch.randn(2, 64, 3, generator=g)
    torch.manual_seed(80)
    with Recorder() as rec:
        res = comp(pts)
    kinds = [k for k, _ in rec.draws]
    assert kinds == ["randn"] * (1 + cfg.compressor.n_layers), kinds
    out.update(b_pts=pts, b_fwd_seed_eps=rec.draws[0][1],
               b_post_noise=torch.stack([d.transpose(1, 2) for _, d in rec.draws[1:]], 0),
               b_all_eps=res["all_eps"], b_set=res["set"], **sd_np(comp.state_dict(), "b::"))
    # ---- (c) pos_embedding: mlp (per-token position condition) --------------------------------------------------------
    cfg = tiny_cfg()
    cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
    cfg.compressor.pos_embedding = "mlp"
    torch.manual_seed(7)
    comp = Compressor(cfg.compressor).eval()
    comp.init()
    randomize_norm_stats(comp, g)
    pts = torch.randn(2, 64, 3, generator=g)
    torch.manual_seed(81)
    with Recorder() as rec:
        res = comp(pts)
    out.update(c_pts=pts, c_post_noise=torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0),
               c_all_eps=res["all_eps"], c_set=res["set"], **sd_np(comp.state_dict(), "c::"))
    # ---- (d) class_condition: LabelEmbedding into the position condition and the decoder blocks -----------------------
    cfg = tiny_cfg()
    cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
    cfg.compressor.class_condition, cfg.compressor.num_categorys = True, 5
    torch.manual_seed(8)
    comp = Compressor(cfg.compressor).eval()
    comp.init()
    randomize_norm_stats(comp, g)
    label = torch.tensor([3, 1])
    torch.manual_seed(82)
    with Recorder() as rec:
        res = comp(pts, label=label)
    geps = out["b_given_eps"]
    torch.manual_seed(83)
    dec = comp.sample((2, 64), given_eps=geps)                     # decode ignores labels (:251-268): plain-LayerNorm branch of AdaLN-built blocks
    out.update(d_label=label, d_post_noise=torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0),
               d_all_eps=res["all_eps"], d_set=res["set"], d_points=dec, **sd_np(comp.state_dict(), "d::"))
    save("compressor_options", **out)


if __name__ == "__main__":
    main()
