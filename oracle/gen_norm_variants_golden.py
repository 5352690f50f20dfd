"""TEST INFRASTRUCTURE ONLY — writes tests/golden/norm_variants.npz from the imported reference.

`norm` other than layer_norm (tools/utils.py:168-181 get_norm; cfg.score.norm -> model/scorenet/score.py:58,70-97, cfg.compressor.norm ->
model/Compressor/Network.py:114,147-151): `group_norm` (nn.GroupNorm(min(C // 4, 16), C, eps=1e-6) on the channels-first activations, always
affine) and `~` (Identity), for the tiny Score (plain forward + point / image condition) and the tiny Compressor (decode + encode).
`batch_norm` raises upstream ("running_mean should contain N elements not C": its wrapper transposes before BatchNorm1d): recorded as such.

    python oracle/gen_norm_variants_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import Recorder, save, sd_np, tiny_cfg  # noqa: E402


def main():
    R.setup()
    from model.Compressor.Network import Compressor
    from model.scorenet.score import Score
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(77)
    out = {}
    base = tiny_cfg()
    x = torch.randn(2, base.score.z_scale, base.score.z_dim, generator=g)
    t = torch.tensor([0.83, 0.11])
    pts_cond = torch.randn(2, base.score.hidden_size, 8, generator=g)            # (B, hidden, S) channels-first, as ConditionNet returns it
    img = 0.5 * torch.randn(2, base.score.t_dim, generator=g)
    pts = torch.randn(2, 64, 3, generator=g)
    pts = pts - pts.mean(1, keepdim=True)
    geps = torch.randn(2, base.compressor.z_scales, 2 * base.compressor.z_dim, generator=g)
    out.update(x=x, t=t, pts_cond=pts_cond.transpose(1, 2).contiguous(), img_cond=img, pts=pts, given_eps=geps)
    for tag, kind in (("gn", "group_norm"), ("id", None)):
        cfg = tiny_cfg()
        cfg.score.norm = kind
        cfg.compressor.norm = kind
        cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
        torch.manual_seed(41)
        score = Score(cfg.score).eval()
        for m in score.modules():                                                # non-trivial GroupNorm affines
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
        out[tag + "_out"] = score(x, t)
        out[tag + "_out_cond"] = score(x, t, condition=(pts_cond, img))
        out.update(sd_np(score.state_dict(), tag + "s::"))
        torch.manual_seed(42)
        comp = Compressor(cfg.compressor).eval()
        comp.init()
        for m in comp.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
        torch.manual_seed(90)
        out[tag + "_points"] = comp.sample((2, 64), given_eps=geps)
        torch.manual_seed(100)
        with Recorder() as rec:
            res = comp(pts)
        out[tag + "_post_noise"] = torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0)
        out[tag + "_all_eps"], out[tag + "_set"] = res["all_eps"], res["set"]
        out.update(sd_np(comp.state_dict(), tag + "c::"))
    try:                                                                         # the third kind get_norm knows: fails upstream
        cfg = tiny_cfg()
        cfg.score.norm = "batch_norm"
        Score(cfg.score).eval()(x, t)
        out["batch_norm_error"] = torch.zeros(1)
    except RuntimeError as e:
        assert "running_mean should contain" in str(e), e
        out["batch_norm_error"] = torch.ones(1)
    save("norm_variants", **out)


if __name__ == "__main__":
    main()
