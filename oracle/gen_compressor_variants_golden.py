"""Golden vectors for Compressor variants no shipped YAML selects, captured from the reference itself:
`decoder_act` (an activation behind the norms of the decoder blocks' no-condition branch: model/layers.py:224-226, DecoderBlock act=,
model/Compressor/Network.py:55-58,152), `ActNorm: ~` (no conv_in ActNorm: Network.py:121-123,200-201) and `AdaLN: False` (stored and never
read upstream, Network.py:133: the outputs must equal the AdaLN: True model's).

    python oracle/gen_compressor_variants_golden.py   # writes tests/golden/compressor_variants.npz (+ its weights)
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
    g = torch.Generator().manual_seed(23)
    out = {}
    pts = torch.randn(2, 64, 3, generator=g)
    # ONE weight set for every variant (an activation has no parameters; the ActNorm-free model simply lacks conv_in.*)
    base = tiny_cfg()
    base.compressor.n_layers, base.compressor.encoder_layers = 2, 1
    torch.manual_seed(11)
    ref = Compressor(base.compressor).eval()
    ref.init()
    randomize_norm_stats(ref, g)
    weights = {k: v.clone() for k, v in ref.state_dict().items()}
    geps = torch.randn(2, base.compressor.z_scales, base.compressor.n_layers * base.compressor.z_dim, generator=g)
    for tag, act, seed in (("g", "gelu", 11), ("l", "leakyrelu0.2", 12), ("h", "hardswish", 13), ("r", "anything-else-is-relu", 14)):
        # ---- decoder_act: decode (every decoder block runs its no-condition branch) and encode (the posterior blocks too) ----------
        cfg = tiny_cfg()
        cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
        cfg.compressor.decoder_act = act
        comp = Compressor(cfg.compressor).eval()
        comp.load_state_dict(weights, strict=True)
        comp.init()
        torch.manual_seed(90 + seed)
        dec = comp.sample((2, 64), given_eps=geps)
        torch.manual_seed(100 + seed)
        with Recorder() as rec:
            res = comp(pts)
        out.update({tag + "_points": dec, tag + "_post_noise": torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0),
                    tag + "_all_eps": res["all_eps"], tag + "_set": res["set"]})
    # ---- ActNorm: ~ (and AdaLN: False, a dead flag) -------------------------------------------------------------------
    cfg = tiny_cfg()
    cfg.compressor.n_layers, cfg.compressor.encoder_layers = 2, 1
    cfg.compressor.ActNorm = None
    cfg.compressor.AdaLN = False
    comp = Compressor(cfg.compressor).eval()
    assert not any(k.startswith("conv_in.") for k in comp.state_dict())
    comp.load_state_dict({k: v for k, v in weights.items() if not k.startswith("conv_in.")}, strict=True)
    comp.init()
    torch.manual_seed(115)
    with Recorder() as rec:
        res = comp(pts)
    out.update(n_post_noise=torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0),
               n_all_eps=res["all_eps"], n_set=res["set"], given_eps=geps, **sd_np(weights, "w::"))
    out["pts"] = pts
    save("compressor_variants", **out)


if __name__ == "__main__":
    main()
