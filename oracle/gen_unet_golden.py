"""TEST INFRASTRUCTURE ONLY — writes tests/golden/score_unet_tiny.npz from the imported reference.

`Score` with `unet: True` (model/scorenet/score.py:67-83,138-146): num_blocks//2 up blocks, a mid block, num_blocks//2 down blocks
(ResidualBlock(2C, 2C, t_dim, heads, dim_out=C): conv shortcut + adaLN1/adaLN2, model/layers.py:155-176,216-218).

    python oracle/gen_unet_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import save, sd_np, tiny_cfg  # noqa: E402


def main():
    R.setup()
    from model.scorenet.score import Score
    torch.set_grad_enabled(False)
    cfg = tiny_cfg()
    cfg.score.unet = True
    cfg.score.num_blocks = 2
    torch.manual_seed(31)
    score = Score(cfg.score).eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, cfg.score.z_scale, cfg.score.z_dim, generator=g)
    t = torch.tensor([0.83, 0.11])
    img = 0.5 * torch.randn(2, cfg.score.t_dim, generator=g)
    save("score_unet_tiny", x=x, t=t, out=score(x, t), img_cond=img, out_img=score(x, t, condition=(None, img)),
         num_blocks=cfg.score.num_blocks, **sd_np(score.state_dict()))


if __name__ == "__main__":
    main()
