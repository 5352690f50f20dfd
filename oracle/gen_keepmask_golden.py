"""TEST INFRASTRUCTURE ONLY — golden for Compressor.sample with num_points < max_outputs (InitialSet keeps a random subset
of its learned prior rows: model/Compressor/ops.py:6-14, layers.py:26-37), captured from the imported reference.

    python oracle/gen_keepmask_golden.py     # writes tests/golden/decoder_keepmask.npz

Weights: the Compressor of tests/golden/trainer_sample_tiny.npz (c:: entries).  The B randperm draws are recorded, and the
seed is stored so that a seeded CPU generator reproduces them (`reference_rng` mode).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import OUT, Recorder, save, tiny_cfg  # noqa: E402


def main():
    R.setup()
    from model.Compressor.Network import Compressor
    torch.set_grad_enabled(False)
    cfg = tiny_cfg()
    z = np.load(os.path.join(OUT, "trainer_sample_tiny.npz"))
    comp = Compressor(cfg.compressor).eval()
    comp.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("c::")}, strict=True)
    comp.init()
    g = torch.Generator().manual_seed(31)
    B, npts = 3, 40                                                 # max_outputs = 64
    geps = torch.randn(B, cfg.compressor.z_scales, cfg.compressor.n_layers * cfg.compressor.z_dim, generator=g)
    seed = 2024
    torch.manual_seed(seed)
    with Recorder() as rec:
        pts = comp.sample((B, npts), given_eps=geps)
    kinds = [k for k, _ in rec.draws]
    assert kinds == ["randperm"] * B, kinds
    perms = torch.stack([d for _, d in rec.draws], 0)
    save("decoder_keepmask", given_eps=geps, points=pts, perms=perms, keep_mask=(perms < npts), num_points=npts, seed=seed)


if __name__ == "__main__":
    main()
