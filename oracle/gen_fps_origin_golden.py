"""TEST INFRASTRUCTURE ONLY — writes tests/golden/compressor_fwd_origin.npz.

Runs only in the build container (imports /root/reference through oracle/ref_import.py).  The reference's
`Compressor.forward` (model/Compressor/Network.py:188-249) on clouds that CONTAIN POINTS NEAR THE ORIGIN, with the FPS stub
following upstream pointnet2_ops' rule (points with |p|^2 <= 1e-3 are never selected and never update their distance —
ADVICE r2 / SURVEY §8c; the library itself is not vendored, so FPS parity stays unpinned): the centres differ from the
vendored twin's, and so do the tokens, posteriors and latents captured here.  Weights: the Compressor of
tests/golden/trainer_sample_tiny.npz.

    python oracle/gen_fps_origin_golden.py
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
    import model.Compressor.layers as L
    cfg = tiny_cfg()
    z = np.load(os.path.join(OUT, "trainer_sample_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("c::")}
    torch.manual_seed(0)
    comp = Compressor(cfg.compressor)
    comp.load_state_dict(sd, strict=True)
    comp.eval(); comp.init()
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(31)
    pts = torch.randn(3, 64, 3, generator=g)
    pts = pts - pts.mean(1, keepdim=True)
    pts = pts / pts.norm(dim=-1).amax(dim=1)[:, None, None]
    # a cluster of points inside the 1e-3 ball (|p| <= 0.0316), placed where the twin WOULD pick one of them early:
    # the rest of the cloud is pushed to one side, so the origin cluster is far from everything already chosen
    pts[:, :, 0] = pts[:, :, 0] * 0.3 + 0.65
    pts[:, 5:29:4] = torch.randn(3, 6, 3, generator=g) * 0.012
    assert int(((pts ** 2).sum(-1) <= 1e-3).sum()) >= 12
    from oracle import ldt_oracle as O
    idx_up, idx_twin = O.fps(pts, cfg.compressor.z_scales, skip_near_origin=True), O.fps(pts, cfg.compressor.z_scales, skip_near_origin=False)
    assert not torch.equal(idx_up, idx_twin), "the cloud does not tell the two FPS rules apart"
    caps = {}
    o_cluster = L.cluster

    def cluster_spy(xyz, Ng, k, center=None):
        r = o_cluster(xyz, Ng, k, center)
        caps["fps_idx"], caps["knn_idx"] = r[1].clone(), r[2].clone()
        return r

    L.cluster = cluster_spy
    torch.manual_seed(78)
    with Recorder() as rec:
        out = comp(pts)
    L.cluster = o_cluster
    assert torch.equal(caps["fps_idx"].long(), idx_up)
    post_noise = torch.stack([d.transpose(1, 2) for k, d in rec.draws if k == "randn"], 0)
    save("compressor_fwd_origin", pts=pts, post_noise=post_noise, all_eps=out["all_eps"], set=out["set"],
         fps_idx=caps["fps_idx"], fps_idx_twin=idx_twin, knn_idx=caps["knn_idx"])


if __name__ == "__main__":
    main()
