"""TEST INFRASTRUCTURE ONLY — writes tests/golden/condition_net_pts.npz from the imported reference.

The ViPC `ConditionNet` (model/scorenet/score.py:13-44) point branch: Conv1d 3->128, LocalGrouper(128,
normalize='center'), Conv1d 128->hidden, run by the reference's own module (built with img_condition=False:
its image branch needs torchvision, which this image lacks — that branch stays parity-unpinned).  FPS comes from
the shimmed pointnet2_ops (oracle/ref_import.py), as everywhere else.

    python oracle/gen_condition_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import randomize_norm_stats, save, sd_np  # noqa: E402


def main():
    R.setup()
    from model.scorenet.score import ConditionNet
    import model.Compressor.layers as L
    torch.set_grad_enabled(False)
    torch.manual_seed(21)
    hidden, p_dim, patch = 128, 64, 8
    net = ConditionNet(hidden, p_dim, patch_size=patch, img_condition=False, pt_condition=True).eval()
    g = torch.Generator().manual_seed(8)
    randomize_norm_stats(net, g)
    net.group.affine_alpha.copy_(1 + 0.3 * torch.randn(net.group.affine_alpha.shape, generator=g))
    net.group.affine_beta.copy_(0.2 * torch.randn(net.group.affine_beta.shape, generator=g))
    pts = torch.randn(3, 96, 3, generator=g)
    pts = pts - pts.mean(1, keepdim=True)
    pts = pts / pts.norm(dim=-1).amax(dim=1)[:, None, None]
    seen = {}
    orig = L.cluster

    def spy(xyz, N, k, center=None):
        out = orig(xyz, N, k, center)
        seen["fps_idx"], seen["knn_idx"] = out[1].clone(), out[2].clone()
        return out

    L.cluster = spy
    try:
        pts_cond, img_cond = net({"pts": pts})
    finally:
        L.cluster = orig
    assert img_cond == 0. and pts_cond.shape == (3, hidden, patch)
    save("condition_net_pts", pts=pts, pts_condition=pts_cond, fps_idx=seen["fps_idx"], knn_idx=seen["knn_idx"],
         hidden=hidden, p_dim=p_dim, patch_size=patch, k=seen["knn_idx"].shape[-1], **sd_np(net.state_dict()))


if __name__ == "__main__":
    main()
