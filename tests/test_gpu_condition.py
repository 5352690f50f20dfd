"""GPU (-m gpu): ViPC ConditionNet (reference model/scorenet/score.py:13-44) and raw-dict conditioning.

  * point branch vs the golden captured from the reference's own module (tests/golden/condition_net_pts.npz);
    bf16 MFMA PreExtraction: relative MSE <= 1e-4
  * image branch (resnet18[:6], PyTorch-ROCm conv2d, fp32) vs the oracle's restatement: <= 1e-6 — torchvision is not
    in the image, so this branch is parity-UNPINNED against the real trunk
  * Score(cfg.condition=True).forward(x, t, condition={'img','pts'}) vs the oracle composed the same way: <= 1e-4
"""
import copy

import pytest
import torch

from conftest import load_golden, rel_mse

pytestmark = pytest.mark.gpu


def test_points_branch_golden():
    import ldt_amd
    a, sds = load_golden("condition_net_pts")
    net = ldt_amd.ConditionNet(int(a["hidden"]), int(a["p_dim"]), patch_size=int(a["patch_size"]), img_condition=False)
    net.load_state_dict(sds["w"], strict=True)
    net = net.cuda()
    pts_cond, img_cond = net({"pts": a["pts"].cuda()})
    assert img_cond == 0. and pts_cond.shape == a["pts_condition"].shape
    assert rel_mse(pts_cond.cpu(), a["pts_condition"]) < 1e-4
    assert net({"img": torch.zeros(1, 3, 8, 8)}) == (0., 0.)          # branch switched off -> the reference's 0. placeholders


def _randomize(net, seed):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(0.2 * torch.randn(m.bias.shape, generator=g))


def test_image_branch_vs_oracle():
    import ldt_amd
    from oracle import ldt_oracle as O
    torch.manual_seed(3)
    net = ldt_amd.ConditionNet(128, 64, patch_size=8)
    with torch.no_grad():
        _randomize(net, 4)
    sd = {"c_net." + k: v.clone() for k, v in net.state_dict().items()}
    img = torch.randn(3, 3, 64, 64)
    ref = O.condition_net_image(sd, "c_net", img)
    net = net.cuda()
    _, out = net({"img": img.cuda()})
    assert out.shape == (3, 64) and rel_mse(out.cpu(), ref) < 1e-6


def test_score_with_raw_condition_dict_vs_oracle(tiny_cfg):
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = copy.deepcopy(tiny_cfg)
    cfg.score.condition = True
    torch.manual_seed(6)
    score = ldt_amd.Score(cfg.score)
    with torch.no_grad():
        _randomize(score, 7)
    sd = {k: v.clone() for k, v in score.state_dict().items()}
    score = score.cuda()
    B, T = 2, cfg.score.z_scale
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(B, T, cfg.score.z_dim, generator=g), torch.tensor([0.7, 0.2])
    pts = torch.randn(B, 96, 3, generator=g)
    img = torch.randn(B, 3, 64, 64, generator=g)
    pc, _, _ = O.condition_net_points(sd, "c_net", pts, cfg.score.z_scale)
    ic = O.condition_net_image(sd, "c_net", img)
    ref = O.score_forward(sd, cfg.score, x, t, condition=(pc, ic))
    out = score(x.cuda(), t.cuda(), condition={"img": img.cuda(), "pts": pts.cuda()})
    assert rel_mse(out.cpu(), ref) < 1e-4
    ref_pts_only = O.score_forward(sd, cfg.score, x, t, condition=(pc, 0.))
    out_pts_only = score(x.cuda(), t.cuda(), condition={"pts": pts.cuda()})
    assert rel_mse(out_pts_only.cpu(), ref_pts_only) < 1e-4 and rel_mse(out_pts_only.cpu(), ref) > 1e-3
    # Trainer.sample with the raw dict == with ConditionNet's output (computed once per call, completion trainer :150-151)
    comp = ldt_amd.Compressor(cfg.compressor)
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    cond = {"img": img.cuda(), "pts": pts.cuda()}
    x0 = torch.randn(B, T, cfg.score.z_dim, generator=g)
    p1, e1 = tr.sample(B, condition=cond, x0=x0, seed=5)
    p2, e2 = tr.sample(B, condition=score.c_net(cond), x0=x0, seed=5)
    assert torch.equal(e1, e2) and torch.equal(p1, p2)
