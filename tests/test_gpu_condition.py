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


def test_completion_trainer_sample_and_valsample(tiny_cfg):
    """completion_trainer/Latent_SDE_Trainer.py:147-215: CompletionTrainer.sample(condition={'img','pts'}) returns the decoded
    clouds only and equals Trainer.sample on ConditionNet's output; valsample reduces both clouds to <= 2048 points by FPS,
    samples conditioned on (views, partial cloud) and reports L2_ChamferEval_1000 / F1Score as defined upstream (:41-53)."""
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = copy.deepcopy(tiny_cfg)
    cfg.score.condition = True
    torch.manual_seed(8)
    score = ldt_amd.Score(cfg.score)
    with torch.no_grad():
        _randomize(score, 9)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    ct = ldt_amd.CompletionTrainer(cfg, score, comp, "cuda:0")
    g = torch.Generator().manual_seed(2)
    B, T = 2, cfg.score.z_scale
    npts = cfg.data.tr_max_sample_points                          # (the reference's bmm distChamfer needs equal point counts)
    views = torch.randn(B, 3, 64, 64, generator=g); pc = torch.randn(B, npts, 3, generator=g) * 0.3; part = pc[:, :npts // 2] + 0.01
    x0 = torch.randn(B, T, cfg.score.z_dim, generator=g)
    cond = {"img": views.cuda(), "pts": part.cuda()}
    only_pts = ct.sample(B, condition=cond, x0=x0, seed=3)
    both = ldt_amd.Trainer.sample(ct, B, condition=ct.model.c_net(cond), x0=x0, seed=3)
    assert torch.is_tensor(only_pts) and torch.equal(only_pts, both[0])
    res = ct.valsample([(views, pc, part), (views * 0.5, pc * 1.1, part)])
    smp, ref = res["samples"], res["refs"]
    assert smp.shape == (2 * B, npts, 3) and ref.shape == (2 * B, npts, 3) and res["rate"] > 0
    dl, dr = O.dist_chamfer(smp.cpu(), ref.cpu())
    cd_ref = float((dl.mean() + dr.mean()) * 1000)
    p1, p2 = (dl < 0.001).float().mean(1), (dr < 0.001).float().mean(1)
    f = 2 * p1 * p2 / (p1 + p2); f[torch.isnan(f)] = 0
    assert abs(res["cd"] - cd_ref) <= 1e-4 * abs(cd_ref) and abs(res["f1"] - float(f.mean())) < 1e-6
    from ldt_amd.metrics import F1Score
    fs, q1, q2 = F1Score(ref, ref + 0.001)                       # identical clouds up to a 1e-3 shift: every point within threshold
    assert torch.allclose(fs.cpu(), torch.ones(2 * B)) and torch.allclose(q1.cpu(), torch.ones(2 * B))


def test_condition_cache_not_reused_across_batches(tiny_cfg):
    """Two sample() calls with DIFFERENT partial clouds: ConditionNet's output of the second call lands in the storage the
    caching allocator recycled from the first (same address, shape, version 0); the step-invariant cross-attention K/V
    must be rebuilt, not served from the first batch's cache entry.  Checked against a model that never saw batch 1."""
    import ldt_amd
    cfg = copy.deepcopy(tiny_cfg)
    cfg.score.condition = True

    def make():
        torch.manual_seed(8)
        score = ldt_amd.Score(cfg.score)
        with torch.no_grad():
            _randomize(score, 9)
        comp = ldt_amd.Compressor(cfg.compressor)
        comp.init()
        return ldt_amd.CompletionTrainer(cfg, score, comp, "cuda:0")

    g = torch.Generator().manual_seed(3)
    B, T = 2, cfg.score.z_scale
    views = torch.randn(B, 3, 64, 64, generator=g).cuda()
    part1 = (torch.randn(B, 96, 3, generator=g) * 0.3).cuda()
    part2 = (torch.randn(B, 96, 3, generator=g) * 0.3).cuda()
    x0 = torch.randn(B, T, cfg.score.z_dim, generator=g)
    ct = make()
    s1 = ct.sample(B, condition={"img": views, "pts": part1}, x0=x0, seed=3)
    s2 = ct.sample(B, condition={"img": views, "pts": part2}, x0=x0, seed=3)
    fresh = make().sample(B, condition={"img": views, "pts": part2}, x0=x0, seed=3)
    assert torch.equal(s2, fresh)
    assert not torch.equal(s1, s2)
    # the same holds for repeated Score.forward calls on ConditionNet outputs
    xs, t = torch.randn(B, T, cfg.score.z_dim, generator=g).cuda(), torch.tensor([0.6, 0.3]).cuda()
    m = ct.model
    m(xs, t, condition=m.c_net({"img": views, "pts": part1}))
    o2 = m(xs, t, condition=m.c_net({"img": views, "pts": part2}))
    assert torch.equal(o2, make().model(xs, t, condition={"img": views, "pts": part2}))
