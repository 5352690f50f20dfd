"""GPU (-m gpu): validation metrics (reference evaluation/evaluation_metrics.py) through the C-ABI.

  * pairwise Chamfer matrices, MMD-CD / COV-CD / 1-NN-CD vs the golden captured from the reference's module (fp32;
    P = |x|^2+|y|^2-2xy suffers cancellation of order 1e-7 * |p|^2, so the matrices are compared at rtol 1e-4 and the
    derived counts exactly)
  * approximate-matching EMD vs the oracle's restatement of approxmatch.cu (parity unpinned upstream: CUDA-only):
    relative error <= 2e-3 (v_exp_f32 vs libm expf inside a 9-level annealing)
"""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_pairwise_cd_and_cd_metrics_golden():
    from ldt_amd import metrics as Mx
    a, _ = load_golden("metrics_cd")
    ref, smp = a["ref"].cuda(), a["smp"].cuda()
    M_rs = Mx._pairwise_CD_(ref, smp)
    assert M_rs.shape == a["M_rs"].shape
    assert torch.allclose(M_rs.cpu(), a["M_rs"], rtol=1e-4, atol=1e-6)
    assert torch.allclose(Mx._pairwise_CD_(ref, ref).cpu(), a["M_rr"], rtol=1e-4, atol=1e-6)
    res = Mx.compute_CD_metrics(smp, ref, 5)
    assert torch.allclose(res["mmd-CD"].cpu(), a["mmd_cd"], rtol=1e-4)
    assert float(res["cov-CD"]) == float(a["cov_cd"]) and float(res["1-NN-CD-acc"]) == float(a["one_nn_cd_acc"])
    k3 = Mx.knn(a["M_rr"].cuda(), a["M_rs"].cuda(), a["M_ss"].cuda(), 3, sqrt=True)
    for nm in ("acc", "precision", "recall"):
        assert abs(float(k3[nm]) - float(a["knn3_sqrt_" + nm])) < 1e-6, nm
    dl, dr = Mx.distChamfer(smp[:5], ref[:5])
    assert torch.allclose((dl.mean(1) + dr.mean(1)).cpu(), a["M_rs"].t()[:5, :5].diagonal(), rtol=1e-4, atol=1e-6)
    with pytest.raises(RuntimeError):
        Mx._pairwise_CD_(a["ref"], a["smp"])                       # CPU tensors: no fallback


@pytest.mark.parametrize("n,m", [(96, 96), (300, 300), (256, 128), (100, 230)])
def test_pairwise_cd_ragged_vs_direct(n, m):
    """Sizes that are not multiples of the kernel's chunking, and n != m (the reference's bmm form only takes n == m):
    compared with the direct fp64 definition min ||x_i - y_j||^2."""
    from ldt_amd import ops
    g = torch.Generator().manual_seed(n + m)
    x, y = torch.randn(3, n, 3, generator=g), torch.randn(4, m, 3, generator=g) * 0.7
    got = ops.chamfer_pairwise(x.cuda(), y.cuda()).cpu()
    d = ((x.double()[:, None, :, None, :] - y.double()[None, :, None, :, :]) ** 2).sum(-1)        # [3,4,n,m]
    want = (d.min(dim=2)[0].mean(-1) + d.min(dim=3)[0].mean(-1)).float()
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)


def test_emd_approx_vs_oracle_and_bounds():
    from ldt_amd import metrics as Mx, ops
    from oracle import ldt_oracle as O
    a, _ = load_golden("metrics_cd")
    x, y = a["smp"][:6], a["ref"][:6]
    want = O.emd_approxmatch_cost(x, y)
    got = ops.emd_approx(x.cuda(), y.cuda()).cpu()
    assert torch.allclose(got, want, rtol=2e-3), (got, want)
    emd = Mx.emd_approx(x.cuda(), y.cuda()).cpu()                  # / n, as emd_approx_cuda
    assert torch.allclose(emd, want / x.shape[1], rtol=2e-3)
    assert (emd > 0.97 * a["emd_exact"]).all() and (emd < 1.6 * a["emd_exact"]).all()      # vs the exact assignment
    pw = ops.emd_approx(x[:3].cuda(), y[:4].cuda(), pairwise=True).cpu()                    # all pairs, row-major
    for i in range(3):
        for j in range(4):
            assert abs(float(pw[i, j]) - float(O.emd_approxmatch_cost(x[i:i + 1], y[j:j + 1])[0])) < 2e-3 * float(pw[i, j])
    # unequal sizes follow the kernel's integer mass ratios (approxmatch.cu:6-12)
    g = torch.Generator().manual_seed(5)
    u, v = torch.rand(2, 128, 3, generator=g), torch.rand(2, 64, 3, generator=g)
    assert torch.allclose(ops.emd_approx(u.cuda(), v.cuda()).cpu(), O.emd_approxmatch_cost(u, v), rtol=2e-3)


def test_compute_all_metrics_keys_and_full_size_smoke():
    """2048-point clouds (the shipped evaluation size): all-pairs CD + EMD + 1-NN run and are self-consistent."""
    from ldt_amd import metrics as Mx
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(6, 2048, 3, generator=g).cuda()
    smp = (torch.randn(6, 2048, 3, generator=g) * 1.1).cuda()
    res = Mx.compute_all_metrics(smp, ref, batch_size=64)
    assert sorted(res) == sorted(["mmd-CD", "cov-CD", "mmd-EMD", "cov-EMD", "1-NN-CD-acc", "1-NN-EMD-acc"])
    assert all(torch.isfinite(v).all() for v in res.values())
    same = Mx.compute_all_metrics(ref, ref, batch_size=64)
    assert float(same["cov-CD"]) == 1.0 and float(same["mmd-CD"]) < 1e-5 and float(same["mmd-EMD"]) < 0.02
    both = Mx.EMD_CD(smp, ref, 64, reduced=False)
    M_cd, M_emd = Mx._pairwise_EMD_CD_(smp, ref)
    assert torch.allclose(both["mmd-CD"], M_cd.diagonal(), rtol=1e-5) and torch.allclose(both["mmd-EMD"], M_emd.diagonal(), rtol=1e-5)


def test_valsample_reports_rate_dump_and_metrics(tiny_cfg, tmp_path):
    """valsample (reference :167-226) called the reference's way — a loader of dict batches — on the GPU: sample-rate print,
    smp_ep<epoch>.npy dump, compute_all_metrics against the batches' te_points; plus the int-loader extension."""
    import copy
    import numpy as np
    import ldt_amd
    cfg = copy.deepcopy(tiny_cfg)
    cfg.log.save_path = str(tmp_path)
    cfg.sde.sample_N = 30
    torch.manual_seed(0)
    tr = ldt_amd.Trainer(cfg, ldt_amd.Score(cfg.score), ldt_amd.Compressor(cfg.compressor), "cuda:0")
    P = cfg.data.tr_max_sample_points
    loader = [{"te_points": torch.randn(3, P, 3), "tr_points": torch.randn(3, P, 3), "cate_idx": torch.zeros(3, dtype=torch.long)},
              {"te_points": torch.randn(2, P, 3), "tr_points": torch.randn(2, P, 3), "cate_idx": torch.zeros(2, dtype=torch.long)}]
    res = tr.valsample(test_loader=loader, val_cate=0)
    smp, rate = tr.last_valsample["samples"], tr.last_valsample["rate"]
    assert smp.shape == (5, P, 3) and rate > 0 and bool(torch.isfinite(smp).all())
    assert sorted(res) == sorted("val/gen/" + k for k in ("mmd-CD", "cov-CD", "mmd-EMD", "cov-EMD", "1-NN-CD-acc", "1-NN-EMD-acc"))
    assert all(isinstance(v, float) for v in res.values())
    dumped = np.load(tmp_path / ("smp_ep%d.npy" % tr.epoch))
    assert np.array_equal(dumped, smp.cpu().numpy())
    ref = torch.cat([b["te_points"] for b in loader]).cuda()
    res2 = tr.valsample(2, batch_size=3, ref=ref, save_npy=False)             # extension: 2 batches of 3, cut to len(ref)
    assert tr.last_valsample["samples"].shape == (5, P, 3) and sorted(res2) == sorted(res)


def test_small_elementwise_entries(tiny_cfg):
    """ldt_vpsde_score (Trainer.score_fn's -params / sqrt(var(t)), Latent_SDE_Trainer.py:57-61), ldt_add_f32, ldt_widen_bf16."""
    import ldt_amd
    from ldt_amd import ops
    g = torch.Generator().manual_seed(3)
    params = torch.randn(5, 8, 12, generator=g)
    t = torch.tensor([1.0, 0.5, 1e-3, 1e-6, 0.25])
    sde = ldt_amd.DiffusionVPSDE(tiny_cfg.sde)
    want = -params / torch.sqrt(sde.var(t))[:, None, None]
    got = ops.vpsde_score(params.cuda(), t.cuda(), sde.beta_start, sde.beta_end, sde.sigma2_0).cpu()
    var = sde.var(t)
    tol = (2e-6 + 1.5e-7 / var)[:, None, None]                                # one ulp of exp() moves 1 - exp() by 6e-8 / var relative
    assert bool(((got - want).abs() <= tol * want.abs()).all())
    a, b = torch.randn(7, 33, generator=g), torch.randn(7, 33, generator=g)
    assert torch.equal(ops.add_f32(a.cuda(), b.cuda()).cpu(), a + b)
    w = torch.randn(9, 64, generator=g).bfloat16()
    assert torch.equal(ops.widen_bf16(w.cuda()).cpu(), w.float())
