"""GPU (-m gpu): bit-reproducibility soaks of the hand-scheduled kernels (the screens that lived in tools/dbg).

A mis-counted `s_waitcnt vmcnt`, an LDS read one phase early, an LDS-DMA hand-over without its barrier or an MFMA operand overwritten
in flight do not show as an error bar: they show as outputs that differ from launch to launch under memory load (the round-4 hazard:
profiles/r04_mid_epilogue_hazard.txt was 9e-5 instead of 2.7e-6, different every run).  Every case runs the SAME inputs 50 times at
the production width beside a memory-bound stream on a second HIP stream and requires bit-identical outputs:

  * Score forward, B = 64 x T = 32 (mid-tile GEMMs, LN folded through the loader waves, QKV + self-attention in one launch)
  * Score forward, B = 32 x T = 32 ViPC-conditioned (64 x 64 tiles, q + cross-attention in one launch, per-sample AdaLN rows)
  * Score forward, B = 64 x T = 256 (persistent 256^2 GEMMs, LN folded, QKV + attention in one launch, ring-landed residual)
  * the LN-folded GEMM pair alone at M = 2048 — the shape the round-4 hazard showed on — and at M = 16,384
Reference for the math: model/scorenet/score.py:117-151, model/layers.py:183-229."""
import pytest
import torch

pytestmark = pytest.mark.gpu

REPEATS = 50


class _Load:
    """an uneven memory-bound load on a second stream while the kernels under test run"""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.junk = torch.empty(64 << 20, device="cuda")

    def kick(self, n=2):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.junk.add_(1.0)


@pytest.fixture(scope="module")
def model4():
    """production width (hidden 1024, 16 heads), 4 blocks: every kernel form of the 24-block model, a sixth of its run time"""
    import ldt_amd
    cfg = ldt_amd.airplane_config(latent_tokens=256, sample_N=10, **{"score.num_blocks": 4})
    torch.manual_seed(11)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    score = ldt_amd.Trainer(cfg, ldt_amd.Score(cfg.score), comp, "cuda:0").model
    return cfg, score


def _soak(fn, load, what):
    ref = None
    for it in range(REPEATS):
        load.kick(1 + it % 3)
        outs = fn()
        torch.cuda.synchronize()
        outs = [o.clone() for o in (outs if isinstance(outs, (tuple, list)) else (outs,))]
        if ref is None:
            ref = outs
            assert all(bool(torch.isfinite(o.float()).all()) for o in outs), what
            continue
        for i, (p, q) in enumerate(zip(ref, outs)):
            assert torch.equal(p, q), "%s: output %d differs from the first launch in %d elements at repeat %d" % (what, i, int((p != q).sum()), it)


@pytest.mark.parametrize("B,T,cond", [(64, 32, False), (32, 32, True), (64, 256, False)])
def test_score_forward_is_bit_reproducible_under_load(model4, B, T, cond):
    cfg, score = model4
    g = torch.Generator().manual_seed(100 + B + T)
    x = torch.randn(B, T, cfg.score.z_dim, generator=g).cuda()
    load = _Load()
    if cond:
        condition = (torch.randn(B, cfg.score.hidden_size, 32, generator=g).cuda(), torch.randn(B, cfg.score.t_dim, generator=g).cuda())
        t = (torch.rand(B, generator=g) * 0.98 + 0.01).cuda()
        _soak(lambda: score(x, t, condition=condition), load, "conditioned forward B=%d T=%d" % (B, T))
    else:
        assert score.can_fold(B, T)                                # the production route: LN folded into the GEMM epilogues
        _soak(lambda: score.forward_shared_t(x, 0.37), load, "forward B=%d T=%d" % (B, T))


@pytest.mark.parametrize("M", [2048, 16384])
def test_lnfold_gemm_pair_is_bit_reproducible_under_load(M):
    """producer (fc_o form: + gate x, + residual, emits x (1 + scale) bf16 and the row statistics) -> consumers (MLP-up + GELU, QKV)"""
    from ldt_amd import ops
    from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16
    D = 1024
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, D, generator=g).cuda().to(torch.bfloat16)
    wo = (torch.randn(D, D, generator=g) / 32).cuda().to(torch.bfloat16)
    bo = torch.randn(D, generator=g).cuda()
    x0 = (torch.randn(M, D, generator=g) + 0.5).cuda()
    gate = torch.randn(D, generator=g).cuda()
    sc = (0.3 * torch.randn(D, generator=g)).cuda()
    w2 = (torch.randn(4096, D, generator=g) / 32).cuda().to(torch.bfloat16)
    S, C = torch.randn(4096, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
    w3 = (torch.randn(3072, D, generator=g) / 32).cuda().to(torch.bfloat16)
    S3, C3 = S[:3072].contiguous(), C[:3072].contiguous()
    rows, granule = (256, 256) if M >= 8192 else (32, 32)          # 256-tile kernels / the mid-tile kernels (statistics per 32 columns)

    def run():
        x = x0.clone()
        xs, st = ops.gemm_resid_lnstats(a, wo, bo, x, sc, gate=gate, rows_per_sample=rows, granule=granule)
        y1 = ops.gemm_lnfold(xs, w2, st, S, C, EPI_GELU_BF16)
        y2 = ops.gemm_lnfold(xs, w3, st, S3, C3, EPI_BF16)
        return x, xs, st, y1, y2

    _soak(run, _Load(), "LN-folded GEMM pair, M = %d" % M)
