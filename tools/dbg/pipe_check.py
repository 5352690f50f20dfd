"""tools/dbg: parity check of the role-specialised persistent kernel experiment (gemm_pipe_exp.hip; needs its wiring — see README — and LDT_GEMM_PIPE=1)."""
import sys, torch
sys.path.insert(0, '.')
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32
def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum() / (b ** 2).sum())
bf = lambda t: t.to(torch.bfloat16)
for (M, N, K) in [(8192, 1024, 1024), (16384, 1024, 576), (4096, 3072, 1024), (8192, 1024, 4096), (16384, 4096, 640)]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = bf(torch.randn(M, K, device="cuda", generator=g)); w = bf(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = x.double() @ w.double().T + bias.double()
    o = ops.gemm_bf16(x, w, bias, EPI_BF16)
    assert rel(o, ref) < 1e-5, ("bf16", M, N, K, rel(o, ref))
    assert torch.equal(o, ops.gemm_bf16(x, w, bias, EPI_BF16))
    assert rel(ops.gemm_bf16(x, w, bias, EPI_F32), ref) < 1e-9
    assert rel(ops.gemm_bf16(x, w, bias, EPI_GELU_BF16), torch.nn.functional.gelu(ref)) < 1e-5
    resid = torch.randn(M, N, device="cuda", generator=g)
    gates = torch.randn(3, 2 * N, device="cuda", generator=g)
    stp = torch.tensor([1], dtype=torch.int32, device="cuda")
    rd = resid.clone()
    ops.gemm_bf16(x, w, bias, EPI_RESID_F32, out=rd, resid=rd, gate=gates[:, N:], gate_sample_stride=0, rows_per_sample=M, step_ptr=stp, gate_step_stride=2 * N)
    assert rel(rd, resid.double() + gates[1, N:].double() * ref) < 1e-9
    gs = torch.randn(M // 256, 3 * N, device="cuda", generator=g)
    rd = resid.clone()
    ops.gemm_bf16(x, w, bias, EPI_RESID_F32, out=rd, resid=rd, gate=gs[:, N:2 * N], gate_sample_stride=3 * N, rows_per_sample=256)
    assert rel(rd, resid.double() + gs[:, N:2 * N].double().repeat_interleave(256, 0) * ref) < 1e-9
    if N == 1024:
        # LN-folded pair: producer (this GEMM) -> consumer (N2 = 3072 / 4096 columns)
        sc = 0.3 * torch.randn(N, device="cuda", generator=g); sh = 0.3 * torch.randn(N, device="cuda", generator=g)
        x0 = torch.randn(M, N, device="cuda", generator=g) * 1.5 + 0.6
        xd = x0.clone()
        xs, stats = ops.gemm_resid_lnstats(x, w, bias, xd, sc, gate=gates[:, N:], gate_sample_stride=0, rows_per_sample=M, step_ptr=stp,
                                           gate_step_stride=2 * N, granule=128)
        xref = x0.double() + gates[1, N:].double() * ref
        assert rel(xd, xref) < 1e-9 and rel(xs, xref * (1 + sc.double())) < 1e-5
        t = xd.double().view(M, N // 128, 128)
        assert rel(stats[..., 0], t.sum(-1).T) < 1e-10 and rel(stats[..., 1], (t ** 2).sum(-1).T) < 1e-10
        xs2, stats2 = ops.gemm_resid_lnstats(x, w, bias, x0.clone(), sc, gate=gates[:, N:], gate_sample_stride=0, rows_per_sample=M, step_ptr=stp,
                                             gate_step_stride=2 * N, granule=128)
        assert torch.equal(xs, xs2) and torch.equal(stats, stats2)
        for N2, epi in ((3072, EPI_BF16), (4096, EPI_GELU_BF16)):
            w2 = bf(torch.randn(N2, N, device="cuda", generator=g) / N ** 0.5); b2 = torch.randn(N2, device="cuda", generator=g)
            S = (w2.double() * (1 + sc.double())).sum(1).float(); C = (w2.double() @ sh.double() + b2.double()).float()
            y = ops.gemm_lnfold(xs, w2, stats, S, C, epi)
            xn = xd.double()
            h = (xn - xn.mean(1, keepdim=True)) / torch.sqrt(xn.var(1, unbiased=False, keepdim=True) + 1e-6) * (1 + sc.double()) + sh.double()
            r2 = h @ w2.double().T + b2.double()
            if epi == EPI_GELU_BF16: r2 = torch.nn.functional.gelu(r2)
            e = rel(y, r2)
            assert e < 3e-5, ("consumer", M, N2, e)
            assert torch.equal(y, ops.gemm_lnfold(xs, w2, stats, S, C, epi))
    print("ok", M, N, K, flush=True)
