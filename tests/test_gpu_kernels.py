"""GPU (-m gpu): each HIP kernel through the C-ABI vs the CPU oracle / an fp32 torch-CPU restatement.

Tolerances.  Inputs to the MFMA kernels are bf16-rounded FIRST, so GEMM / attention differ from the fp32
reference only by accumulation order and the bf16 rounding of the output: rel-MSE <= 1e-5 (bf16 out:
(2^-9)^2/3 ~ 1.3e-6) and <= 1e-9 for fp32 outputs.  fp32 kernels (sgemm, LN statistics, sampler update) are
held to fp32 round-off; the ancestral update is bit-exact."""
import numpy as np
import pytest
import torch

from conftest import rel_mse

pytestmark = pytest.mark.gpu

ops = None
O = None


@pytest.fixture(scope="module", autouse=True)
def _mods():
    global ops, O
    assert torch.cuda.is_available()
    from ldt_amd import ops as _ops
    from oracle import ldt_oracle as _O
    ops, O = _ops, _O
    torch.backends.cuda.matmul.allow_tf32 = False


def bf(x):
    return x.to(torch.bfloat16).float()


def dev(x, dt=None):
    return x.to("cuda", dt) if dt else x.to("cuda")


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (16, 120, 1024), (1000, 3072, 1024),
                                   (77, 64, 192), (2048, 1024, 4096), (130, 132, 64)])
def test_gemm_all_epilogues(M, N, K):
    from ldt_amd._lib import EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RELU_BF16, EPI_RESID_F32
    g = torch.Generator().manual_seed(M * 7 + N)
    x = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + bias.double()
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(bias)
    out = ops.gemm_bf16(xd, wd, bd, EPI_F32)
    assert rel_mse(out.cpu(), ref) < 1e-9
    out = ops.gemm_bf16(xd, wd, bd, EPI_BF16)
    assert rel_mse(out.float().cpu(), ref) < 1e-5
    out = ops.gemm_bf16(xd, wd, bd, EPI_GELU_BF16)
    assert rel_mse(out.float().cpu(), torch.nn.functional.gelu(ref)) < 1e-5
    skip = bf(torch.randn(M, N, generator=g))
    out = ops.gemm_bf16(xd, wd, bd, EPI_RELU_BF16, skip=dev(skip, torch.bfloat16))
    assert rel_mse(out.float().cpu(), torch.relu(ref + skip.double())) < 1e-5
    out = ops.gemm_bf16(xd, wd, bd, EPI_RELU_BF16)
    assert rel_mse(out.float().cpu(), torch.relu(ref)) < 1e-5
    # residual + per-sample gate (rows_per_sample = 8 if it divides M)
    rps = 8 if M % 8 == 0 else M
    S = M // rps
    resid = torch.randn(M, N, generator=g); gate = torch.randn(S, 3 * N, generator=g)
    rd = dev(resid.clone())
    ops.gemm_bf16(xd, wd, bd, EPI_RESID_F32, out=rd, resid=rd, gate=dev(gate)[:, N:2 * N], gate_sample_stride=3 * N,
                  rows_per_sample=rps)
    gref = resid.double() + gate[:, N:2 * N].double().repeat_interleave(rps, 0) * ref
    assert rel_mse(rd.cpu(), gref) < 1e-9
    rd = dev(resid.clone())
    ops.gemm_bf16(xd, wd, bd, EPI_RESID_F32, out=rd, resid=rd)
    assert rel_mse(rd.cpu(), resid.double() + ref) < 1e-9


@pytest.mark.parametrize("M,N,K", [(2048, 4096, 1024), (2048, 3072, 1024), (2048, 1024, 4096), (1024, 4096, 1024), (1024, 1024, 1024),
                                   (2048, 4096, 64), (2048, 4096, 128), (2048, 4096, 192), (1152, 1536, 64), (1152, 1536, 192),
                                   (1000, 2304, 320), (1960, 4096, 320), (4096, 1024, 1024),
                                   # 64 x 128 tiles (4-stage ring, K-tile-deep register double buffering): 1..5 K-tiles, ragged rows
                                   (2048, 1024, 64), (2048, 1024, 128), (2048, 1024, 192), (2048, 1024, 256), (2048, 1024, 320), (1000, 1024, 448),
                                   # 128 x 192 tiles (bf16-output epilogues; the fp32 ones of these shapes take another form)
                                   (2048, 3072, 64), (2048, 3072, 192), (1990, 3072, 320),
                                   # 64 x 64 tiles (6-stage ring): fewer K-tiles than stages, exactly as many, more; ragged rows
                                   (1024, 1024, 64), (1024, 1024, 128), (1024, 1024, 384), (1024, 1024, 448), (1000, 1024, 704)])
def test_gemm_mid_kernel(M, N, K):
    """The mid-size tile kernel (csrc/gemm_mid.hip: 128 x 256 / 128 x 128 / 64 x 128 tiles, loader waves, 3- / 4-stage ring) on the shapes the launcher hands
    it — incl. 1 / 2 / 3 / 5 K-tiles (prologue and drain paths of the ring), a ragged last row tile, a step-indexed shared gate, a per-sample
    gate — vs fp64 on the same bf16 operands."""
    from ldt_amd._lib import EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + bias.double()
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(bias)
    assert rel_mse(ops.gemm_bf16(xd, wd, bd, EPI_F32).cpu(), ref) < 1e-9
    o1 = ops.gemm_bf16(xd, wd, bd, EPI_BF16)
    assert rel_mse(o1.float().cpu(), ref) < 1e-5
    assert torch.equal(o1, ops.gemm_bf16(xd, wd, bd, EPI_BF16))                  # no race: identical twice
    assert rel_mse(ops.gemm_bf16(xd, wd, bd, EPI_GELU_BF16).float().cpu(), torch.nn.functional.gelu(ref)) < 1e-5
    resid = torch.randn(M, N, generator=g)
    # shared gate picked by a device-side step counter (unconditional sampling: mod[step][...])
    gates = torch.randn(3, 2 * N, generator=g)
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    rd = dev(resid.clone())
    ops.gemm_bf16(xd, wd, bd, EPI_RESID_F32, out=rd, resid=rd, gate=dev(gates)[:, N:], gate_sample_stride=0, rows_per_sample=M,
                  step_ptr=step, gate_step_stride=2 * N)
    assert rel_mse(rd.cpu(), resid.double() + gates[2, N:].double() * ref) < 1e-9
    # per-sample gate (conditional sampling), rows_per_sample = 8
    if M % 8 == 0:
        gate = torch.randn(M // 8, 3 * N, generator=g)
        rd = dev(resid.clone())
        ops.gemm_bf16(xd, wd, bd, EPI_RESID_F32, out=rd, resid=rd, gate=dev(gate)[:, N:2 * N], gate_sample_stride=3 * N, rows_per_sample=8)
        assert rel_mse(rd.cpu(), resid.double() + gate[:, N:2 * N].double().repeat_interleave(8, 0) * ref) < 1e-9


@pytest.mark.parametrize("M,D,N2,gelu,granule", [(512, 1024, 768, False, 256), (256, 512, 2048, True, 256), (768, 256, 256, False, 256),
                                                 (66560, 256, 256, True, 256),     # 260 tiles: persistent workgroups take a 2nd tile
                                                 # small batches (statistics per 32 columns): the mid-size tile kernel's folded forms
                                                 # (csrc/gemm_mid.hip: 64x128 / 128x128 producers, 128x192 / 128x256 / 128x128 consumers)
                                                 (2048, 1024, 3072, False, 32), (2048, 1024, 4096, True, 32), (1024, 1024, 1024, False, 32),
                                                 (4096, 1024, 1024, False, 32)])
def test_gemm_lnfold_pair(M, D, N2, gelu, granule):
    """LN folding (include/ldt_hip.h): residual GEMM that also emits xs = x(1+scale) + row statistics, then the
    projection that applies the LayerNorm algebraically in its epilogue — against LayerNorm -> modulate -> Linear in
    fp64 on the same bf16 weights.  The row mean is deliberately NOT small (|mean| ~ 0.5 std)."""
    from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16
    g = torch.Generator().manual_seed(M + D + N2)
    K1 = 512
    a = bf(torch.randn(M, K1, generator=g)); wo = bf(torch.randn(D, K1, generator=g) / K1 ** 0.5); bo = torch.randn(D, generator=g)
    x0 = torch.randn(M, D, generator=g) * 1.5 + 0.6
    rps = 128 if M < 4096 else M // 2
    gate = torch.randn(M // rps, D, generator=g)
    sc = 0.3 * torch.randn(D, generator=g); sh = 0.3 * torch.randn(D, generator=g)
    w2 = bf(torch.randn(N2, D, generator=g) / D ** 0.5); b2 = torch.randn(N2, generator=g)
    # ---- producer
    xd = dev(x0.clone())
    xs, stats = ops.gemm_resid_lnstats(dev(a, torch.bfloat16), dev(wo, torch.bfloat16), dev(bo), xd, dev(sc), gate=dev(gate),
                                       gate_sample_stride=D, rows_per_sample=rps, granule=granule)
    xref = x0.double() + gate.double().repeat_interleave(rps, 0) * (a.double() @ wo.double().T + bo.double())
    assert rel_mse(xd.cpu(), xref) < 1e-9
    assert rel_mse(xs.float().cpu(), xref * (1 + sc.double())) < 1e-5
    st = stats.cpu().double()
    assert stats.shape == (D // granule, M, 2)
    tiles = xd.cpu().double().view(M, D // granule, granule)
    assert rel_mse(st[..., 0], tiles.sum(-1).T) < 1e-10 and rel_mse(st[..., 1], (tiles ** 2).sum(-1).T) < 1e-10
    again = ops.gemm_resid_lnstats(dev(a, torch.bfloat16), dev(wo, torch.bfloat16), dev(bo), dev(x0.clone()), dev(sc), gate=dev(gate),
                                   gate_sample_stride=D, rows_per_sample=rps, granule=granule)
    assert torch.equal(again[0], xs) and torch.equal(again[1], stats)          # fixed summation order: reproducible
    # ---- consumer
    S = (w2.double() * (1 + sc.double())).sum(1).float(); C = (w2.double() @ sh.double() + b2.double()).float()
    y = ops.gemm_lnfold(xs, dev(w2, torch.bfloat16), stats, dev(S), dev(C), EPI_GELU_BF16 if gelu else EPI_BF16)
    xn = xd.cpu().double()
    h = (xn - xn.mean(1, keepdim=True)) / torch.sqrt(xn.var(1, unbiased=False, keepdim=True) + 1e-6) * (1 + sc.double()) + sh.double()
    ref = h @ w2.double().T + b2.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    # unfused path on the same inputs for comparison: LN kernel -> bf16 h -> GEMM
    hb = ops.layernorm_modulate(xd, shift=dev(sh), scale=dev(sc), rows_per_sample=M)
    y0 = ops.gemm_bf16(hb, dev(w2, torch.bfloat16), dev(b2), EPI_GELU_BF16 if gelu else EPI_BF16)
    e_fold, e_ln = rel_mse(y.float().cpu(), ref), rel_mse(y0.float().cpu(), ref)
    assert e_fold < 3e-5 and e_fold < 4 * e_ln + 1e-6, (e_fold, e_ln)


@pytest.mark.parametrize("ratio", [0.0, 2.0, 4.0, 8.0])
def test_gemm_lnfold_error_law_vs_row_mean(ratio):
    """The folded form rounds xs = x (1 + scale) to bf16 BEFORE the row mean is removed, so its error variance is
    (1 + mu^2 / sigma^2) x that of rounding the centred value (DESIGN.md §4 "LN folding").  Rows with |mean| / std = ratio:
    the measured error follows the law (within 2x), stays below the 1e-4 parity bar up to |mean| = 4 std, and the row-offset
    monitor the sampler consults (ops.fold_mean_ratio -> Score.can_fold) reports mu^2 / sigma^2."""
    from ldt_amd._lib import EPI_BF16
    M, D, N2, K1 = 512, 1024, 1024, 256
    g = torch.Generator().manual_seed(int(ratio * 10) + 5)
    a = bf(torch.randn(M, K1, generator=g)); wo = bf(torch.randn(D, K1, generator=g) / K1 ** 0.5 * 0.05); bo = torch.zeros(D)
    x0 = torch.randn(M, D, generator=g) + ratio                        # sigma = 1, row mean = ratio
    sc = 0.2 * torch.randn(D, generator=g); sh = 0.2 * torch.randn(D, generator=g)
    w2 = bf(torch.randn(N2, D, generator=g) / D ** 0.5); b2 = torch.randn(N2, generator=g)
    xd = dev(x0.clone())
    xs, stats = ops.gemm_resid_lnstats(dev(a, torch.bfloat16), dev(wo, torch.bfloat16), dev(bo), xd, dev(sc))
    S = (w2.double() * (1 + sc.double())).sum(1).float(); C = (w2.double() @ sh.double() + b2.double()).float()
    y = ops.gemm_lnfold(xs, dev(w2, torch.bfloat16), stats, dev(S), dev(C), EPI_BF16)
    xn = xd.cpu().double()
    h = (xn - xn.mean(1, keepdim=True)) / torch.sqrt(xn.var(1, unbiased=False, keepdim=True) + 1e-6) * (1 + sc.double()) + sh.double()
    ref = h @ w2.double().T + b2.double()
    hb = ops.layernorm_modulate(xd, shift=dev(sh), scale=dev(sc), rows_per_sample=M)
    y0 = ops.gemm_bf16(hb, dev(w2, torch.bfloat16), dev(b2), EPI_BF16)
    e_fold, e_ln = rel_mse(y.float().cpu(), ref), rel_mse(y0.float().cpu(), ref)
    # the bf16 OUTPUT rounding is common to both; what differs is the operand rounding, which obeys the law
    out_round = rel_mse(ref.float().bfloat16().double(), ref)
    law = (1 + ratio ** 2) * max(e_ln - out_round, 2e-7) + out_round
    print("LN-fold error law, |mean|/std = %g: folded %.2e, LayerNorm kernel %.2e, law predicts %.2e" % (ratio, e_fold, e_ln, law))
    assert e_fold < 2.0 * law + 1e-6, (e_fold, e_ln, law)
    if ratio <= 4.0:
        assert e_fold < 1e-4
    mon = ops.fold_mean_ratio(stats, D)
    assert 0.85 * ratio ** 2 <= mon <= 1.5 * ratio ** 2 + 0.05, mon      # (the MAX over 512 rows of a 1024-sample estimate)


def test_gemm_lnfold_rejects_bad_shapes():
    from ldt_amd._lib import LdtHipError
    x = torch.zeros(256, 512, dtype=torch.bfloat16, device="cuda"); w = torch.zeros(300, 512, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(LdtHipError):
        ops.gemm_resid_lnstats(x, w, None, torch.zeros(256, 300, device="cuda"), torch.zeros(300, device="cuda"))


def test_gemm_identity_asymmetric():
    """A = I with an ASYMMETRIC B catches a transposed C write (guide §3)."""
    from ldt_amd._lib import EPI_F32
    K = N = 128
    x = torch.eye(128, K)
    w = bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 100)
    out = ops.gemm_bf16(dev(x, torch.bfloat16), dev(w, torch.bfloat16), None, EPI_F32)
    assert torch.equal(out.cpu(), w.T.contiguous())


def test_gemm_rejects_bad_shapes():
    from ldt_amd._lib import LdtHipError
    x = torch.zeros(8, 72, dtype=torch.bfloat16, device="cuda"); w = torch.zeros(8, 72, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(LdtHipError):
        ops.gemm_bf16(x, w)
    with pytest.raises(LdtHipError):
        ops.gemm_bf16(x.cpu(), w.cpu())


# ------------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, H):
    """oracle/ldt_oracle.attention minus the projections: -> [B,H,N,Dh] contiguous."""
    B, N, C = q.shape
    M = k.shape[1]
    dh = C // H
    sp = lambda z, n: z.reshape(B, n, H, dh).permute(0, 2, 1, 3).double()
    w = (sp(q, N) @ sp(k, M).transpose(-1, -2) * dh ** -0.5).softmax(-1)
    return (w @ sp(v, M)).contiguous()


@pytest.mark.parametrize("B,H,Nq,Nk,dh", [(2, 16, 256, 256, 64), (3, 4, 32, 32, 64), (2, 2, 8, 8, 64), (2, 4, 300, 77, 32),
                                         (1, 4, 2048, 256, 32), (2, 4, 40, 2048, 32), (1, 2, 130, 512, 64), (2, 2, 8, 5, 32),
                                         # the whole-head 8-wave kernel (Dh = 64, 129..256 queries, <= 256 keys): idle waves, ragged key tiles
                                         (3, 16, 129, 129, 64), (2, 4, 200, 200, 64), (2, 2, 160, 256, 64), (1, 2, 256, 130, 64), (2, 2, 255, 65, 64)])
def test_attention(B, H, Nq, Nk, dh):
    g = torch.Generator().manual_seed(Nq + Nk)
    C = H * dh
    q = bf(torch.randn(B, Nq, C, generator=g)); kv = bf(torch.randn(B, Nk, 2 * C, generator=g) * 1.5)
    ref = ref_attention(q, kv[..., :C], kv[..., C:], H)
    qd = dev(q, torch.bfloat16).view(B * Nq, C); kvd = dev(kv, torch.bfloat16).view(B * Nk, 2 * C)
    out = ops.attention_fwd(qd, kvd[:, :C], kvd[:, C:], B, H, Nq, Nk, dh)
    assert out.shape == (B, H, Nq, dh)
    assert rel_mse(out.float().cpu(), ref) < 2e-5
    assert float((out.float().cpu() - ref).abs().max()) < 0.05


def test_attention_softmax_spike():
    """A key row that dominates one query row late in the sequence forces the online-softmax rescale."""
    B, H, N, dh = 1, 1, 256, 64
    q = torch.zeros(B, N, dh); k = torch.zeros(B, N, dh); v = torch.randn(B, N, dh, generator=torch.Generator().manual_seed(3))
    q[0, 5, 0] = 8.0; k[0, 200, 0] = 16.0          # score 128/8 = 16 at key 200 only for query 5
    q, k, v = bf(q), bf(k), bf(v)
    ref = ref_attention(q, k, v, H)
    out = ops.attention_fwd(dev(q, torch.bfloat16).view(N, dh), dev(k, torch.bfloat16).view(N, dh),
                            dev(v, torch.bfloat16).view(N, dh), B, H, N, N, dh)
    assert float((out.float().cpu() - ref).abs().max()) < 0.02


@pytest.mark.parametrize("dh,step,N", [(64, 2.0, 320), (64, 5.0, 320), (32, 3.0, 320), (64, 2.0, 256), (64, 5.0, 256), (64, 7.0, 192)])
def test_attention_softmax_staircase(dh, step, N):
    """Row maxima that keep growing along the key axis: every 32-key block raises the maximum of every query row by
    `step` (in log2 units), below and above the deferred-rescale threshold of the kernel (2^6), so both the "keep the
    stale reference" and the "rescale" paths run many times in one row; full-tensor fp64 reference.  N = 320 runs the streaming
    kernel (per 32-key block), N <= 256 at Dh = 64 the whole-head kernel (one online-softmax step per 64-key tile)."""
    B, H = 2, 2
    C = H * dh
    g = torch.Generator().manual_seed(int(step * 10) + dh)
    q = torch.randn(B, N, C, generator=g) * 0.1; k = torch.randn(B, N, C, generator=g) * 0.1; v = torch.randn(B, N, C, generator=g)
    # channel 0 of every head carries the staircase: q = a, k_j = (j // 32) * step * ln2 * sqrt(dh) / a  ->  score_j = (j//32) * step (log2 units)
    a = 4.0
    stair = (torch.arange(N) // 32).float() * step * 0.6931471805599453 * dh ** 0.5 / a
    for h in range(H):
        q[:, :, h * dh] = a
        k[:, :, h * dh] = stair
    q, k, v = bf(q), bf(k), bf(v)
    ref = ref_attention(q, k, v, H)
    out = ops.attention_fwd(dev(q, torch.bfloat16).view(B * N, C), dev(k, torch.bfloat16).view(B * N, C),
                            dev(v, torch.bfloat16).view(B * N, C), B, H, N, N, dh)
    assert rel_mse(out.float().cpu(), ref) < 2e-5
    assert float((out.float().cpu() - ref).abs().max()) < 0.05


# ------------------------------------------------------------------------------------------- LN / modulate
@pytest.mark.parametrize("M,C", [(64, 1024), (10, 128), (33, 64), (8, 256), (5, 96)])
def test_layernorm_modulate(M, C):
    g = torch.Generator().manual_seed(C)
    rps = 1 if M % 2 else 2
    S = M // rps
    x = torch.randn(M, C, generator=g) * 3 + 1
    mod = torch.randn(S, 2 * C, generator=g) * 0.5
    w = torch.rand(C, generator=g) + 0.5; b = torch.randn(C, generator=g)
    md = dev(mod)
    out = ops.layernorm_modulate(dev(x), shift=md[:, :C], scale=md[:, C:], mod_sample_stride=2 * C, rows_per_sample=rps)
    ref = O.modulate(O.layer_norm(x.double()), mod[:, :C].double().repeat_interleave(rps, 0), mod[:, C:].double().repeat_interleave(rps, 0))
    assert rel_mse(out.float().cpu(), ref) < 1e-5
    out = ops.layernorm_modulate(dev(x), w=dev(w), b=dev(b))
    assert rel_mse(out.float().cpu(), O.layer_norm(x.double(), w.double(), b.double())) < 1e-5
    out = ops.layernorm_modulate(dev(x))
    assert rel_mse(out.float().cpu(), O.layer_norm(x.double())) < 1e-5


# ------------------------------------------------------------------------------------------- fp32 linears
@pytest.mark.parametrize("M,N,K", [(5, 64, 3), (300, 128, 20), (1000, 768, 64), (7, 2048, 1024), (65, 40, 128), (130, 300, 256),
                                   (32, 1000, 512), (1000, 2048, 1024), (4096, 40, 128), (2048, 128, 20),
                                   # LDS-DMA streaming form of the skinny AdaLN-row GEMM (M <= 32, N >= 2048, K % 32 == 0): ragged N, M < 32
                                   (19, 2085, 512), (32, 4224, 256), (1, 2048, 1024),
                                   # streaming forms (skinny_linear.hip): millions-of-rows convs of the Compressor, ragged tails
                                   (20001, 128, 3), (16385, 128, 20), (9000, 64, 32), (30003, 3, 128), (8193, 8, 64), (10000, 1, 512)])
def test_sgemm(M, N, K):
    from ldt_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    F = torch.nn.functional
    out = ops.sgemm(dev(a), dev(w), dev(b))
    assert rel_mse(out.cpu(), a.double() @ w.double().T + b.double()) < 1e-12
    out = ops.sgemm(dev(a), dev(w), dev(b), act_in=ACT_SILU, act_out=ACT_RELU)
    assert rel_mse(out.cpu(), torch.relu(F.silu(a.double()) @ w.double().T + b.double())) < 1e-12
    out = ops.sgemm(dev(a), dev(w), None, act_out=ACT_GELU, out_bf16=True)
    assert rel_mse(out.float().cpu(), F.gelu(a.double() @ w.double().T)) < 1e-5
    # strided input / output views
    big = torch.randn(M, K + 11, generator=g)
    outb = torch.zeros(M, N + 5, device="cuda")
    ops.sgemm(dev(big)[:, 3:3 + K], dev(w), dev(b), out=outb[:, 2:2 + N])
    assert rel_mse(outb[:, 2:2 + N].cpu(), big[:, 3:3 + K].double() @ w.double().T + b.double()) < 1e-12
    assert float(outb[:, :2].abs().sum()) == 0


def test_time_embedding_golden():
    """sinusoid + TimeEmbedding.mlp on device vs the golden captured from the reference (a8, Q5)."""
    from conftest import load_golden
    from ldt_amd._lib import ACT_SILU
    a, sds = load_golden("time_embedding")
    half = 128
    freq = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1)))
    e = ops.sinusoid(dev(a["t"]), dev(freq))
    assert float((e.cpu() - a["sinusoid"]).abs().max()) < 2e-6
    w = sds["w"]
    h = ops.sgemm(e, dev(w["mlp.0.weight"]), dev(w["mlp.0.bias"]), act_out=ACT_SILU)
    out = ops.sgemm(h, dev(w["mlp.2.weight"]), dev(w["mlp.2.bias"]))
    assert rel_mse(out.cpu(), a["out"]) < 1e-10


# ------------------------------------------------------------------------------------------- sampler update
def test_sampler_step_ancestral_bit_exact(tiny_cfg):
    from ldt_amd.diffusion import DiffusionVPSDE
    N = 1000
    tiny_cfg.sde.sample_N = N
    sde = DiffusionVPSDE(tiny_cfg.sde)
    osde = O.VPSDE(tiny_cfg.sde)
    tiny_cfg.sde.sample_N = 50
    ts, coef, mode = sde.step_table(N, "ancestral", 1e-6)
    assert mode == 0
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 120, generator=g) * 5; p = torch.randn(2, 8, 120, generator=g); z = torch.randn(2, 8, 120, generator=g)
    for i in (0, 1, 500, 998, 999):
        t = torch.ones(2) * ts[i]
        idx = (t * (N - 1)).long(); beta = osde.betas[idx]
        score = -p / torch.sqrt(osde.var(t))[:, None, None]
        xm = (x + beta[:, None, None] * score) / torch.sqrt(1. - beta)[:, None, None]
        xn = xm + torch.sqrt(beta)[:, None, None] * z
        xmd = torch.empty_like(x, device="cuda")
        out = ops.sampler_step(dev(x), dev(p), dev(coef), i, 0, noise=dev(z), x_mean_out=xmd)
        assert torch.equal(xmd.cpu(), xm), i
        assert torch.equal(out.cpu(), xn), i


def test_sampler_step_device_counter_and_noise_stride():
    coef = torch.tensor([[1.5, -0.5, 2.0, 0.0], [0.5, 0.25, 1.0, 0.0]])
    x = torch.randn(64); p = torch.randn(64); nz = torch.randn(2, 64)
    ctr = torch.tensor([1], dtype=torch.int32, device="cuda")
    out = ops.sampler_step(dev(x), dev(p), dev(coef), 0, 1, noise=dev(nz), noise_step_stride=64, step_ptr=ctr)
    assert torch.allclose(out.cpu(), 0.5 * x + 0.25 * p + nz[1], atol=1e-6)


def test_philox_normal_statistics_and_shard_invariance():
    n = 1 << 20
    a = ops.philox_normal((n,), "cuda", seed=1234, step=7)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 3).mean())) < 2e-2 and abs(float((a ** 4).mean()) - 3) < 5e-2
    lo = ops.philox_normal((n // 2,), "cuda", seed=1234, step=7)
    hi = ops.philox_normal((n // 2,), "cuda", seed=1234, step=7, elem_offset=n // 2)
    assert torch.equal(torch.cat([lo, hi]), a)                      # sharding by element offset is invisible
    b = ops.philox_normal((n,), "cuda", seed=1234, step=8)
    assert abs(float((a * b).mean())) < 5e-3                        # steps are independent streams
    # the fused step draws the same stream
    x = torch.zeros(n, device="cuda"); coef = torch.tensor([[1.0, 0.0, 1.0, 0.0]] * 8, device="cuda")
    out = ops.sampler_step(x, x, coef, 7, 1, seed=1234)
    assert torch.equal(out, a)


@pytest.mark.parametrize("C,M,mode", [(128, 4096, "affine"), (128, 300, "mod"), (64, 16, "mod"), (64, 1000, "affine"), (128, 128, "plain")])
def test_fused_ln_mlp_resid(C, M, mode):
    """x += gate * W_dn.GELU(W_up.LN(x)) in one kernel vs a plain fp32 PyTorch reference of the same op (bf16 operands:
    relative MSE of the UPDATE <= 1e-4) — tail rows, both channel widths, affine / modulated / plain LayerNorm."""
    import torch.nn.functional as F
    from ldt_amd import ops
    g = torch.Generator().manual_seed(C + M)
    rps = 100 if mode == "mod" else 0
    nS = (M + 99) // 100
    x = torch.randn(M, C, generator=g) * 2 + 0.3
    w_up = torch.randn(4 * C, C, generator=g) / C ** 0.5
    w_dn = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5
    b_up, b_dn = torch.randn(4 * C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    ln_w = ln_b = mod = None
    h = F.layer_norm(x, (C,), None, None, 1e-6)
    gate = 1.0
    if mode == "affine":
        ln_w, ln_b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
        h = h * ln_w + ln_b
    elif mode == "mod":
        mod = torch.randn(nS, 3 * C, generator=g) * 0.5                       # shift | scale | gate per sample
        idx = torch.arange(M) // rps
        h = h * (1 + mod[idx, C:2 * C]) + mod[idx, :C]
        gate = mod[idx, 2 * C:]
    upd = gate * (F.gelu(h @ w_up.t() + b_up) @ w_dn.t() + b_dn)
    xd = x.cuda()
    kw = {}
    if mode == "affine":
        kw = dict(ln_w=ln_w.cuda(), ln_b=ln_b.cuda())
    elif mode == "mod":
        md = mod.cuda()
        kw = dict(shift=md[:, :C], scale=md[:, C:2 * C], gate=md[:, 2 * C:], mod_sample_stride=3 * C, rows_per_sample=rps)
    mirror = torch.full((M, C + 64), -7.0, device="cuda", dtype=torch.bfloat16)[:, :C] if M % 2 == 0 else None   # strided rows
    ops.ln_mlp_resid_(xd, w_up.cuda().to(torch.bfloat16).contiguous(), b_up.cuda(), w_dn.cuda().to(torch.bfloat16).contiguous(),
                      b_dn.cuda(), x_bf16_out=mirror, **kw)
    got_upd = xd.cpu() - x
    assert rel_mse(got_upd, upd) < 1e-4
    if mirror is not None:                      # the bf16 copy written in the same pass == a cast of the fp32 result, bit for bit
        assert torch.equal(mirror, xd.to(torch.bfloat16))
    # the follow-on LayerNorm + linear of the next block, computed by the same launch on the rows it has just written,
    # == ldt_ln_linear run on that result afterwards, bit for bit (plain and modulated next-LN)
    wn = (torch.randn(2 * C, C, generator=g) / C ** 0.5).cuda().to(torch.bfloat16).contiguous()
    bn = torch.randn(2 * C, generator=g).cuda()
    nlw, nlb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    for nkw in (dict(ln_w=nlw, ln_b=nlb),) + ((dict(shift=md[:, :C], scale=md[:, C:2 * C], mod_sample_stride=3 * C, rows_per_sample=rps),) if mode == "mod" else ()):
        x2 = x.cuda()
        _, qn = ops.ln_mlp_resid_(x2, w_up.cuda().to(torch.bfloat16).contiguous(), b_up.cuda(), w_dn.cuda().to(torch.bfloat16).contiguous(),
                                  b_dn.cuda(), next_linear=dict(w=wn, bias=bn, **nkw), **kw)
        assert torch.equal(x2, xd)
        assert torch.equal(qn, ops.ln_linear(xd, wn, bn, **nkw))
    with pytest.raises(Exception):
        ops.ln_mlp_resid_(torch.zeros(8, 96, device="cuda"), torch.zeros(384, 96, device="cuda", dtype=torch.bfloat16),
                          torch.zeros(384, device="cuda"), torch.zeros(96, 384, device="cuda", dtype=torch.bfloat16), torch.zeros(96, device="cuda"))


@pytest.mark.parametrize("B,H,Nq,Nk,gated", [(3, 4, 2048, 256, False), (2, 4, 256, 2048, True), (2, 2, 64, 8, False), (3, 2, 8, 64, True),
                                              (2, 4, 200, 100, True),
                                              # resident form (one workgroup per (cloud, head): B*H >= 256, >= 2 query blocks)
                                              (64, 4, 256, 256, True), (128, 2, 130, 8, False),
                                              (256, 4, 512, 256, False), (512, 2, 520, 40, True), (256, 4, 1000, 500, True)])
def test_fused_attention_oproj_resid(B, H, Nq, Nk, gated):
    """x += gate * (Wo . Attn(q,k,v)' + bo) with the raw head-merge reinterpret (quirk Q1) in one kernel vs a plain fp32
    PyTorch reference of the same op (relative MSE of the update <= 1e-4), and vs the two-kernel path it replaces."""
    from ldt_amd import ops
    from ldt_amd._lib import EPI_RESID_F32
    dh = 32
    C = H * dh
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q = (torch.randn(B * Nq, C, generator=g)).to(torch.bfloat16)
    kv = (torch.randn(B * Nk, 2 * C, generator=g)).to(torch.bfloat16)
    wo = (torch.randn(C, C, generator=g) / C ** 0.5).to(torch.bfloat16)
    bo = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B * Nq, C, generator=g)
    gate = torch.randn(B, 3 * C, generator=g) if gated else None
    # fp32 reference (model/layers.py:190-199): heads split on channels, (B,H,N,Dh) result reshaped RAW to (B,N,C)
    qf = q.float().view(B, Nq, H, dh).permute(0, 2, 1, 3)
    kf = kv.float()[:, :C].reshape(B, Nk, H, dh).permute(0, 2, 1, 3)
    vf = kv.float()[:, C:].reshape(B, Nk, H, dh).permute(0, 2, 1, 3)
    att = torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf            # (B,H,Nq,dh)
    upd = att.contiguous().reshape(B * Nq, C) @ wo.float().t() + bo
    if gated:
        upd = upd * gate[:, C:2 * C].repeat_interleave(Nq, 0)
    xd = x.cuda()
    gd = None if gate is None else gate.cuda()
    kvd = kv.cuda()
    ops.attention_oproj_resid_(q.cuda(), kvd[:, :C], kvd[:, C:], B, H, Nq, Nk, dh, wo.cuda(), bo.cuda(), xd,
                               gate=None if gd is None else gd[:, C:2 * C], gate_sample_stride=3 * C if gated else 0)
    assert rel_mse(xd.cpu() - x, upd) < 1e-4
    # the two-kernel path it replaces (bf16 O round trip): agreement well inside the bf16 tolerance
    x2 = x.cuda()
    o = ops.attention_fwd(q.cuda(), kvd[:, :C], kvd[:, C:], B, H, Nq, Nk, dh)
    ops.gemm_bf16(o.view(B * Nq, C), wo.cuda(), bo.cuda(), EPI_RESID_F32, out=x2, resid=x2, gate=None if gd is None else gd[:, C:2 * C],
                  gate_sample_stride=3 * C if gated else 0, rows_per_sample=Nq)
    assert rel_mse(xd.cpu() - x, x2.cpu() - x) < 1e-5


@pytest.mark.parametrize("C,M,N,mode", [(128, 4096, 128, "affine"), (128, 333, 384, "mod"), (64, 16, 192, "mod"), (64, 1000, 64, "plain")])
def test_fused_ln_linear(C, M, N, mode):
    """bf16 out = LN(x)[affine | modulated] @ W^T + b in one kernel vs a plain fp32 PyTorch reference (rel MSE <= 1e-4;
    the bf16 output rounding alone is ~3e-6) and vs the LayerNorm + GEMM pair it replaces."""
    import torch.nn.functional as F
    from ldt_amd import ops
    g = torch.Generator().manual_seed(C + M + N)
    rps = 50
    x = torch.randn(M, C, generator=g) * 1.5 - 0.2
    w = torch.randn(N, C, generator=g) / C ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    h = F.layer_norm(x, (C,), None, None, 1e-6)
    kw = {}
    if mode == "affine":
        lw, lb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
        h = h * lw + lb
        kw = dict(ln_w=lw.cuda(), ln_b=lb.cuda())
    elif mode == "mod":
        mod = torch.randn((M + rps - 1) // rps, 2 * C, generator=g) * 0.5
        idx = torch.arange(M) // rps
        h = h * (1 + mod[idx, C:]) + mod[idx, :C]
        md = mod.cuda()
        kw = dict(shift=md[:, :C], scale=md[:, C:], mod_sample_stride=2 * C, rows_per_sample=rps)
    want = h @ w.t() + b
    wb = w.cuda().to(torch.bfloat16).contiguous()
    got = ops.ln_linear(x.cuda(), wb, b.cuda(), **kw)
    assert got.dtype == torch.bfloat16 and got.shape == (M, N)
    assert rel_mse(got.float().cpu(), want) < 1e-4
    two = ops.gemm_bf16(ops.layernorm_modulate(x.cuda(), **{({"ln_w": "w", "ln_b": "b"}.get(k, k)): v for k, v in kw.items()}), wb, b.cuda())
    assert rel_mse(got.float().cpu(), two.float().cpu()) < 2e-5


def test_resid_ring_epilogue_is_bit_identical_to_plain_loads():
    """The residual GEMM of a one-tile-per-workgroup launch reads its fp32 residual rows through the idle operand ring
    (LDS-DMA, counted waits); any non-zero debug knob switches that off without changing the arithmetic (bit 32 skips
    nothing): x, x(1+scale) and the row statistics must agree bit for bit, producer (LN folding) and plain form."""
    from ldt_amd import _lib
    from ldt_amd._lib import EPI_RESID_F32
    g = torch.Generator().manual_seed(12)
    for M, K in ((4096, 1024), (16384, 4096)):
        N = 1024                                              # M/256 * 4 <= 256 tiles: every workgroup has one tile
        a = (torch.randn(M, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
        b = torch.randn(N, generator=g).cuda(); gate = torch.randn(N, generator=g).cuda(); sc = (0.2 * torch.randn(N, generator=g)).cuda()
        x0 = (torch.randn(M, N, generator=g) + 0.3).cuda()
        outs = []
        for bits in (0, 32):
            _lib.lib().ldt_dbg_gemm_epi(bits)
            x = x0.clone()
            xs, st = ops.gemm_resid_lnstats(a, w, b, x, sc, gate=gate, rows_per_sample=256)
            y = x0.clone()
            ops.gemm_bf16(a, w, b, EPI_RESID_F32, out=y, resid=y, gate=gate, rows_per_sample=256)
            torch.cuda.synchronize()
            outs.append((x, xs, st, y))
        _lib.lib().ldt_dbg_gemm_epi(-1)
        for p, q in zip(*outs):
            assert torch.equal(p, q)
        assert rel_mse(outs[0][0].cpu(), outs[0][3].cpu()) < 1e-12   # producer's x vs the plain residual GEMM's (128^2 kernel at small M)
        ref = x0.double().cpu() + gate.double().cpu() * (a.double().cpu() @ w.double().cpu().T + b.double().cpu())
        assert rel_mse(outs[0][0].cpu(), ref) < 1e-9


def test_gemm256_variants_bit_equal(tmp_path):
    """The one-tile-per-workgroup residual GEMMs of the 256x256 kernel have two epilogues for the same arithmetic: XRING (residual rows by
    LDS-DMA through the idle operand ring, hand-counted `s_waitcnt vmcnt(N)`: LDT_RESID_RING, the default) or the register epilogue.  Both
    must agree BIT FOR BIT (a toolchain that emitted one VMEM op more or fewer per pass, or a mis-counted wait, would read data
    before it lands).  Same seeded problems in two child processes: plain RESID_F32 and the LN-fold producer (M = 16,384 x N = 1,024: exactly
    256 tiles), the LN-fold consumer with GELU on a multi-tile persistent shape (N = 4,096: four tiles per workgroup) and a plain bf16
    projection (N = 3,072), each launched three times (run-to-run differences would betray a race).  (Round 3 also held the round-2 operand
    stream, removed in round 4, to the same bits; round 6 holds the W-from-registers form of the one-tile kernels to them.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    child = r'''
import sys, torch
sys.path.insert(0, %r)
from ldt_amd import ops
from ldt_amd._lib import EPI_RESID_F32, EPI_GELU_BF16, EPI_BF16
g = torch.Generator().manual_seed(11)
M, D, K = 16384, 1024, 1024
a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(D, K, generator=g) / 32).bfloat16().cuda()
a4 = torch.randn(M, 4 * K, generator=g).bfloat16().cuda(); w4 = (torch.randn(D, 4 * K, generator=g) / 64).bfloat16().cuda()
b = torch.randn(D, generator=g).cuda(); gate = torch.randn(1, D, generator=g).cuda(); sc = (0.3 * torch.randn(D, generator=g)).cuda()
wu = (torch.randn(4 * D, D, generator=g) / 32).bfloat16().cuda(); S = torch.randn(4 * D, generator=g).cuda(); C = torch.randn(4 * D, generator=g).cuda()
wq = (torch.randn(3 * D, D, generator=g) / 32).bfloat16().cuda(); bq = torch.randn(3 * D, generator=g).cuda()
x0 = torch.randn(M, D, generator=g).cuda()
outs = {}
for rep in range(3):
    x1 = x0.clone()
    ops.gemm_bf16(a, w, b, EPI_RESID_F32, out=x1, resid=x1, gate=gate, gate_sample_stride=0, rows_per_sample=M)
    x2 = x0.clone()
    xs, st = ops.gemm_resid_lnstats(a, w, b, x2, sc, gate=gate, gate_sample_stride=0, rows_per_sample=M)
    x3 = x0.clone()
    xs3, st3 = ops.gemm_resid_lnstats(a4, w4, b, x3, sc, gate=gate, gate_sample_stride=0, rows_per_sample=M)      # K = 4096 (mlp.out)
    u = ops.gemm_lnfold(xs, wu, st, S, C, EPI_GELU_BF16)
    q = ops.gemm_bf16(xs, wq, bq, EPI_BF16)
    cur = dict(x1=x1.cpu(), x2=x2.cpu(), xs=xs.cpu(), st=st.cpu(), x3=x3.cpu(), xs3=xs3.cpu(), st3=st3.cpu(), u=u.cpu(), q=q.cpu())
    if outs:
        assert all(torch.equal(outs[k], cur[k]) for k in cur), "run-to-run difference"
    outs = cur
torch.save(outs, sys.argv[1])
''' % ROOT
    res = {}
    # third child (round 6): the one-tile residual GEMMs with the weight operand loaded straight into registers from a fragment-order copy
    # (gemm_bf16.hip WREG: asm loads, hand-counted waits, no W in LDS; LDT_GEMM_WREG=1 packs the copy on the fly) — same MFMA order, same bits
    # fourth / fifth child (round 6): the grouped tile order of the multi-tile kernels (groups of 4 row panels by default; 8 until round 5, 1 =
    # row-major) only permutes which workgroup computes which tile — every tile's arithmetic is the same
    for ring, wreg, gm in (("0", "0", ""), ("1", "0", ""), ("1", "1", ""), ("1", "0", "8"), ("1", "0", "1")):
        out = tmp_path / ("ring%s_wreg%s_gm%s.pt" % (ring, wreg, gm))
        env = dict(os.environ, LDT_RESID_RING=ring, LDT_GEMM_WREG=wreg)
        if gm:
            env["LDT_GEMM_GM"] = gm
        r = subprocess.run([sys.executable, "-c", child, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[(ring, wreg, gm)] = torch.load(out)
    base = res[("0", "0", "")]
    for key, cur in res.items():
        for k in base:
            assert torch.equal(cur[k], base[k]), "LDT_RESID_RING=%s LDT_GEMM_WREG=%s LDT_GEMM_GM=%s differs from the register-epilogue path in %s" % (key + (k,))
    assert bool(torch.isfinite(base["x1"]).all()) and float(base["x1"].abs().mean()) > 0.1 and float(base["u"].float().abs().mean()) > 0.01
