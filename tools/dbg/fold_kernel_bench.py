"""Plain vs LN-folded GEMM kernels at the Score shapes (M = 16384), interleaved rounds in one process."""
import os, sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
M, D = 16384, 1024
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xs = torch.randn(M, D, device="cuda").to(torch.bfloat16)
stats = torch.rand(D // 256, M, 2, device="cuda") + 1.0
stats[..., 1] += 300.0
x32 = torch.randn(M, D, device="cuda"); gate = torch.randn(D, device="cuda"); sc = torch.randn(D, device="cuda")
cases = []
for name, N, K, epi in (("qkv", 3072, 1024, EPI_BF16), ("up", 4096, 1024, EPI_GELU_BF16)):
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16); b = torch.randn(N, device="cuda")
    S = torch.randn(N, device="cuda"); C = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cases.append((name + " plain", (lambda w=w, b=b, epi=epi, out=out: ops.gemm_bf16(xs, w, b, epi, out=out)), 2.0 * M * N * K))
    cases.append((name + " fold ", (lambda w=w, S=S, C=C, epi=epi: ops.gemm_lnfold(xs, w, stats, S, C, epi)), 2.0 * M * N * K))
for name, K in (("o", 1024), ("dn", 4096)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(D, K, device="cuda") / K ** 0.5).to(torch.bfloat16); b = torch.randn(D, device="cuda")
    cases.append((name + " plain", (lambda a=a, w=w, b=b: ops.gemm_bf16(a, w, b, EPI_RESID_F32, out=x32, resid=x32, gate=gate, rows_per_sample=256)), 2.0 * M * D * K))
    cases.append((name + " fold ", (lambda a=a, w=w, b=b: ops.gemm_resid_lnstats(a, w, b, x32, sc, gate=gate, rows_per_sample=256)), 2.0 * M * D * K))
res = {}
for rnd in range(3):
    for name, fn, fl in cases:
        res.setdefault(name, []).append(timeit(fn))
tag = os.environ.get("LDT_HIP_LIB", "product")[-16:]
for name, fn, fl in cases:
    print("%s %s: %.1f us (%.0f TF)" % (tag, name, min(res[name]), fl / min(res[name]) / 1e6), flush=True)
