"""Are kernels slower inside a forward than stand-alone because their operands come cold (HBM instead of the 256 MB
Infinity Cache), or because of the clock?  Times each kernel (a) back to back on the same buffers (cache-warm) and
(b) after a 1 GiB write to an unrelated buffer (cache-cold), HIP events around the single launch."""
import sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
M, D = 16384, 1024
torch.manual_seed(0)
flush = torch.empty(1 << 28, device="cuda")                       # 1 GiB of fp32
x32 = torch.randn(M, D, device="cuda"); gate = torch.randn(D, device="cuda"); sc = torch.randn(D, device="cuda")
cases = []
for name, N, K, epi in (("qkv", 3072, 1024, EPI_BF16), ("up+gelu", 4096, 1024, EPI_GELU_BF16)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cases.append((name, lambda a=a, w=w, b=b, epi=epi, out=out: ops.gemm_bf16(a, w, b, epi, out=out)))
for name, K in (("fc_o (fold producer)", 1024), ("mlp.out (fold producer)", 4096)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(D, K, device="cuda") / K ** 0.5).to(torch.bfloat16); b = torch.randn(D, device="cuda")
    cases.append((name, lambda a=a, w=w, b=b: ops.gemm_resid_lnstats(a, w, b, x32, sc, gate=gate, rows_per_sample=256)))
qkv = torch.randn(M, 3 * D, device="cuda").to(torch.bfloat16); o = torch.empty(64, 16, 256, 64, device="cuda", dtype=torch.bfloat16)
cases.append(("attention", lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], 64, 16, 256, 256, 64, out=o)))
xl = torch.randn(M, D, device="cuda")
cases.append(("layernorm", lambda: ops.layernorm_modulate(xl, shift=gate, scale=sc, rows_per_sample=M)))
for name, fn in cases:
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for it in range(8):
            if mode == "cold":
                flush.add_(1.0)
            else:
                fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    print("%-26s warm %.1f us   cold %.1f us" % (name, res["warm"], res["cold"]), flush=True)
