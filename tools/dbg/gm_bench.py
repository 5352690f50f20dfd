"""A/B of the grouped tile order (rows per group) on the Score GEMM shapes, interleaved in one process."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops, _lib
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
M = int(os.environ.get("M", 16384))
shapes = [("qkv", 3072, 1024, EPI_BF16), ("up", 4096, 1024, EPI_GELU_BF16), ("dn", 1024, 4096, EPI_RESID_F32), ("o", 1024, 1024, EPI_RESID_F32)]
h = _lib.lib()
torch.manual_seed(0)
for name, N, K, epi in shapes:
    x = (torch.randn(M, K, device="cuda")).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); gate = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(out=out)
    if epi == EPI_RESID_F32: kw.update(resid=out, gate=gate, rows_per_sample=256)
    res = {}
    for rnd in range(3):
        for gm in (1, 2, 4, 8, 16):
            h.ldt_dbg_gemm_group_m(gm)
            for _ in range(2): ops.gemm_bf16(x, w, b, epi, **kw)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n): ops.gemm_bf16(x, w, b, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(gm, []).append(e0.elapsed_time(e1) / n * 1e3)
    print(name, " ".join("gm%d: %.1f us (%.0f TF)" % (gm, min(v), 2.0 * M * N * K / min(v) / 1e6) for gm, v in res.items()), flush=True)
