"""tools/dbg: the four GEMMs of a Score block at M = 2048 / 1024 / 4096 — mid-size tile kernel (csrc/gemm_mid.hip) against the 2-phase v1
kernels (LDT_GEMM_MID=0), alternating child processes.  `python tools/dbg/mid_bench.py`"""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
    shapes = [("qkv", 3072, 1024, EPI_BF16), ("o", 1024, 1024, EPI_RESID_F32), ("up", 4096, 1024, EPI_GELU_BF16), ("dn", 1024, 4096, EPI_RESID_F32)]
    torch.manual_seed(0)
    cold = torch.empty(256 * 1024 * 1024, device="cuda", dtype=torch.float32)      # 1 GiB: evicts the Infinity Cache between timed launches

    def timeit(fn, n=30, flush=False):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        if not flush:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        tot = 0.0
        for _ in range(8):                                                      # weights cold in HBM, as inside the SDE loop
            cold.fill_(1.0)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / 8 * 1e3

    for M in (2048, 1024, 4096):
        line = "M=%d mid=%s:" % (M, os.environ.get("LDT_GEMM_MID", "1"))
        for name, N, K, epi in shapes:
            x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            b = torch.randn(N, device="cuda"); gate = torch.randn(1, N, device="cuda")
            f32 = epi == EPI_RESID_F32
            r = torch.randn(M, N, device="cuda") if f32 else None
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
            kw = dict(out=out)
            if f32: kw.update(resid=r, gate=gate, rows_per_sample=M)
            us = timeit(lambda: ops.gemm_bf16(x, w, b, epi, **kw))
            usc = timeit(lambda: ops.gemm_bf16(x, w, b, epi, **kw), flush=True)
            line += "  %s %.1f us (%.0f TF; cold %.1f)" % (name, us, 2.0 * M * N * K / us / 1e6, usc)
        print(line, flush=True)
else:
    for rep in range(2):
        for mid in ("1", "0"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LDT_GEMM_MID=mid), stderr=subprocess.DEVNULL)
