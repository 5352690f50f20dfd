"""tools/dbg: the four GEMMs of a Score block at small M under each v1 tile shape (LDT_GEMM_V1_SHAPE; 3 / 4 = the 8-wave forms)."""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
    shapes = [("qkv", 3072, 1024, EPI_BF16), ("o", 1024, 1024, EPI_RESID_F32), ("up", 4096, 1024, EPI_GELU_BF16), ("dn", 1024, 4096, EPI_RESID_F32)]
    torch.manual_seed(0)
    for M in (2048, 1024):
        line = "M=%d shape=%s map=%s:" % (M, os.environ.get("LDT_GEMM_V1_SHAPE", "auto"), os.environ.get("LDT_GEMM_V1_MAP", "0"))
        for name, N, K, epi in shapes:
            x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            b = torch.randn(N, device="cuda"); gate = torch.randn(1, N, device="cuda")
            f32 = epi == EPI_RESID_F32
            r = torch.randn(M, N, device="cuda") if f32 else None
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
            kw = dict(out=out)
            if f32: kw.update(resid=r, gate=gate, rows_per_sample=M)
            for _ in range(5): ops.gemm_bf16(x, w, b, epi, **kw)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 50
            for _ in range(n): ops.gemm_bf16(x, w, b, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            line += "  %s %.1f us (%.0f TF)" % (name, us, 2.0 * M * N * K / us / 1e6)
        print(line, flush=True)
else:
    for mp in ("0", "1"):
        for shape in ("-1", "0", "1", "2", "3", "4"):
            env = dict(os.environ, LDT_GEMM_FORCE="128", LDT_GEMM_V1_MAP=mp)
            if shape != "-1": env["LDT_GEMM_V1_SHAPE"] = shape
            subprocess.run([sys.executable, __file__, "child"], env=env, stderr=subprocess.DEVNULL)
