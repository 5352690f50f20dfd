"""tools/dbg: the four Score GEMM shapes at M = 16384 — role-specialised persistent 128 x 128 kernel (LDT_GEMM_PIPE=1, csrc/gemm_pipe.hip) against
the 256^2 persistent kernel, alternating child processes; plain epilogues and the LN-folded producer."""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
    M = int(os.environ.get("PB_M", "16384"))
    pipe = os.environ.get("LDT_GEMM_PIPE", "0") == "1"
    shapes = [("qkv", 3072, 1024, EPI_BF16), ("o", 1024, 1024, EPI_RESID_F32), ("up", 4096, 1024, EPI_GELU_BF16), ("dn", 1024, 4096, EPI_RESID_F32)]
    torch.manual_seed(0)
    line = "M=%d pipe=%d:" % (M, pipe)
    for name, N, K, epi in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda"); gate = torch.randn(1, N, device="cuda")
        f32 = epi == EPI_RESID_F32
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        kw = dict(out=out)
        if f32: kw.update(resid=out, gate=gate, rows_per_sample=M)
        fn = lambda: ops.gemm_bf16(x, w, b, epi, **kw)
        if f32:
            sc = torch.randn(N, device="cuda") * 0.1
            fnp = lambda: ops.gemm_resid_lnstats(x, w, b, out, sc, gate=gate, rows_per_sample=M, granule=128 if pipe else 256)
        for f, tag in ((fn, ""), (fnp if f32 else None, "+lnfold")):
            if f is None: continue
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            line += "  %s%s %.1f us (%.0f TF)" % (name, tag, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
else:
    for rep in range(2):
        for pipe in ("1", "0"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LDT_GEMM_PIPE=pipe), stderr=subprocess.DEVNULL)
