"""tools/dbg: residual GEMM + LayerNorm at small M — unsplit (RESID_F32 epilogue + ln) vs split-K partials + ln-with-reduce."""
import os, sys, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    from ldt_amd._lib import EPI_RESID_F32
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    torch.manual_seed(0)
    D = 1024
    for M in (1024, 2048, 4096):
        for K in (1024, 4096):
            x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(D, K, device="cuda") / K ** 0.5).bfloat16()
            b = torch.randn(D, device="cuda"); gate = torch.randn(1, D, device="cuda"); sh = torch.randn(1, D, device="cuda"); sc = torch.randn(1, D, device="cuda")
            X = torch.randn(M, D, device="cuda")
            base_g = t(lambda: ops.gemm_bf16(x, w, b, EPI_RESID_F32, out=X, resid=X, gate=gate, rows_per_sample=M))
            base_l = t(lambda: ops.layernorm_modulate(X, shift=sh, scale=sc, rows_per_sample=M))
            line = "M=%d K=%d shape=%s: unsplit gemm %.1f + ln %.1f = %.1f us |" % (M, K, os.environ.get("LDT_GEMM_SPLITK_SHAPE", "auto"), base_g, base_l, base_g + base_l)
            for S in (2, 4, 8):
                if K % (S * 64) or K // S < 128: continue
                parts = ops.gemm_bf16_splitk(x, w, S)
                tg = t(lambda: ops.gemm_bf16_splitk(x, w, S))
                tl = t(lambda: ops.layernorm_modulate_resid_(X, parts, bias=b, gate=gate, shift=sh, scale=sc, rows_per_sample=M))
                line += "  S=%d: %.1f + %.1f = %.1f" % (S, tg, tl, tg + tl)
            print(line, flush=True)
else:
    for shape, row in (("0", "1"), ("1", "1"), ("2", "1"), ("0", "0")):
        print("--- split-K tile shape %s (0: 128x128, 1: 128x64, 2: 64x64), LN reduce form %s (1: workgroup per row, 0: wave per row)" % (shape, row), flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LDT_GEMM_SPLITK_SHAPE=shape, LDT_LN_RESID_ROW=row))
