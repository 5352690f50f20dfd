"""tools/dbg: ln_mlp_resid_kernel with 2 (4 waves x 32 rows) vs 1 (8 waves x 16 rows) row tiles per wave: time + bit-equality (LDT_MLP_RT)."""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    for C, M, gated in ((128, 128 * 2048, False), (128, 1024 * 2048, False), (128, 128 * 2048, True), (64, 100 * 2048 + 77, False)):
        torch.manual_seed(C + M % 1000)
        x0 = torch.randn(M, C, device="cuda")
        w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
        b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = torch.rand(C, device="cuda") + 0.5; lb = torch.randn(C, device="cuda")
        S = M // 2048 + 1
        mod = torch.randn(S, 3 * C, device="cuda") * 0.3
        kw = dict(shift=mod[:, :C], scale=mod[:, C:2 * C], gate=mod[:, 2 * C:], mod_sample_stride=3 * C, rows_per_sample=2048) if gated else dict(ln_w=lw, ln_b=lb)
        xb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        x = x0.clone(); ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, x_bf16_out=xb, **kw); torch.cuda.synchronize()
        h = hashlib.sha1(x.cpu().numpy().tobytes() + xb.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        x = x0.clone()
        for _ in range(2): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, **kw)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print("RT=%s C=%d M=%d gated=%d: %.1f us  %.0f TFLOP/s  x-traffic %.2f TB/s  sha %s" % (os.environ.get("LDT_MLP_RT"), C, M, gated, us, 2.0 * M * C * 4 * C * 2 / us / 1e6, M * C * 8 / us / 1e6, h), flush=True)
else:
    for rt in ("2", "1", "2", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LDT_MLP_RT=rt), stderr=subprocess.DEVNULL)
