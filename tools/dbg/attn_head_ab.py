"""tools/dbg: the Score's self-attention at T tokens — streaming (LDT_ATTN_FORCE=1), resident (2), whole-head 8-wave (3): time + bit-equality."""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ldt_amd import ops
    for (B, H, T) in ((64, 16, 256), (64, 16, 200), (3, 16, 129), (64, 16, 160)):
        dh, C = 64, 16 * 64
        torch.manual_seed(T)
        qkv = (torch.randn(B * T, 3 * C, device="cuda") * 1.5).to(torch.bfloat16)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        out = torch.zeros(B, H, T, dh, device="cuda", dtype=torch.bfloat16)
        for _ in range(3): ops.attention_fwd(q, k, v, B, H, T, T, dh, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.attention_fwd(q, k, v, B, H, T, T, dh, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        h = hashlib.sha1(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        nb = min(B, 4)                                                  # fp32 torch reference on a few samples (tools/dbg only)
        qq, kk, vv = (t_[:nb * T].float().view(nb, T, H, dh).transpose(1, 2) for t_ in (q, k, v))
        ref = torch.softmax(qq @ kk.transpose(-1, -2) * dh ** -0.5, -1) @ vv
        err = ((out[:nb].float() - ref) ** 2).sum() / (ref ** 2).sum()
        h += " rel-mse %.2e" % err.item()
        print("force=%s B=%d T=%d: %.1f us  %.0f GB/s  sha %s" % (os.environ.get("LDT_ATTN_FORCE"), B, T, us, 4 * B * T * C * 2 / us / 1e3, h), flush=True)
else:
    for f in ("1", "3", "1", "3"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LDT_ATTN_FORCE=f), stderr=subprocess.DEVNULL)
