import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
for (B, H, T, dh) in ((64, 16, 256, 64), (64, 16, 32, 64), (64, 4, 2048, 32)):
    C = H * dh
    Nk = T if dh == 64 else 256
    qkv = torch.randn(B * T, 3 * C, device="cuda").to(torch.bfloat16) if dh == 64 else None
    if dh == 64:
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        q = torch.randn(B * T, C, device="cuda").to(torch.bfloat16); kv = torch.randn(B * Nk, 2 * C, device="cuda").to(torch.bfloat16)
        k, v = kv[:, :C], kv[:, C:]
    out = torch.empty(B, H, T, dh, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.attention_fwd(q, k, v, B, H, T, Nk, dh, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.attention_fwd(q, k, v, B, H, T, Nk, dh, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byt = (B * T * C * 2 + B * Nk * C * 2 * 2) * 2 // 2 + B * T * C * 2
    print("attn B=%d H=%d Nq=%d Nk=%d dh=%d: %.1f us  %.0f GB/s  %.0f TFLOP/s" % (B, H, T, Nk, dh, ms * 1e3, (B*T*C*2*2 + B*Nk*C*2*2) / ms / 1e6, 4.0 * B * H * T * Nk * dh / ms / 1e9))
