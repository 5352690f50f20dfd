"""tools/dbg: what a read-only stream gets on this box (torch reductions over a 604 MB fp32 tensor = the size of configs[4]'s per-step AdaLN weights)."""
import torch
x = torch.randn(149504, 1024, device="cuda")
y = torch.empty(8, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, f in (("sum", lambda: x.sum()), ("amax", lambda: x.amax()), ("sum(dim=1)", lambda: x.sum(dim=1)), ("bf16 copy-out", lambda: x.to(torch.bfloat16))):
    us = t(f)
    print("%-14s %.1f us  %.2f TB/s read" % (name, us, x.numel() * 4 / us / 1e6))
