import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
B, H, T, dh = 64, 16, 256, 64
C = H * dh
qkv = torch.randn(B * T, 3 * C, device="cuda").to(torch.bfloat16)
out = torch.empty(B, H, T, dh, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.attention_fwd(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], B, H, T, T, dh, out=out)
torch.cuda.synchronize()
