import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
torch.manual_seed(0)
for C in (64, 128):
    for M in (16, 128, 1000, 4096, 65536):
        x0 = torch.randn(M, C, device="cuda")
        w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
        b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = torch.rand(C, device="cuda") + 0.5; lb = torch.randn(C, device="cuda")
        mod = torch.randn((M + 63) // 64, 3 * C, device="cuda")
        for gated in (False, True):
            outs = []
            for rep in range(4):
                x = x0.clone()
                if gated: ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, shift=mod[:, :C], scale=mod[:, C:2*C], gate=mod[:, 2*C:], mod_sample_stride=3*C, rows_per_sample=64)
                else: ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, ln_w=lw, ln_b=lb)
                outs.append(x)
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            print("C=%d M=%d gated=%s deterministic=%s maxdiff=%.3g" % (C, M, gated, same, max((outs[0] - o).abs().max().item() for o in outs[1:])), flush=True)
