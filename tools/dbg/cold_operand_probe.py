#!/usr/bin/env python3
"""Which operand's coldness costs the in-loop GEMMs their 6-8 % against warm back-to-back launches?  The mlp.out-shaped LN-fold producer
(16384 x 1024 x 4096) and the MLP-up consumer (16384 x 4096 x 1024) launched back to back with (a) everything reused (warm: W and the activations
sit in the Infinity Cache), (b) a different weight buffer per launch (40 x 8 MB: W always from HBM), (c) different activation / residual buffers per
launch (cycled through > 256 MB), (d) both.   usage: cold_operand_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_GELU_BF16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(3)
M, D = 16384, 1024
NW, NA = 40, 4
def bf(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).bfloat16().cuda()
w_dn = [bf(D, 4 * D, sc=1 / 64) for _ in range(NW)]; w_up = [bf(4 * D, D, sc=1 / 32) for _ in range(NW)]
u = [bf(M, 4 * D) for _ in range(NA)]                       # 128 MB each
xs = [bf(M, D) for _ in range(NA * 4)]                      # 32 MB each
x = [torch.randn(M, D, generator=g).cuda() for _ in range(NA * 2)]   # 64 MB each (fp32 residual stream)
b = torch.randn(D).cuda(); gate = torch.randn(1, D).cuda(); sc = (0.3 * torch.randn(D)).cuda()
S = torch.randn(4 * D).cuda(); C = torch.randn(4 * D).cuda()
_, st = ops.gemm_resid_lnstats(u[0], w_dn[0], b, x[0].clone(), sc, gate=gate, gate_sample_stride=0, rows_per_sample=M)
def timeit(fn):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
arms = (("warm", 0, 0), ("cold W", 1, 0), ("cold activations", 0, 1), ("cold W + activations", 1, 1))
best = {}
for rnd in range(4):                                      # (alternating rounds: the clock state after a minute of MFMA load differs from a cold start)
    for name, cw, ca in (arms if rnd % 2 == 0 else arms[::-1]):
        dn = timeit(lambda i: ops.gemm_resid_lnstats(u[i % NA if ca else 0], w_dn[i % NW if cw else 0], b, x[i % len(x) if ca else 0], sc, gate=gate, gate_sample_stride=0, rows_per_sample=M))
        up = timeit(lambda i: ops.gemm_lnfold(xs[i % len(xs) if ca else 0], w_up[i % NW if cw else 0], st, S, C, EPI_GELU_BF16))
        best[name] = (min(best.get(name, (1e9, 1e9))[0], dn), min(best.get(name, (1e9, 1e9))[1], up))
        print("round %d %-22s mlp.out producer %6.1f us   MLP-up + GELU consumer %6.1f us" % (rnd, name, dn, up), flush=True)
for name, _, _ in arms:
    print("best    %-22s mlp.out producer %6.1f us   MLP-up + GELU consumer %6.1f us" % ((name,) + best[name]))
