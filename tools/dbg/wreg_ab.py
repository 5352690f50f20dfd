#!/usr/bin/env python3
"""W-from-registers form of the 256-tile GEMM (gemm_bf16.hip WREG) against the LDS form, one process, one box: (1) bit equality of every
epilogue form on the Score shapes, three launches each; (2) alternating timings (HIP events, `reps` launches per arm and round).
usage: wreg_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ldt_amd import ops
from ldt_amd._lib import lib, EPI_RESID_F32, EPI_GELU_BF16, EPI_BF16
L = lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(11)
M, D = 16384, 1024
a = torch.randn(M, D, generator=g).bfloat16().cuda(); w = (torch.randn(D, D, generator=g) / 32).bfloat16().cuda()
a4 = torch.randn(M, 4 * D, generator=g).bfloat16().cuda(); w4 = (torch.randn(D, 4 * D, generator=g) / 64).bfloat16().cuda()
b = torch.randn(D, generator=g).cuda(); gate = torch.randn(1, D, generator=g).cuda(); sc = (0.3 * torch.randn(D, generator=g)).cuda()
wu = (torch.randn(4 * D, D, generator=g) / 32).bfloat16().cuda(); S = torch.randn(4 * D, generator=g).cuda(); C = torch.randn(4 * D, generator=g).cuda()
bu = torch.randn(4 * D, generator=g).cuda()
wq = (torch.randn(3 * D, D, generator=g) / 32).bfloat16().cuda(); bq = torch.randn(3 * D, generator=g).cuda()
x0 = torch.randn(M, D, generator=g).cuda()
xs0, st0 = ops.gemm_resid_lnstats(a, w, b, x0.clone(), sc, gate=gate, gate_sample_stride=0, rows_per_sample=M)
xbuf = x0.clone()

def forms():
    yield "o.resid", lambda: ops.gemm_bf16(a, w, b, EPI_RESID_F32, out=xbuf, resid=xbuf, gate=gate, gate_sample_stride=0, rows_per_sample=M), 2.0 * M * D * D
    yield "o.producer", lambda: ops.gemm_resid_lnstats(a, w, b, xbuf, sc, gate=gate, gate_sample_stride=0, rows_per_sample=M), 2.0 * M * D * D
    yield "dn.producer", lambda: ops.gemm_resid_lnstats(a4, w4, b, xbuf, sc, gate=gate, gate_sample_stride=0, rows_per_sample=M), 8.0 * M * D * D
    yield "up.gelu.consumer", lambda: ops.gemm_lnfold(xs0, wu, st0, S, C, EPI_GELU_BF16), 8.0 * M * D * D
    yield "up.gelu.plain", lambda: ops.gemm_bf16(xs0, wu, bu, EPI_GELU_BF16), 8.0 * M * D * D
    yield "up.bf16.plain", lambda: ops.gemm_bf16(xs0, wu, bu, EPI_BF16), 8.0 * M * D * D
    yield "qkv.bf16.plain", lambda: ops.gemm_bf16(xs0, wq, bq, EPI_BF16), 6.0 * M * D * D
    yield "qkv.bf16.consumer", lambda: ops.gemm_lnfold(xs0, wq, st0, S[:3 * D].contiguous(), C[:3 * D].contiguous(), EPI_BF16), 6.0 * M * D * D

def outputs(fn):
    xbuf.copy_(x0)
    r = fn()
    r = r if isinstance(r, tuple) else (r,)
    return [t.clone() for t in r] + [xbuf.clone()]

bad = 0
for name, fn, _ in forms():
    L.ldt_dbg_gemm_wreg(0); ref = outputs(fn)
    for rep in range(3):
        L.ldt_dbg_gemm_wreg(1); cur = outputs(fn)
        ok = all(torch.equal(x, y) for x, y in zip(ref, cur))
        if not ok:
            bad += 1
            d = [float((x.float() - y.float()).abs().max()) for x, y in zip(ref, cur)]
            print("MISMATCH %s rep %d: max abs diff per output %s" % (name, rep, d), flush=True)
            break
    else:
        print("bit-equal %s" % name, flush=True)
print("bit equality: %s" % ("ALL EQUAL" if not bad else "%d forms differ" % bad), flush=True)

reps = 30
best = {}
for r in range(rounds):
    for name, fn, flop in forms():
        for arm in (0, 1):
            L.ldt_dbg_gemm_wreg(arm)
            for _ in range(3): fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            best[(name, arm)] = min(best.get((name, arm), 1e9), us)
for name, fn, flop in forms():
    u0, u1 = best[(name, 0)], best[(name, 1)]
    print("%-20s LDS %7.1f us (%4.0f TF)   WREG %7.1f us (%4.0f TF)   %+.1f %%" % (name, u0, flop / u0 / 1e6, u1, flop / u1 / 1e6, 100 * (u1 / u0 - 1)), flush=True)
L.ldt_dbg_gemm_wreg(-1)
