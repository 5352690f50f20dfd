#!/usr/bin/env python3
"""The two cross-attention microbench shapes of BASELINE configs[3] (128 clouds x 4 heads x Dh 32: Q = 2048 x K/V = T and the reverse) + the
2048 x 32-key shape, for one or more builds in alternating child processes.   usage: attn_micro.py lib.so|product ... [rounds]"""
import os, subprocess, sys
child = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ldt_amd import ops
B, H, dh, d = 128, 4, 32, 128
def t(nq, nk, n=30):
    q = torch.randn(B * nq, d, device="cuda").to(torch.bfloat16); kv = torch.randn(B * nk, 2 * d, device="cuda").to(torch.bfloat16)
    o = torch.empty(B, H, nq, dh, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention_fwd(q, kv[:, :d], kv[:, d:], B, H, nq, nk, dh, out=o)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("%.1f %.1f %.1f %.1f" % (t(2048, 256), t(256, 2048), t(2048, 32), t(256, 256)))
'''
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2
best = {}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "product": env["LDT_HIP_LIB"] = l
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        try:
            v = [float(x) for x in out.stdout.strip().splitlines()[-1].split()]
        except (ValueError, IndexError):
            print(out.stdout[-300:], out.stderr[-1200:]); raise
        best[l] = [min(a, b) for a, b in zip(best.get(l, v), v)]
        print("round %d %-36s 2048x256 %6.1f us  256x2048 %6.1f us  2048x32 %6.1f us  256x256 %6.1f us" % ((r, os.path.basename(l)) + tuple(v)), flush=True)
for l in libs:
    print("best    %-36s 2048x256 %6.1f us  256x2048 %6.1f us  2048x32 %6.1f us  256x256 %6.1f us" % ((os.path.basename(l),) + tuple(best[l])))
