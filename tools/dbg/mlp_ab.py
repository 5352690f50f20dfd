#!/usr/bin/env python3
"""A/B of the Compressor's fused MLP kernel (`ln_mlp_resid_kernel`) between builds of libldt_hip.so, alternating child processes on one box:
the kernel alone at ROWS rows (default 2 M = a decoder level of 1024 clouds; plain and AdaLN-gated), a CRC of its output (same arithmetic per
row => the builds must agree bit for bit), and Compressor encode / decode of 1024 clouds.
usage: mlp_ab.py libA.so libB.so [...] [rounds]      ("product" = the in-tree library; "lib@VAR=VAL" sets an environment switch for that leg)"""
import os, subprocess, sys
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2
child = r'''
import os, sys, time, zlib, torch
sys.path.insert(0, os.getcwd())
import ldt_amd
from ldt_amd import ops
C, M = 128, int(os.environ.get("ROWS", 1024 * 2048))
torch.manual_seed(0)
x0 = torch.randn(M, C, device="cuda")
w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = 1 + 0.1 * torch.randn(C, device="cuda"); lb = 0.1 * torch.randn(C, device="cuda")
B = M // 2048 if M >= 2048 else 1
mod = 0.3 * torch.randn(B, 3 * C, device="cuda")
def plain(x): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, ln_w=lw, ln_b=lb)
def gated(x): ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, shift=mod[:, :C], scale=mod[:, C:2 * C], gate=mod[:, 2 * C:], mod_sample_stride=3 * C, rows_per_sample=M // B)
out = []
for name, fn in (("plain", plain), ("gated", gated)):
    x = x0.clone(); fn(x); torch.cuda.synchronize()
    crc = zlib.crc32(x[:: max(1, M // 65536)].contiguous().cpu().numpy().tobytes())
    x = x0.clone()
    for _ in range(2): fn(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn(x)
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.1f us crc %08x" % (name, e0.elapsed_time(e1) / 10 * 1e3, crc))
if os.environ.get("AB_C4", "1") != "0":
    cfg = ldt_amd.airplane_config(latent_tokens=256)
    torch.manual_seed(0)
    comp = ldt_amd.Compressor(cfg.compressor).cuda(); comp.init()
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(1024, 2048, 3, generator=g); pts = pts - pts.mean(1, keepdim=True); pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()
    f = lambda: comp(pts)["all_eps"]
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    out.append("encode %.0f clouds/s" % (1024 / dt))
    eps = torch.randn(1024, 256, 120, device="cuda")
    comp.sample((1024, 2048), given_eps=eps); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = comp.sample((1024, 2048), given_eps=eps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = o[0] if isinstance(o, (tuple, list)) else o
    out.append("decode %.0f clouds/s crc %08x" % (1024 / dt, zlib.crc32(o[::64].contiguous().cpu().numpy().tobytes())))
print(" | ".join(out))
'''
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        lib, _, kv = l.partition("@")                     # "lib.so@VAR=VAL": the library with an environment switch set
        if kv:
            env[kv.split("=")[0]] = kv.split("=")[1]
        if lib != "product":
            env["LDT_HIP_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(out.stdout[-500:], out.stderr[-2000:]); sys.exit(1)
        print("round %d %-40s %s" % (r, l, out.stdout.strip().splitlines()[-1]), flush=True)
