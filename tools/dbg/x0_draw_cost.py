#!/usr/bin/env python3
"""Scale-out hygiene: what the redundant host-side x0 draw costs.  Every rank of a sharded `Trainer.sample(B)` draws the
FULL global x0 = torch.randn(B, T, z) on its CPU generator (the reference's draw, diffusion_continuous.py:237) and keeps its 1/W slice.  This
times that draw for the BASELINE configs[2] shape (B = 512, T = 256, z = 120: 15.7 M normals) alone and with W = 8 processes drawing at once
on this box's host cores (the container's thread quota applies to all of them together).   usage: x0_draw_cost.py [W]"""
import os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    torch.manual_seed(1234)
    torch.randn(8, 256, 120)                                  # warm the generator / allocator
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); x = torch.randn(512, 256, 120); ts.append(time.perf_counter() - t0)
    print("%.4f %.4f %d" % (min(ts), sorted(ts)[2], torch.get_num_threads()))
    sys.exit(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def run(n):
    ps = [subprocess.Popen([sys.executable, __file__, "child"], stdout=subprocess.PIPE, text=True) for _ in range(n)]
    return [tuple(float(v) for v in p.communicate()[0].split()) for p in ps]
one = run(1)[0]
many = run(W)
print("host cores visible: %s; torch threads per process: %d" % (os.cpu_count(), int(one[2])))
print("one process:   torch.randn(512, 256, 120) best %.1f ms, median %.1f ms" % (one[0] * 1e3, one[1] * 1e3))
print("%d at once:     best %.1f .. %.1f ms, median %.1f .. %.1f ms" % (W, min(m[0] for m in many) * 1e3, max(m[0] for m in many) * 1e3,
                                                                      min(m[1] for m in many) * 1e3, max(m[1] for m in many) * 1e3))
