"""Time the one-kernel grouper against the five-kernel chain at the C4 shape (B=1024, 2048 points, 256 groups of 16) and at the
shipped shape (32 groups of 128).  LDT_FUSED_GROUPER is flipped in-process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import compressor as Cm, ops

torch.manual_seed(0)
grp = Cm.LocalGrouper(128).cuda()
G = grp.pack()
B = int(os.environ.get("B", 1024))
for S, k in ((256, 16), (32, 128), (128, 32)):
    pts = torch.randn(B, 2048, 3, device="cuda"); feat = torch.randn(B, 2048, 128, device="cuda")
    fi = ops.fps(pts, S); ki = ops.knn(pts, ops.gather_rows(pts, fi), k)
    def fused(): return ops.grouper_mlp(feat, pts, fi, ki, G["alpha"], G["beta"], G["wimg"], G["b_pre1"], G["b_pre2"], G["b_pre3"])
    def chain():
        from ldt_amd._lib import EPI_RELU_BF16
        U = ops.group_normalize(feat, pts, fi, ki, G["alpha"], G["beta"])
        h1 = ops.gemm_bf16(U, G["w_pre1"], G["b_pre1"], EPI_RELU_BF16)
        r = ops.gemm_bf16(h1, G["w_pre2"], G["b_pre2"], EPI_RELU_BF16)
        return ops.maxpool(ops.gemm_bf16(r, G["w_pre3"], G["b_pre3"], EPI_RELU_BF16, skip=h1), B * S, k)
    for name, fn in (("fused", fused), ("chain", chain)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("S=%d k=%d %s: %.3f ms" % (S, k, name, dt * 1e3), flush=True)
