"""Throughput of the validation-metric kernels at the shipped evaluation size (2048-point clouds).
    python tools/bench_metrics.py [N_clouds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ldt_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 2048, 3, generator=g).cuda()
y = torch.randn(N, 2048, 3, generator=g).cuda()
for name, fn in (("chamfer_pairwise", lambda: ops.chamfer_pairwise(x, y)), ("emd_approx pairwise", lambda: ops.emd_approx(x, y, pairwise=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t0
    pairs = N * N
    print("%-22s %d x %d pairs of 2048 pts: %.3f s  = %.0f pairs/s  (%.2f G point-pair evals/s)"
          % (name, N, N, dt, pairs / dt, pairs * 2048 * 2048 * (2 if "chamfer" in name else 27) / dt / 1e9))
