"""rocprofv3 target (prof_issue_split.sh TAG wreg): the mlp.out-shaped LN-fold producer GEMM (16384 x 1024 x 4096) and fc_o (K = 1024), each in the
LDS form (symbol <4, 1, 1, 0>) and the W-from-registers form (<4, 1, 1, 1>) of the 256-tile kernel, REPS launches per arm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops
from ldt_amd._lib import lib
L = lib()
g = torch.Generator().manual_seed(11)
M, D = 16384, 1024
reps = int(os.environ.get("REPS", 10))
b = torch.randn(D, generator=g).cuda(); gate = torch.randn(1, D, generator=g).cuda(); sc = (0.3 * torch.randn(D, generator=g)).cuda()
x = torch.randn(M, D, generator=g).cuda()
for K in (4 * D, D):
    a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(D, K, generator=g) / K ** 0.5).bfloat16().cuda()
    for arm in (0, 1):
        L.ldt_dbg_gemm_wreg(arm)
        for _ in range(reps):
            ops.gemm_resid_lnstats(a, w, b, x, sc, gate=gate, gate_sample_stride=0, rows_per_sample=M)
        torch.cuda.synchronize()
L.ldt_dbg_gemm_wreg(-1)
