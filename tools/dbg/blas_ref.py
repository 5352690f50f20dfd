"""Reference point: PyTorch-ROCm's library GEMM (hipBLASLt / rocBLAS) on the Score shapes, same random operands as gemm_bench.py."""
import torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ldt_amd import ops
M = int(os.environ.get("M", 16384))
torch.manual_seed(0)
for name, N, K in (("qkv", 3072, 1024), ("o", 1024, 1024), ("up", 4096, 1024), ("dn", 1024, 4096)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    def lib_mm(): return torch.matmul(x, w.t())
    def mine(): return ops.gemm_bf16(x, w, b, 1)
    def mine_discard(): return ops.gemm_bf16(x, w, b, 5, out=dis)
    dis = torch.empty(M, N, device="cuda")
    res = []
    for fn in (lib_mm, mine):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append("%s %.1f us %.0f TF" % (fn.__name__, us, 2.0 * M * N * K / us / 1e6))
    print(name, " | ".join(res), flush=True)
