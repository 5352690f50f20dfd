"""tools/dbg: do two streams' small-M GEMMs overlap?  n launches on one stream vs n/2 + n/2 on two streams (host: one thread)."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16, EPI_RESID_F32
torch.manual_seed(0)
for M in (1024, 2048):
    for name, N, K, epi in (("qkv", 3072, 1024, EPI_BF16), ("dn", 1024, 4096, EPI_RESID_F32)):
        bufs = []
        for i in range(2):
            x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
            kw = dict(out=out)
            if epi == EPI_RESID_F32: kw.update(resid=out, rows_per_sample=M)
            bufs.append((x, w, b, kw))
        s = [torch.cuda.Stream(), torch.cuda.Stream()]
        def run(two, n=200):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(n):
                j = i & 1
                with torch.cuda.stream(s[j if two else 0]):
                    x, w, b, kw = bufs[j]
                    ops.gemm_bf16(x, w, b, epi, **kw)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        run(False, 20); run(True, 20)
        print("M=%d %s: one stream %.1f us/launch, two streams %.1f us/launch" % (M, name, run(False), run(True)), flush=True)
# graph form: capture 20 launches per stream into one graph each, replay both
g = []
M, N, K = 1024, 1024, 4096
for i in range(2):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        ops.gemm_bf16(x, w, b, EPI_RESID_F32, out=out, resid=out, rows_per_sample=M)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(20): ops.gemm_bf16(x, w, b, EPI_RESID_F32, out=out, resid=out, rows_per_sample=M)
    g.append((gr, st, x, w, b, out))
def rg(two, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for j in range(2):
            gr, st = g[j if two else 0][:2]
            with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * 40) * 1e6
rg(False, 3); rg(True, 3)
print("graphs of 20 dn launches (M=1024): same graph twice %.1f us/launch, two graphs on two streams %.1f us/launch" % (rg(False), rg(True)))
