"""tools/dbg: can the per-step conditional AdaLN-row GEMM (skinny fp32, 604 MB of weights) hide under the Score blocks of a small-batch step?
Runs the sampling loop (B = 32, T = 32, N steps, graph replay) alone, N skinny GEMMs alone, and both at once on two streams / two host threads."""
import os, sys, time, threading
sys.path.insert(0, '.')
import torch, ldt_amd
from ldt_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N)
torch.manual_seed(0)
score = ldt_amd.Score(cfg.score); comp = ldt_amd.Compressor(cfg.compressor); comp.init()
tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
M, K, Nn = 32, 1024, 24 * 6144 + 2048
a = torch.randn(M, K, device="cuda"); w = torch.randn(Nn, K, device="cuda") * 0.03; b = torch.randn(Nn, device="cuda"); out = torch.empty(M, Nn, device="cuda")
side = torch.cuda.Stream()
def main_loop():
    torch.cuda.set_device(0)
    tr.sample(32)
def side_loop(n):
    torch.cuda.set_device(0)
    with torch.cuda.stream(side):
        for _ in range(n): ops.sgemm(a, w, b, out=out)
def timed(fs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=f) for f in fs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
main_loop(); side_loop(5); torch.cuda.synchronize()
for rep in range(2):
    tm = timed([main_loop]); ts = timed([lambda: side_loop(N)]); tb = timed([main_loop, lambda: side_loop(N)])
    print("N=%d: loop alone %.1f ms (%.3f ms/step), %d skinny GEMMs alone %.1f ms (%.1f us each), both %.1f ms -> hidden %.0f %% of the GEMM time"
          % (N, tm * 1e3, tm / N * 1e3, N, ts * 1e3, ts / N * 1e6, tb * 1e3, 100 * (tm + ts - tb) / ts), flush=True)
