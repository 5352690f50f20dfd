"""(any non-zero knob also switches the ring-landed residual read off: the numbers attribute the plain epilogue)
Where does the residual GEMM's epilogue time go?  LN-fold producer (fc_o: K=1024, mlp.out: K=4096) at M=16384 with parts of
the epilogue switched off (ldt_dbg_gemm_epi bits), on rotating cache-cold buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldt_amd import ops, _lib
M, N = 16384, 1024
torch.manual_seed(0)
NB = 6
outs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
gate = torch.randn(N, device="cuda"); sc = torch.randn(N, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
variants = [("full", 0), ("no-resid-read", 1), ("no-x-store", 2), ("no-xs-store", 4), ("no-stats", 8), ("no-read,no-xs,no-stats", 13),
            ("stores only off (2|4)", 6), ("nothing (15)", 15)]
for K in (1024, 4096):
    xs_in = [(torch.randn(M, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(NB)]
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    res = {}
    for rnd in range(3):
        for name, bits in variants:
            _lib.lib().ldt_dbg_gemm_epi(bits)
            for i in range(NB): ops.gemm_resid_lnstats(xs_in[i], w, b, outs[i], sc, gate=gate, rows_per_sample=256)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(4):
                for i in range(NB): ops.gemm_resid_lnstats(xs_in[i], w, b, outs[i], sc, gate=gate, rows_per_sample=256)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
            for o in outs: o.normal_()
    for name, _ in variants:
        print("K=%d %-28s %s us" % (K, name, " ".join("%.1f" % t for t in res[name])), flush=True)
_lib.lib().ldt_dbg_gemm_epi(-1)
