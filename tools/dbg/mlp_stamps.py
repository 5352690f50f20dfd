"""tools/dbg: where a workgroup of the Compressor's fused MLP kernel spends its life — wall-clock stamps of wave 0 of every workgroup
(fused_mlp.hip built with -DMLP_STAMPS: `bash tools/dbg/build_variant.sh mlpst "-DMLP_STAMPS"`; run with
LDT_HIP_LIB=tools/dbg/lib/libldt_mlpst.so).  Prints the median duration of each segment, and per CU how its workgroups tile the launch."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops, _lib
L = _lib.lib()
L.ldt_dbg_mlp_stamps.argtypes = [ctypes.c_void_p]
C, M = 128, int(os.environ.get("ROWS", 1024 * 2048))
torch.manual_seed(0)
x = torch.randn(M, C, device="cuda")
w_up = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16); w_dn = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
b_up = torch.randn(4 * C, device="cuda"); b_dn = torch.randn(C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
fn = lambda: ops.ln_mlp_resid_(x, w_up, b_up, w_dn, b_dn, ln_w=lw, ln_b=lb)
nwg = (M + 127) // 128
buf = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
for _ in range(3): fn()
torch.cuda.synchronize()
assert L.ldt_dbg_mlp_stamps(buf.data_ptr()) == 0
fn(); torch.cuda.synchronize()
L.ldt_dbg_mlp_stamps(None)
st = buf.view(nwg, 64).cpu()
hw, xcc = st[:, 62], st[:, 63] & 0xF
t = st.double()
t0 = t[:, 0].min()
us = lambda a: (a - t0) / 100.0
seg = lambda a, b: float(((t[:, b] - t[:, a]) / 100.0).median())
print("rows %d, %d workgroups, launch span %.1f us" % (M, nwg, float(us(t[:, 42]).max())))
print("life of a workgroup (median us): total %.2f | entry -> x landed + LN -> fragments %.2f | -> chunk 0 wait %.2f, barrier %.2f" %
      (seg(0, 42), seg(0, 1), seg(1, 3), seg(3, 4)))
print("  prologue: kernel entry -> weight requests out %.2f | row + LN-vector requests out %.2f (addresses: %.2f) | all landed %.2f | LN math + h image written %.2f | fragments read %.2f" %
      (seg(0, 49), seg(49, 51), seg(49, 50), seg(51, 52), seg(52, 53), seg(53, 1)))
for ch in range(8):
    nxt = 2 + 4 * (ch + 1) if ch < 7 else 40
    print("  chunk %d: stage + GEMM1 + GELU %.2f | GEMM2 %.2f | next wait %.2f barrier %.2f" %
          (ch, seg(4 + 4 * ch, 5 + 4 * ch), seg(5 + 4 * ch, nxt), seg(2 + 4 * ch, 3 + 4 * ch), seg(3 + 4 * ch, 4 + 4 * ch)))
print("  stores issued %.2f | stores retired %.2f" % (seg(40, 41), seg(41, 42)))
# per CU: (xcc, se, cu) from HW_ID: cu_id bits 11:8, sh 12, se 15:13 (gfx9 layout)
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)
simd = (hw >> 4) & 3
ids = cu.unique()
print("%d distinct CUs seen; workgroups per CU %.1f" % (len(ids), nwg / len(ids)))
busy, gaps, conc = [], [], []
for c in ids[:64].tolist():
    m = cu == c
    s, e = us(t[m, 0]), us(t[m, 42])
    o = s.argsort(); s, e = s[o], e[o]
    span = float(e.max() - s.min())
    busy.append(float((e - s).sum()) / span)                # average number of resident workgroups
    # time between a workgroup's retirement and the next entry on this CU (any slot): entries sorted, match each entry k >= 2 to the (k-2)-th earliest end
    es = e.sort().values
    if len(s) > 2:
        gaps.append(float((s[2:] - es[:-2]).median()))
print("average resident workgroups per CU %.2f (2 = both slots always full); median slot turn-around (retire -> next entry) %.2f us" %
      (sum(busy) / len(busy), sorted(gaps)[len(gaps) // 2]))
