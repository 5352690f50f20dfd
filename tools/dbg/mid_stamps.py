"""tools/dbg: where a mid-size GEMM workgroup's time goes — wall-clock stamps of its loader wave / compute wave (gemm_mid.hip built with
-DMID_STAMPS: `bash tools/dbg/build_variant.sh midst "-DMID_STAMPS"`; run with LDT_HIP_LIB=tools/dbg/lib/libldt_midst.so).
Prints medians over workgroups (us since the earliest stamp of the launch)."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops, _lib
from ldt_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32
L = _lib.lib()
L.ldt_dbg_mid_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
cold = torch.empty(256 * 1024 * 1024, device="cuda", dtype=torch.float32)
for (name, M, N, K, epi, flush) in [("up", 2048, 4096, 1024, EPI_GELU_BF16, False), ("up", 2048, 4096, 1024, EPI_GELU_BF16, True),
                                    ("qkv", 2048, 3072, 1024, EPI_BF16, True), ("dn", 2048, 1024, 4096, EPI_RESID_F32, True), ("dn", 2048, 1024, 4096, EPI_RESID_F32, False), ("o", 2048, 1024, 1024, EPI_RESID_F32, True),
                                    ("up1k", 1024, 4096, 1024, EPI_GELU_BF16, True)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    f32 = epi == EPI_RESID_F32
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(out=out)
    if f32: kw.update(resid=torch.randn(M, N, device="cuda"), gate=torch.randn(1, N, device="cuda"), rows_per_sample=M)
    nwg = 4096
    buf = torch.zeros(nwg * 2 * 64, dtype=torch.int64, device="cuda")
    for _ in range(3): ops.gemm_bf16(x, w, b, epi, **kw)
    torch.cuda.synchronize()
    if flush: cold.fill_(1.0)
    torch.cuda.synchronize()
    assert L.ldt_dbg_mid_stamps(buf.data_ptr()) == 0
    buf.zero_(); torch.cuda.synchronize()
    ops.gemm_bf16(x, w, b, epi, **kw)
    torch.cuda.synchronize()
    L.ldt_dbg_mid_stamps(None)
    st = buf.view(nwg, 2, 64).cpu().double()
    used = st[:, 0, 0] > 0
    st = st[used]
    t0 = st[st > 0].min()
    us = lambda t: (t - t0) / 100.0                         # 100 MHz ticks -> us
    med = lambda t: float(us(t).median()); mx = lambda t: float(us(t).max())
    nkt = K // 64 if not f32 else K // 64
    print("== %s M=%d N=%d K=%d %s: %d workgroups" % (name, M, N, K, "cold" if flush else "warm", st.shape[0]))
    c, l = st[:, 0], st[:, 1]
    print("  compute: entry %.2f | prologue barrier passed %.2f | main loop done %.2f | epilogue stores retired %.2f (max %.2f)"
          % (med(c[:, 0]), med(c[:, 1]), med(c[:, 2]), med(c[:, 3]), mx(c[:, 3])))
    print("  loader : entry %.2f | 3 K-tiles issued %.2f | K-tile 0 landed %.2f | barrier %.2f" % (med(l[:, 0]), med(l[:, 1]), med(l[:, 2]), med(l[:, 3])))
    rows = []
    for kt in range(min(nkt - 1, 19)):
        a_, b_, c_ = l[:, 4 + 3 * kt], l[:, 5 + 3 * kt], l[:, 6 + 3 * kt]
        prev = l[:, 3] if kt == 0 else l[:, 6 + 3 * (kt - 1)]
        rows.append("kt%02d issue %.2f wait-data %.2f wait-barrier %.2f (released at %.2f)" %
                    (kt, float(((a_ - prev) / 100).median()), float(((b_ - a_) / 100).median()), float(((c_ - b_) / 100).median()), med(c_)))
    print("  " + "\n  ".join(rows))
