"""Round 5: the Compressor's cross-attention shapes (BASELINE configs[3] microbench: 128 clouds x 4 heads, Dh = 32) through the streaming kernel
(LDT_ATTN_FORCE=1) and the resident kernel with its query blocks split over 1 / 2 / 4 / 8 workgroups per head (LDT_ATTN_FORCE=2 LDT_ATTN_QSPLIT=n),
one child process per setting (the switches are read once), three rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from ldt_amd import ops
B, H, dh = 128, 4, 32
d = H * dh
for (Nq, Nk) in ((2048, 256), (2048, 32), (256, 2048)):
    if Nk > 512: 
        import os
        if os.environ.get("LDT_ATTN_FORCE") == "2": continue
    q = torch.randn(B * Nq, d, device="cuda").to(torch.bfloat16); kv = torch.randn(B * Nk, 2 * d, device="cuda").to(torch.bfloat16)
    o = torch.empty(B, H, Nq, dh, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.attention_fwd(q, kv[:, :d], kv[:, d:], B, H, Nq, Nk, dh, out=o)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print("  %%dx%%d: %%.1f us" %% (Nq, Nk, e0.elapsed_time(e1) / 30 * 1e3), end="")
print()
''' % ROOT
for rnd in range(3):
    for name, env in (("streaming", {"LDT_ATTN_FORCE": "1"}), ("default", {}), ("resident qsplit 1", {"LDT_ATTN_FORCE": "2", "LDT_ATTN_QSPLIT": "1"}),
                      ("resident qsplit 2", {"LDT_ATTN_FORCE": "2", "LDT_ATTN_QSPLIT": "2"}), ("resident qsplit 4", {"LDT_ATTN_FORCE": "2", "LDT_ATTN_QSPLIT": "4"}),
                      ("resident qsplit 8", {"LDT_ATTN_FORCE": "2", "LDT_ATTN_QSPLIT": "8"}), ("resident qsplit 16", {"LDT_ATTN_FORCE": "2", "LDT_ATTN_QSPLIT": "16"})):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("%-20s %s" % (name, out.stdout.strip() or out.stderr[-300:]), flush=True)
