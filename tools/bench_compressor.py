#!/usr/bin/env python3
"""BASELINE configs[3]: Compressor encode + decode only (2048 points -> T latent tokens -> 2048 points), batch 1024,
1 GPU.  Prints one JSON line: clouds/s for encode (Compressor.forward, which also reconstructs) and decode
(Compressor.sample), plus the cross-attention kernel microbench (Q = 2048 points x K/V = T tokens and the reverse)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=1024, help="clouds per encode pass (one call for the whole batch measured fastest)")
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    import ldt_amd
    from ldt_amd import ops
    cfg = ldt_amd.airplane_config(latent_tokens=a.tokens)
    torch.manual_seed(0)
    comp = ldt_amd.Compressor(cfg.compressor).cuda()
    comp.init()
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(a.batch, 2048, 3, generator=g)
    pts = pts - pts.mean(1, keepdim=True)
    pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()

    def encode():
        outs = [comp(pts[i:i + a.chunk])["all_eps"] for i in range(0, a.batch, a.chunk)]
        return torch.cat(outs)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps, r

    t_enc, eps = timed(encode)
    t_dec, dec = timed(lambda: comp.sample((a.batch, 2048), given_eps=eps))
    assert torch.isfinite(dec).all()
    # cross-attention microbench at the decoder's shape (d=128, 4 heads x 32)
    B, H, dh, T = 128, 4, 32, a.tokens
    q = torch.randn(B * 2048, 128, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B * T, 256, device="cuda").to(torch.bfloat16)
    o = torch.empty(B, H, 2048, dh, device="cuda", dtype=torch.bfloat16)
    def attn_time(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    t1 = attn_time(lambda: ops.attention_fwd(q, kv[:, :128], kv[:, 128:], B, H, 2048, T, dh, out=o))
    b1 = (2 * B * 2048 * 128 + 2 * B * T * 128) * 2
    q2 = torch.randn(B * T, 128, device="cuda").to(torch.bfloat16)
    kv2 = torch.randn(B * 2048, 256, device="cuda").to(torch.bfloat16)
    o2 = torch.empty(B, H, T, dh, device="cuda", dtype=torch.bfloat16)
    t2 = attn_time(lambda: ops.attention_fwd(q2, kv2[:, :128], kv2[:, 128:], B, H, T, 2048, dh, out=o2))
    b2 = (2 * B * T * 128 + 2 * B * 2048 * 128) * 2
    print(json.dumps({
        "workload": "BASELINE configs[3]: Compressor encode+decode, batch %d, 2048 pts, %d tokens" % (a.batch, a.tokens),
        "encode_clouds_per_s": round(a.batch / t_enc, 1), "decode_clouds_per_s": round(a.batch / t_dec, 1),
        "encode_s": round(t_enc, 4), "decode_s": round(t_dec, 4),
        "cross_attn_q2048_kvT": {"us": round(t1 * 1e6, 1), "GBps": round(b1 / t1 / 1e9, 1), "batch": B},
        "cross_attn_qT_kv2048": {"us": round(t2 * 1e6, 1), "GBps": round(b2 / t2 / 1e9, 1), "batch": B}}))


if __name__ == "__main__":
    main()
