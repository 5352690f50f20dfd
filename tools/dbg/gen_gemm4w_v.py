#!/usr/bin/env python3
"""Generator of the hand-placed K loop of tools/dbg/gemm4w_v.hip (experiment, round 5): a 256 x 256 tile, FOUR waves (one per SIMD,
128 x 128 per wave, 256 accumulator AGPRs), 64-deep K-tiles through TWO 64-KiB LDS buffers, the whole K loop of a tile as ONE
`asm volatile` statement with literal registers — what the vendor library's kernel for these shapes does
(Custom_Cijk_Alik_Bljk_BBS_BH_MT256x256x64_MI16x16x1, read from its disassembly: DESIGN.md §4 "the vendor kernel"), re-derived
for this repository's LDS image (128-B rows, 16-B chunk index XOR (row >> 1) & 7) and operand conventions.

Per K-tile t (buffer b = t & 1), per wave:
  phase A   64 MFMAs on k-half 0 | 16 ds_read_b128: k-half 1 of tile t (buffer b) -> lgkmcnt(0), s_barrier (buffer b is free)
                                 | 8 LDS-DMA pieces: X of tile t+2 -> buffer b
  phase B   64 MFMAs on k-half 1 | 8 LDS-DMA pieces: W of tile t+2 -> buffer b | vmcnt(16) (tile t+1 landed), s_barrier
                                 | 16 ds_read_b128: k-half 0 of tile t+1 (buffer b ^ 1) -> lgkmcnt(0)
No VALU in the loop except four v_xor (buffer toggle of the read addresses); DMA addresses = buffer descriptor (base advanced by
s_add) + a per-lane offset VGPR that never changes + a per-piece SGPR offset.

Emits tools/dbg/gemm4w_v_loop.inc: the asm template string (C string literals) for the statement's text.
Schedule knobs (env): RD_EVERY (MFMAs per ds_read gap, default 2), DMA_EVERY (default 3)."""
import os
import sys

RD_EVERY = int(os.environ.get("RD_EVERY", "2"))
DMA_EVERY = int(os.environ.get("DMA_EVERY", "3"))
NO_DMA = int(os.environ.get("NO_DMA", "0"))       # ablations (timing only: results are wrong)
NO_READ = int(os.environ.get("NO_READ", "0"))
NO_BAR = int(os.environ.get("NO_BAR", "0"))
# literal registers (clobbered by the statement)
W0, X0, W1, X1 = 120, 152, 184, 216          # fragment sets: k-half 0 / 1, 8 fragments x 4 VGPRs each
SRDX, SRDW = 44, 48                          # s[44:47], s[48:51]
SOX, SOW = 52, 60                            # per-piece scalar offsets s[52:59], s[60:67]
CNT, M0X, TMP = 68, 69, 70


def frag(base, i):
    return "v[%d:%d]" % (base + 4 * i, base + 4 * i + 3)


def acc(k, p):
    b = (k * 8 + p) * 4
    return "a[%d:%d]" % (b, b + 3)


def mfma(i, half, first):
    k, p = i // 8, i % 8
    wf, xf = (W0, X0) if half == 0 else (W1, X1)
    c = "0" if first else acc(k, p)
    return "v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc(k, p), frag(wf, k), frag(xf, p), c)


def reads(half):
    """16 fragment reads of one k-half in first-needed order: W[0], X[0..7], W[1..7]"""
    wf, xf = (W0, X0) if half == 0 else (W1, X1)
    wa, xa = ("%[wa0]", "%[xa0]") if half == 0 else ("%[wa1]", "%[xa1]")
    out = ["ds_read_b128 %s, %s" % (frag(wf, 0), wa)]
    out += ["ds_read_b128 %s, %s%s" % (frag(xf, p), xa, " offset:%d" % (p * 2048) if p else "") for p in range(8)]
    out += ["ds_read_b128 %s, %s offset:%d" % (frag(wf, k), wa, k * 2048) for k in range(1, 8)]
    return out


def dma(which, q):
    srd, so, vo = (SRDX, SOX, "%[vox") if which == "x" else (SRDW, SOW, "%[vow")
    return "buffer_load_dwordx4 %s%d], s[%d:%d], s%d offen lds" % (vo, q & 1, srd, srd + 3, so + q)


VARIANT = int(os.environ.get("VARIANT", "2"))    # 2 = two barriers per K-tile (above), 3 = the vendor kernel's placement (three barriers)


def body_v3(first, with_dma, read_next):
    """the vendor kernel's placement: X fragments first, barrier, X pieces of K-tile t+2 interleaved with the W fragment reads, barrier,
    the rest; 13 pieces in flight behind the `tile t+1 landed` wait, the last three pieces of a K-tile trail into the next-tile reads."""
    rd1, rd0 = reads(1), reads(0)
    x1, w1 = rd1[1:9], [rd1[0]] + rd1[9:]
    x0, w0 = rd0[1:9], [rd0[0]] + rd0[9:]
    A, B = {}, {}
    put = lambda D, g, ins: D.setdefault(g, []).append(ins)
    for i, r in enumerate(x1):
        put(A, 2 * i, r)
    put(A, 21, "s_waitcnt lgkmcnt(0)")
    if with_dma:
        put(A, 22, "s_barrier"); put(A, 22, "s_mov_b32 m0, s%d" % M0X)
        for i, g in enumerate((23, 26, 29, 32, 35)):
            put(A, g, dma("x", i)); put(A, g + 1, "s_add_u32 m0, m0, 1024")
    for i, g in enumerate((25, 28, 31, 34, 37, 39, 41, 43)):
        put(A, g, w1[i])
    put(A, 50, "s_waitcnt lgkmcnt(0)")
    put(A, 50, "v_xor_b32 %[xa1], 0x10000, %[xa1]"); put(A, 50, "v_xor_b32 %[wa1], 0x10000, %[wa1]")
    if with_dma:
        put(A, 51, "s_barrier")
        for i, g in enumerate((52, 55, 58)):
            put(A, g, dma("x", 5 + i))
            if i < 2:
                put(A, g + 1, "s_add_u32 m0, m0, 1024")
        put(A, 59, "s_add_u32 s%d, s%d, 128" % (SRDX, SRDX)); put(A, 59, "s_addc_u32 s%d, s%d, 0" % (SRDX + 1, SRDX + 1))
        put(A, 60, "s_add_u32 m0, s%d, 32768" % M0X)
        put(A, 62, dma("w", 0)); put(A, 63, "s_add_u32 m0, m0, 1024")
        put(B, 1, dma("w", 1)); put(B, 2, "s_add_u32 m0, m0, 1024")
        for i, g in enumerate((21, 24, 27)):
            put(B, g, dma("w", 2 + i)); put(B, g + 1, "s_add_u32 m0, m0, 1024")
    if read_next:
        put(B, 28, "s_waitcnt vmcnt(%d)" % (13 if with_dma else 0))
        put(B, 29, "s_barrier")
        for i, g in enumerate((30, 31, 33, 35, 37, 39, 41, 43)):
            put(B, g, x0[i])
        for i, g in enumerate((45, 47, 49, 51, 53, 55, 57, 59)):
            put(B, g, w0[i])
        put(B, 62, "v_xor_b32 %[xa0], 0x10000, %[xa0]"); put(B, 62, "v_xor_b32 %[wa0], 0x10000, %[wa0]")
        put(B, 63, "s_waitcnt lgkmcnt(0)")
    if with_dma:
        put(B, 36, dma("w", 5)); put(B, 38, "s_add_u32 m0, m0, 1024")
        put(B, 42, dma("w", 6)); put(B, 44, "s_add_u32 m0, m0, 1024")
        put(B, 60, dma("w", 7))
        put(B, 61, "s_add_u32 s%d, s%d, 128" % (SRDW, SRDW)); put(B, 61, "s_addc_u32 s%d, s%d, 0" % (SRDW + 1, SRDW + 1))
        put(B, 61, "s_xor_b32 s%d, s%d, 0x10000" % (M0X, M0X))
    L = []
    for i in range(64):
        L.append(mfma(i, 0, first)); L += A.get(i, [])
    for i in range(64):
        L.append(mfma(i, 1, False)); L += B.get(i, [])
    return L


def body(first, with_dma, read_next, last):
    """one K-tile: list of instruction strings"""
    if VARIANT == 3:
        return body_v3(first, with_dma, read_next)
    L = []
    # ---------------- phase A
    fill = {}                                   # gap index (after MFMA i) -> [instructions]
    rd = reads(1)
    g = 0
    for r in rd:
        fill.setdefault(g, []).append(r)
        g += RD_EVERY
    g_reads_done = g
    fill.setdefault(g_reads_done, []).append("s_waitcnt lgkmcnt(0)")
    fill.setdefault(g_reads_done, []).append("v_xor_b32 %[xa1], 0x10000, %[xa1]")
    fill.setdefault(g_reads_done, []).append("v_xor_b32 %[wa1], 0x10000, %[wa1]")
    if with_dma:
        fill.setdefault(g_reads_done + 1, []).append("s_barrier")
        fill.setdefault(g_reads_done + 1, []).append("s_mov_b32 m0, s%d" % M0X)
        g = g_reads_done + 2
        for q in range(8):
            fill.setdefault(g, []).append(dma("x", q))
            fill.setdefault(g + 1, []).append("s_add_u32 m0, m0, 1024")
            g += DMA_EVERY
        assert g <= 64 + DMA_EVERY, "phase A too short for its DMA pieces: %d" % g
        fill.setdefault(63, []).append("s_add_u32 s%d, s%d, 128" % (SRDX, SRDX))
        fill.setdefault(63, []).append("s_addc_u32 s%d, s%d, 0" % (SRDX + 1, SRDX + 1))
    for i in range(64):
        L.append(mfma(i, 0, first))
        L += fill.get(i, [])
    # ---------------- phase B
    fill = {}
    g = 0
    if with_dma:
        fill.setdefault(0, []).append("s_add_u32 m0, s%d, 32768" % M0X)
        g = 1
        for q in range(8):
            fill.setdefault(g, []).append(dma("w", q))
            fill.setdefault(g + 1, []).append("s_add_u32 m0, m0, 1024")
            g += DMA_EVERY
        fill.setdefault(g, []).append("s_add_u32 s%d, s%d, 128" % (SRDW, SRDW))
        fill.setdefault(g, []).append("s_addc_u32 s%d, s%d, 0" % (SRDW + 1, SRDW + 1))
        fill.setdefault(g, []).append("s_xor_b32 s%d, s%d, 0x10000" % (M0X, M0X))
        g += 1
    if read_next:
        fill.setdefault(g, []).append("s_waitcnt vmcnt(%d)" % (16 if with_dma else 0))
        fill.setdefault(g + 1, []).append("s_barrier")
        g += 2
        for r in reads(0):
            fill.setdefault(g, []).append(r)
            g += RD_EVERY
        assert g <= 62 + RD_EVERY, "phase B too short: %d" % g
        fill.setdefault(62, []).append("v_xor_b32 %[xa0], 0x10000, %[xa0]")
        fill.setdefault(62, []).append("v_xor_b32 %[wa0], 0x10000, %[wa0]")
        fill.setdefault(63, []).append("s_waitcnt lgkmcnt(0)")
    for i in range(64):
        L.append(mfma(i, 1, False))
        L += fill.get(i, [])
    return L


def emit():
    T = []
    # ---- set-up: descriptors and per-piece offsets into literal SGPRs
    for j, nm in enumerate(("srdx0", "srdx1", "srdx2", "srdx3")):
        T.append("s_mov_b32 s%d, %%[%s]" % (SRDX + j, nm))
    for j, nm in enumerate(("srdw0", "srdw1", "srdw2", "srdw3")):
        T.append("s_mov_b32 s%d, %%[%s]" % (SRDW + j, nm))
    T.append("s_mov_b32 s%d, %%[sx0]" % SOX)
    for q in range(1, 8):
        T.append("s_add_u32 s%d, s%d, %%[sxs]" % (SOX + q, SOX + q - 1))
    T.append("s_mov_b32 s%d, %%[sw0]" % SOW)
    for q in range(1, 8):
        T.append("s_add_u32 s%d, s%d, %%[sws]" % (SOW + q, SOW + q - 1))
    T.append("s_mov_b32 s%d, %%[ldsx]" % M0X)
    # ---- prologue: K-tiles 0 and 1 -> buffers 0 and 1
    for t in range(2):
        T.append("s_mov_b32 m0, s%d" % M0X)
        for q in range(8):
            T.append("s_nop 0")
            T.append(dma("x", q))
            T.append("s_add_u32 m0, m0, 1024")
        T.append("s_add_u32 m0, s%d, 32768" % M0X)
        for q in range(8):
            T.append("s_nop 0")
            T.append(dma("w", q))
            T.append("s_add_u32 m0, m0, 1024")
        T.append("s_add_u32 s%d, s%d, 128" % (SRDX, SRDX)); T.append("s_addc_u32 s%d, s%d, 0" % (SRDX + 1, SRDX + 1))
        T.append("s_add_u32 s%d, s%d, 128" % (SRDW, SRDW)); T.append("s_addc_u32 s%d, s%d, 0" % (SRDW + 1, SRDW + 1))
        T.append("s_xor_b32 s%d, s%d, 0x10000" % (M0X, M0X))
    T.append("s_waitcnt vmcnt(16)")
    T.append("s_barrier")
    T += reads(0)
    T.append("v_xor_b32 %[xa0], 0x10000, %[xa0]")
    T.append("v_xor_b32 %[wa0], 0x10000, %[wa0]")
    T.append("s_waitcnt lgkmcnt(0)")
    # ---- K-tile 0 (accumulators start from 0), K-tiles 1 .. nkt-3 (loop), nkt-2, nkt-1   (launcher: nkt >= 4)
    T += body(True, True, True, False)
    T.append("s_sub_u32 s%d, %%[nkt], 3" % CNT)
    T.append("1:")
    T += body(False, True, True, False)
    T.append("s_sub_u32 s%d, s%d, 1" % (CNT, CNT))
    T.append("s_cmp_lg_u32 s%d, 0" % CNT)
    T.append("s_cbranch_scc1 1b")
    T += body(False, False, True, False)
    T += body(False, False, False, True)
    T.append("s_nop 15")
    T.append("s_nop 15")
    return T


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm4w_v_loop.inc")
    T = emit()
    with open(out, "w") as f:
        f.write("// generated by tools/dbg/gen_gemm4w_v.py (RD_EVERY=%d DMA_EVERY=%d): %d instructions\n" % (RD_EVERY, DMA_EVERY, len(T)))
        for n_, ln in enumerate(T):
            if (NO_DMA and ln.startswith("buffer_load") and n_ > 120) or (NO_READ and ln.startswith("ds_read")) or (NO_BAR and ln == "s_barrier"):
                ln = "s_nop 0"
            f.write('"%s\\n\\t"\n' % ln)
    print("wrote %s: %d lines" % (out, len(T)))
