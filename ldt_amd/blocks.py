"""Host-side orchestration of one Set-Transformer `ResidualBlock` (reference model/layers.py:183-229) on the
HIP kernels — used by the Compressor (d=128, 4 heads x 32).  The Score network has its own C++
orchestrator (`ldt_score_forward`) because it runs 24 blocks x 1000 steps.

    AdaLN block (Encoder, c = pos embedding):  x += g1 * Attn(mod(LN(x)), y) ; x += g2 * MLP(mod(LN(x)))
    plain block (Decoder, c = None):           x += Attn(LN_affine(x), y)    ; x += MLP(LN_affine(x))

K/V source y (quirk Q2): None -> the normalised/modulated x itself; otherwise the RAW tensor given.
Head merge (quirk Q1): the attention kernel writes [B][H][N][Dh]; that buffer *is* the (B,N,C) raw reinterpret.
"""
import torch

from . import ops
from ._lib import ACT_SILU, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, block_act_id
from .layers import conv_w


# the fused LN+MLP kernel serves 64- and 128-channel blocks; LDT_FUSED_MLP=0 keeps the three-kernel path (A/B runs)
import os
FUSED_MLP = os.environ.get("LDT_FUSED_MLP", "1") != "0"
FUSED_ATTN = os.environ.get("LDT_FUSED_ATTN", "1") != "0"


def _bf(w):
    w = w.detach().float().contiguous()
    return ops.cast_pad_bf16(w, ops.pad64(w.shape[1]))


def pack_block(blk):
    """bf16 operand panels + fp32 vectors of one ResidualBlock holder."""
    C, Co = blk.dim_in, blk.dim_out
    wkv = conv_w(blk.fc_kv)
    P = {
        "C": C, "Co": Co, "H": blk.num_heads,
        "wq": _bf(conv_w(blk.fc_q)), "bq": blk.fc_q.bias.detach().float().contiguous(),
        "wkv": _bf(wkv), "bkv": blk.fc_kv.bias.detach().float().contiguous(),
        "wo": _bf(conv_w(blk.fc_o)), "bo": blk.fc_o.bias.detach().float().contiguous(),
        "wup": _bf(conv_w(blk.mlp.fc[0][0])), "bup": blk.mlp.fc[0][0].bias.detach().float().contiguous(),
        "wdn": _bf(conv_w(blk.mlp.out)), "bdn": blk.mlp.out.bias.detach().float().contiguous(),
        "n1": tuple(None if p is None else p.detach().float().contiguous() for p in blk.norm1.affine),
        "n2": tuple(None if p is None else p.detach().float().contiguous() for p in blk.norm2.affine),
        "act": block_act_id(getattr(blk, "act", None)),             # activation behind the norms of the no-condition branch (0 = Identity)
        "norm": getattr(blk.norm1, "kind", "layer_norm"),            # layer_norm | group_norm | None (tools/utils.py:168-181)
        "groups": (getattr(blk.norm1, "num_groups", 0), getattr(blk.norm2, "num_groups", 0)),
    }
    if C == Co and C in (64, 128) and blk.dim_kv == C:              # fc_q | fc_kv as one operand for the fused LN + linear kernel
        P["wqkv"] = torch.cat([P["wq"], P["wkv"]], 0).contiguous()
        P["bqkv"] = torch.cat([P["bq"], P["bkv"]], 0).contiguous()
    if blk.dim_c is not None and C == Co:
        lin = blk.adaLN[1]
        P["wada"], P["bada"] = lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous()
    elif blk.dim_c is not None:                                      # U-Net down block: two adaLN heads + conv shortcut
        P["wada1"], P["bada1"] = (t.detach().float().contiguous() for t in (blk.adaLN1[1].weight, blk.adaLN1[1].bias))
        P["wada2"], P["bada2"] = (t.detach().float().contiguous() for t in (blk.adaLN2[1].weight, blk.adaLN2[1].bias))
        P["wsc"], P["bsc"] = _bf(conv_w(blk.shortcut)), blk.shortcut.bias.detach().float().contiguous()
    return P


def apply_norm(kind, x, B, N, affine=(None, None), groups=0, shift=None, scale=None, mod_sample_stride=0, rows_per_sample=0):
    """norm(x) [* w + b] then modulate (layers.py:136-137) -> bf16, for the three norms get_norm builds (tools/utils.py:168-181):
    layer_norm (the LayerNorm kernel; affine only when the block has no condition), group_norm (statistics per sample and group over
    C / G channels x the sample's N tokens — the reference applies nn.GroupNorm to the channels-first (B, C, N) tensor — always affine),
    None (identity).  x: token-major fp32 [B*N, C]."""
    if kind == "layer_norm":
        kw = dict(shift=shift, scale=scale, mod_sample_stride=mod_sample_stride, rows_per_sample=rows_per_sample) if shift is not None else {}
        return ops.layernorm_modulate(x, w=affine[0], b=affine[1], **kw)
    stats = ops.group_stats(x, B, N, groups) if kind == "group_norm" else None
    w, b = affine if kind == "group_norm" else (None, None)
    return ops.norm_apply(x, stats=stats, rows_per_stat=N, w=w, b=b, shift=shift, scale=scale, mod_sample_stride=mod_sample_stride,
                          rows_per_sample=rows_per_sample if shift is not None else 1)


def residual_block(P, x, B, Nq, y_bf16=None, Nk=None, c=None, per_token=False, x_bf16_out=None, q_pre=None, next_P=None, kv_pre=None):
    """x fp32 [B*Nq, C] updated IN PLACE (and returned) when dim_out == dim_in; a NEW [B*Nq, dim_out] tensor is returned
    for a U-Net down block.  y_bf16: raw K/V source [B*Nk, Ckv] (bf16) or None (self, modulated).
    c: fp32 [B, dim_c] condition (AdaLN) — or [B*Nq, dim_c] with per_token=True (layers.py:210: a (B, dim_c, N) condition
    modulates every token with its own row; the Compressor's `pos_embedding: mlp`) — or None (plain LayerNorm block).
    x_bf16_out: optional bf16 [B*Nq, dim_out] buffer that receives a copy of the block's result (written by the fused MLP
    kernel's own store pass, or by one cast on the unfused path).
    kv_pre: bf16 [B*Nk, 2*dim_out] = fc_kv(y) computed by the caller (then y_bf16 is not needed: pass Nk).
    next_P: packed holder of the plain-LayerNorm block that will run next on the same rows with a K/V source of its own (the
    next decoder level): its LN1 + fc_q is then computed by THIS block's last kernel and `(x, q_next)` is returned — or
    `(x, None)` when the shapes do not allow it; pass that `q_next` to the next call as `q_pre`."""
    C, Co, H = P["C"], P["Co"], P["H"]
    rps = 1 if per_token else Nq                                                    # rows that share one modulation row
    ln_kw = {}
    if c is not None and C == Co:
        mod = ops.sgemm(c, P["wada"], P["bada"], act_in=ACT_SILU)                  # [B, 6C]  layers.py:214
        sh1, sc1, g1, sh2, sc2, g2 = (mod[:, i * C:(i + 1) * C] for i in range(6))
        s1 = s2 = 6 * C
        ln_kw = dict(shift=sh1, scale=sc1, mod_sample_stride=s1, rows_per_sample=rps)
    elif c is not None:                                                             # layers.py:216-217
        m1 = ops.sgemm(c, P["wada1"], P["bada1"], act_in=ACT_SILU)                 # [B, 2C]   shift_msa | scale_msa
        mod = ops.sgemm(c, P["wada2"], P["bada2"], act_in=ACT_SILU)                # [B, 4Co]  gate_msa | shift_mlp | scale_mlp | gate_mlp
        g1, sh2, sc2, g2 = (mod[:, i * Co:(i + 1) * Co] for i in range(4))
        s2 = 4 * Co
        ln_kw = dict(shift=m1[:, :C], scale=m1[:, C:], mod_sample_stride=2 * C, rows_per_sample=rps)
    else:
        g1 = g2 = None
        s2 = 0
    # no-condition branch with a block activation (layers.py:224-226, `decoder_act`): act(norm1(x)) feeds fc_q (and fc_kv in self-attention),
    # act(norm2(x)) the MLP — the LayerNorm kernels + one element-wise pass + the GEMMs (the fused LN kernels have no activation slot)
    act = P.get("act", 0) if c is None else 0
    kind = P.get("norm", "layer_norm")
    ln = kind == "layer_norm"                                        # the fused LN kernels are LayerNorm kernels: other norms take the chain below
    fused_in = FUSED_ATTN and C in (64, 128) and P["wq"].shape[1] == C and Co % 64 == 0 and x.stride(0) % 4 == 0 and not act and ln
    if fused_in:
        # LN1 (+ modulate | affine) + fc_q [+ fc_kv on the same normalised input] in ONE kernel (csrc/fused_mlp.hip):
        # the normalised activations are never written
        aff = {} if c is not None else dict(ln_w=P["n1"][0], ln_b=P["n1"][1])
        if y_bf16 is None and kv_pre is None and "wqkv" in P:
            qkv = ops.ln_linear(x, P["wqkv"], P["bqkv"], **aff, **ln_kw)
            q, kv, Nk = qkv[:, :Co], qkv[:, Co:], Nq
        else:
            # (q_pre: this block's LN1 + fc_q was already computed by the previous block's MLP kernel)
            ext = y_bf16 is not None or kv_pre is not None                           # K/V come from another tensor
            q = q_pre if (q_pre is not None and ext) else ops.ln_linear(x, P["wq"], P["bq"], **aff, **ln_kw)
            if not ext:                                                             # (not reached with the shipped shapes)
                y_bf16, Nk = ops.layernorm_modulate(x, **({"w": P["n1"][0], "b": P["n1"][1]} if c is None else ln_kw)), Nq
            kv = kv_pre if kv_pre is not None else ops.gemm_bf16(y_bf16, P["wkv"], P["bkv"], EPI_BF16)
    else:
        if c is not None:
            h = ops.layernorm_modulate(x, **ln_kw) if ln else apply_norm(kind, x, B, Nq, P["n1"], P["groups"][0], **ln_kw)
        else:
            h = apply_norm(kind, x, B, Nq, P["n1"], P["groups"][0])
            if act:
                ops.block_activation_(h, act)
        q = ops.gemm_bf16(h, P["wq"], P["bq"], EPI_BF16)
        if y_bf16 is None and kv_pre is None:
            y_bf16, Nk = h, Nq
        kv = kv_pre if kv_pre is not None else ops.gemm_bf16(y_bf16, P["wkv"], P["bkv"], EPI_BF16)   # [B*Nk, 2Co]: K | V  (layers.py:189)
    if C != Co:                                                                     # shortcut(x): Conv1d dim_in -> dim_out
        from ._lib import EPI_F32
        x = ops.gemm_bf16(ops.cast_pad_bf16(x, ops.pad64(C)), P["wsc"], P["bsc"], EPI_F32)
    if FUSED_ATTN and Co // H == 32 and H in (2, 4) and Nq % H == 0 and P["wo"].shape == (Co, Co) and not (per_token and g1 is not None):
        # attention + out-projection + gated residual in ONE kernel (csrc/attention.hip, OPROJ): the [B,H,Nq,Dh]
        # result never goes to HBM — the Compressor's d = 128 blocks
        ops.attention_oproj_resid_(q, kv[:, :Co], kv[:, Co:], B, H, Nq, Nk, 32, P["wo"], P["bo"], x, gate=g1,
                                   gate_sample_stride=s2 if g1 is not None else 0)
    else:
        a = ops.attention_fwd(q, kv[:, :Co], kv[:, Co:], B, H, Nq, Nk, Co // H)     # [B,H,Nq,Dh] == (B*Nq, Co) raw view
        ops.gemm_bf16(a.view(B * Nq, Co), P["wo"], P["bo"], EPI_RESID_F32, out=x, resid=x, gate=g1,
                      gate_sample_stride=s2 if g1 is not None else 0, rows_per_sample=rps)
    if FUSED_MLP and Co in (64, 128) and P["wup"].shape == (4 * Co, Co) and x.stride(0) % 4 == 0 and not act and ln:
        # LN2 + MLP + gated residual in ONE pass over x (csrc/fused_mlp.hip) — the Compressor's d = 128 blocks
        nxt = None
        if next_P is not None and FUSED_ATTN and next_P["C"] == next_P["Co"] == Co and tuple(next_P["wq"].shape) == (Co, Co) \
                and Co % 64 == 0 and next_P["n1"][0] is not None:
            nxt = dict(w=next_P["wq"], bias=next_P["bq"], ln_w=next_P["n1"][0], ln_b=next_P["n1"][1])
        if c is not None:
            r = ops.ln_mlp_resid_(x, P["wup"], P["bup"], P["wdn"], P["bdn"], shift=sh2, scale=sc2, gate=g2,
                                  mod_sample_stride=s2, rows_per_sample=rps, x_bf16_out=x_bf16_out, next_linear=nxt)
        else:
            r = ops.ln_mlp_resid_(x, P["wup"], P["bup"], P["wdn"], P["bdn"], ln_w=P["n2"][0], ln_b=P["n2"][1], x_bf16_out=x_bf16_out,
                                  next_linear=nxt)
        if next_P is not None:
            return (x, r[1]) if nxt is not None else (x, None)
        return x
    if c is not None:
        h2 = apply_norm(kind, x, B, Nq, P["n2"], P["groups"][1], shift=sh2, scale=sc2, mod_sample_stride=s2, rows_per_sample=rps)
    else:
        h2 = apply_norm(kind, x, B, Nq, P["n2"], P["groups"][1])
        if act:
            ops.block_activation_(h2, act)
    u = ops.gemm_bf16(h2, P["wup"], P["bup"], EPI_GELU_BF16)
    ops.gemm_bf16(u, P["wdn"], P["bdn"], EPI_RESID_F32, out=x, resid=x, gate=g2,
                  gate_sample_stride=s2 if g2 is not None else 0, rows_per_sample=rps)
    if x_bf16_out is not None:
        x_bf16_out.copy_(ops.cast_pad_bf16(x, x_bf16_out.shape[1]))
    return (x, None) if next_P is not None else x


def pack_final(fl):
    lin = fl.adaLN[1]
    return {"C": fl.ln.in_channels, "w": _bf(conv_w(fl.ln)), "b": fl.ln.bias.detach().float().contiguous(),
            "n_out": fl.ln.out_channels, "norm": getattr(fl.norm, "kind", "layer_norm"), "groups": getattr(fl.norm, "num_groups", 0),
            "n": tuple(None if p is None else p.detach().float().contiguous() for p in fl.norm.affine),
            "wada": lin.weight.detach().float().contiguous(), "bada": lin.bias.detach().float().contiguous()}


def final_layer(P, x, B, N, c, out_dtype=torch.float32, per_token=False):
    """FinalLayer with condition (layers.py:240-246): Conv(mod(LN(x))); c [B, dim_c], or [B*N, dim_c] with per_token."""
    from ._lib import EPI_F32
    C = P["C"]
    mod = ops.sgemm(c, P["wada"], P["bada"], act_in=ACT_SILU)                      # [B, 2C] shift | scale
    h = apply_norm(P.get("norm", "layer_norm"), x, B, N, P.get("n", (None, None)), P.get("groups", 0), shift=mod[:, :C], scale=mod[:, C:],
                   mod_sample_stride=2 * C, rows_per_sample=1 if per_token else N)
    return ops.gemm_bf16(h, P["w"], P["b"], EPI_F32 if out_dtype == torch.float32 else EPI_BF16, n=P["n_out"])
