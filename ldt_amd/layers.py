"""Parameter containers with the reference's module tree (names AND construction order), so that
reference `state_dict`s load verbatim and `torch.manual_seed(s); Score(cfg)` draws the same
default-init weights as upstream.  They hold weights only: the arithmetic of these layers runs in
libldt_hip.so, orchestrated by ldt_amd/score.py and ldt_amd/compressor.py.

Reference: model/layers.py (TimeEmbedding :14-41, LabelEmbedding :44-52, ActNorm :55-107, MLP :110-133,
ResidualBlock :140-229, FinalLayer :232-248), tools/utils.py:127-133 (LayerNorm wrapper).
"""
import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("%s only holds parameters; use ldt_amd.Score / ldt_amd.Compressor (HIP path)"
                           % type(self).__name__)


class LayerNormC(_Holder):
    """tools/utils.py:127-133 — keys `<name>.norm.{weight,bias}` when affine."""
    kind = "layer_norm"

    def __init__(self, channels, elementwise_affine):
        super().__init__()
        self.norm = nn.LayerNorm(channels, elementwise_affine=elementwise_affine, eps=1e-6)

    @property
    def affine(self):
        return (self.norm.weight, self.norm.bias) if self.norm.elementwise_affine else (None, None)


class IdentityNorm(_Holder):
    """`norm: ~` -> nn.Identity (tools/utils.py:169-170): no parameters."""
    kind = None
    affine = (None, None)


class GroupNormC(_Holder):
    """`norm: group_norm` -> nn.GroupNorm(min(C // 4, groups), C, eps=1e-6) returned AS IS by get_norm (tools/utils.py:177-179): the keys are
    `<name>.weight` / `<name>.bias` (no `.norm.` infix), the affine is always there (elementwise_affine is ignored upstream; ones / zeros at
    construction: no generator draw), and it is applied to the channels-first (B, C, N) activations: statistics per sample and group over
    C / G channels x all tokens.  Parameter holder only (ldt_group_stats / ldt_norm_apply do the arithmetic)."""
    kind = "group_norm"

    def __init__(self, channels, groups=16):
        super().__init__()
        self.num_groups = min(channels // 4, groups)
        self.num_channels = channels
        self.eps = 1e-6
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    @property
    def affine(self):
        return (self.weight, self.bias)


def make_norm(channels, kind, elementwise_affine):
    """tools/utils.py:168-181 get_norm: layer_norm (every shipped config), group_norm, None.  `batch_norm` is refused with the reason:
    upstream's BatchNorm1d wrapper (tools/utils.py:136-142) transposes the (B, C, N) activations to (B, N, C) before nn.BatchNorm1d(C), which
    then normalises over the TOKEN axis and fails with "running_mean should contain N elements not C" unless tokens == channels."""
    if kind is None:
        return IdentityNorm()
    k = str(kind).lower()
    if k == "layer_norm":
        return LayerNormC(channels, elementwise_affine)
    if k == "group_norm":
        return GroupNormC(channels, 16)
    if k == "batch_norm":
        raise NotImplementedError("norm='batch_norm': undefined upstream — its wrapper feeds (B, tokens, channels) to BatchNorm1d(channels), "
                                  "which raises unless tokens == channels (tools/utils.py:136-142)")
    raise TypeError("norm not support")                              # tools/utils.py:181


class TimeEmbedding(_Holder):
    def __init__(self, dim_embed, dim_out):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(dim_embed, dim_out), nn.SiLU(), nn.Linear(dim_out, dim_out))
        self.t_emb_dim = dim_embed


class LabelEmbedding(_Holder):
    def __init__(self, num_categorys, dim_embed, dim_out):
        super().__init__()
        self.label_emb = nn.Embedding(num_categorys, dim_embed)
        self.mlp = nn.Sequential(nn.Linear(dim_embed, dim_out), nn.SiLU(), nn.Linear(dim_out, dim_out))
        self.t_emb_dim = dim_embed


class ActNorm(_Holder):
    """model/layers.py:55-107; the Compressor builds it with feature_type=cfg.ActNorm (True) => per-token
    parameters of shape (1, z_scale, C) (:95-101)."""

    def __init__(self, num_features, z_scale, data_dep_init=True, eps=1e-6, feature_type="set"):
        super().__init__()
        self.num_features, self.z_scale, self.eps, self.feature_type = num_features, z_scale, eps, feature_type
        self.register_buffer("initialized", torch.zeros(1) if data_dep_init else torch.ones(1))
        shape = (1, 1, num_features) if feature_type == "set" else (1, z_scale, num_features)
        self.shift = nn.Parameter(torch.zeros(shape))
        self.log_scale = nn.Parameter(torch.zeros(shape))

    def init(self):
        self.initialized += 1.


class MLP(_Holder):
    def __init__(self, dim_in, dim_hidden, dim_out, n_hidden, activation="gelu", residual=False, dropout_p=0.):
        super().__init__()
        if activation != "gelu" or residual or dropout_p > 0 or n_hidden not in (0, 1):
            raise NotImplementedError("MLP variant not on the shipped path")
        self.fc = nn.ModuleList()
        for i in range(n_hidden):
            self.fc.append(nn.Sequential(nn.Conv1d(dim_in if i == 0 else dim_hidden, dim_hidden, 1)))
        self.out = nn.Conv1d(dim_hidden if n_hidden > 0 else dim_in, dim_out, 1)


class ResidualBlock(_Holder):
    """model/layers.py:140-181.  dim_out == dim_in: the Transformer / Compressor block (one adaLN of 6 chunks, identity
    shortcut).  dim_out != dim_in: the U-Net "down" block (score.py:77-83) — 1x1-conv shortcut, adaLN1 (shift, scale of
    width dim_in) and adaLN2 (gate_msa, shift_mlp, scale_mlp, gate_mlp of width dim_out)."""

    def __init__(self, dim_in, dim_kv, dim_c, num_heads, norm=None, mlp_ratio=4.0, dropout_att=0., dropout_mlp=0.,
                 rescale=False, dim_out=None, AdaLN=True, act=None):
        super().__init__()
        if rescale or dropout_att or dropout_mlp or (dim_c is not None and not AdaLN):
            raise NotImplementedError("ResidualBlock variant not on the shipped path")
        # act: the activation behind norm1 / norm2 in the branches WITHOUT AdaLN modulation (layers.py:224-226: a block called without a
        # condition — the decoder blocks, `decoder_act`); the AdaLN branch (:212-219) never applies it
        self.act = act
        if dim_out is not None and dim_out != dim_in:
            if dim_c is None:
                raise NotImplementedError("dim_in != dim_out without a condition is not used by the reference")
            self.shortcut = nn.Conv1d(dim_in, dim_out, 1)
        else:
            dim_out = dim_in
        self.dim_in, self.dim_out, self.dim_kv, self.dim_c, self.num_heads = dim_in, dim_out, dim_kv, dim_c, num_heads
        self.fc_q = nn.Conv1d(dim_in, dim_out, 1)
        self.fc_kv = nn.Conv1d(dim_kv, 2 * dim_out, 1)
        self.fc_o = nn.Conv1d(dim_out, dim_out, 1)
        self.norm1 = make_norm(dim_in, norm, elementwise_affine=dim_c is None)
        self.norm2 = make_norm(dim_out, norm, elementwise_affine=dim_c is None)
        if dim_c is not None:
            if dim_in == dim_out:
                self.adaLN = nn.Sequential(nn.SiLU(), nn.Linear(dim_c, 6 * dim_out))
            else:
                self.adaLN1 = nn.Sequential(nn.SiLU(), nn.Linear(dim_c, 2 * dim_in))
                self.adaLN2 = nn.Sequential(nn.SiLU(), nn.Linear(dim_c, 4 * dim_out))
        self.mlp = MLP(dim_out, int(mlp_ratio * dim_out), dim_out, 1)


class FinalLayer(_Holder):
    def __init__(self, dim_in, dim_out, dim_c, norm):
        super().__init__()
        self.norm = make_norm(dim_in, norm, elementwise_affine=dim_c is None)
        if dim_c is not None:
            self.adaLN = nn.Sequential(nn.SiLU(), nn.Linear(dim_c, 2 * dim_in))
        self.ln = nn.Conv1d(dim_in, dim_out, 1)


def conv_w(m):
    """(out,in,1) Conv1d or (out,in) Linear weight as a 2-D fp32 [N][K] matrix view."""
    w = m.weight
    return w[:, :, 0] if w.dim() == 3 else w


def params_fingerprint(module):
    """Changes whenever a parameter is re-pointed (EMA swap: p.data = ema) or written in place."""
    return tuple((p.data_ptr(), p._version) for p in module.parameters())
