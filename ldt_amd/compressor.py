"""`Compressor` — the attention-based point-cloud set-VAE (reference: model/Compressor/Network.py:104-285,
model/Compressor/layers.py), MI355X path.

Same constructor / parameter tree / methods as the reference: `forward(x)` (= encode + reconstruct,
Network.py:235-249), `sample(shape, given_eps)` (= decode, :251-268), `init()` (:163-165), plus the aliases
`encode` / `decode` that BASELINE.json's north-star names.  All arithmetic runs in libldt_hip.so.
"""
import os

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, LdtHipError
from .blocks import final_layer, pack_block, pack_final, residual_block
from .layers import ActNorm, FinalLayer, LabelEmbedding, MLP, ResidualBlock, _Holder, conv_w, params_fingerprint


# ----------------------------------------------------------------------------- parameter holders
class InitialSet(_Holder):
    """model/Compressor/layers.py:12-25: a learned (max_outputs, dim) query set, or — max_outputs None — the parameters of a
    mixture of n_mixtures Gaussians the seed rows are drawn from (same parameter names, shapes and construction order)."""

    def __init__(self, dim_seed, max_outputs, n_mixtures=4):
        super().__init__()
        self.dim_seed, self.max_outputs = dim_seed, max_outputs
        if max_outputs is None:
            import math
            self.n_mixtures = n_mixtures
            self.logits = nn.Parameter(torch.ones(n_mixtures, ))
            self.mu = nn.Parameter(torch.randn(n_mixtures, dim_seed))
            self.sig = nn.Parameter(torch.randn(n_mixtures, dim_seed).abs() / math.sqrt(n_mixtures))
            self.output = nn.Sequential(nn.Linear(dim_seed, dim_seed), nn.SiLU(), nn.Linear(dim_seed, dim_seed))
        else:
            self.prior = nn.Parameter(torch.rand((max_outputs, dim_seed), requires_grad=True))


class ConvBNReLU1D(_Holder):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.act = nn.ReLU(inplace=True)
        self.net = nn.Sequential(nn.Conv1d(in_channels, out_channels, 1), nn.BatchNorm1d(out_channels), self.act)


class ConvBNReLURes1D(_Holder):
    def __init__(self, channel):
        super().__init__()
        self.act = nn.ReLU(inplace=True)
        self.net1 = nn.Sequential(nn.Conv1d(channel, channel, 1), nn.BatchNorm1d(channel), self.act)
        self.net2 = nn.Sequential(nn.Conv1d(channel, channel, 1))


class PreExtraction(_Holder):
    def __init__(self, channels, out_channels):
        super().__init__()
        self.transfer = ConvBNReLU1D(3 + 2 * channels, out_channels)
        self.operation = nn.Sequential(ConvBNReLURes1D(out_channels))


class LocalGrouper(_Holder):
    """model/Compressor/layers.py:271-287 (use_xyz=True): 'anchor' (Compressor, cluster_norm) or 'center'
    (ConditionNet, model/scorenet/score.py:22) normalisation."""

    def __init__(self, in_channels, use_xyz=True, normalize="anchor"):
        super().__init__()
        if not use_xyz or normalize is None or normalize.lower() not in ("anchor", "center"):
            raise NotImplementedError("LocalGrouper: only use_xyz=True with normalize 'anchor' or 'center' is built")
        self.in_channels = in_channels
        self.normalize = normalize.lower()
        self.affine_alpha = nn.Parameter(torch.ones([1, 1, 1, in_channels + 3]))
        self.affine_beta = nn.Parameter(torch.zeros([1, 1, 1, in_channels + 3]))
        self.extraction = PreExtraction(in_channels, in_channels)

    def pack(self):
        """fp32 affine vectors + bf16 PreExtraction panels with eval-mode BatchNorm folded into the preceding 1x1 conv
        (Compressor/layers.py:115-160)."""
        f32 = lambda t: t.detach().float().contiguous()
        ex = self.extraction
        G = {"normalize": self.normalize, "alpha": f32(self.affine_alpha.reshape(-1)), "beta": f32(self.affine_beta.reshape(-1))}
        w1, b1 = _fold_bn(ex.transfer.net[0], ex.transfer.net[1])
        G["w_pre1"], G["b_pre1"] = _bf16_panel(w1), b1
        w2, b2 = _fold_bn(ex.operation[0].net1[0], ex.operation[0].net1[1])
        G["w_pre2"], G["b_pre2"] = _bf16_panel(w2), b2
        w3 = f32(conv_w(ex.operation[0].net2[0]))
        G["w_pre3"], G["b_pre3"] = _bf16_panel(w3), f32(ex.operation[0].net2[0].bias)
        if self.in_channels == 128 and self.normalize == "anchor" and w1.shape == (128, 259) and w2.shape == w3.shape == (128, 128):
            G["wimg"] = _grouper_fragment_image(w1, w2, w3)             # operands of the one-kernel grouper
        return G


def _grouper_fragment_image(w1, w2, w3):
    """The three PreExtraction panels as the bf16 MFMA fragments `ldt_grouper_mlp` keeps in LDS (layout: include/ldt_hip.h):
    [fragment][lane = 32 h + i][slot e] = W[32 blk + i][col(step, h, e)], zero where a layer-1 slot has no input column."""
    dev = w1.device
    h = torch.arange(2, device=dev).view(1, 2, 1)
    e = torch.arange(8, device=dev).view(1, 1, 8)
    s1 = torch.arange(17, device=dev).view(17, 1, 1)
    col1 = torch.where(s1 < 8, 16 * s1 + 8 * h + e, 131 + 16 * (s1 - 8) + 8 * h + e)
    col1 = torch.where(s1 == 16, torch.where((h == 0) & (e < 3), 128 + e, torch.full_like(col1, -1)), col1)
    s2 = torch.arange(8, device=dev).view(8, 1, 1)
    col2 = 16 * s2 + 8 * (e // 4) + 4 * h + (e % 4)

    def frags(w, col):                                                   # w [128, K], col [steps, 2, 8] (-1: no input)
        wz = torch.cat([w, torch.zeros((128, 1), device=dev, dtype=w.dtype)], 1)
        c = torch.where(col < 0, torch.full_like(col, w.shape[1]), col)
        img = wz[:, c.reshape(-1)].view(4, 32, col.shape[0], 2, 8)       # [blk, i, step, h, e]
        return img.permute(2, 0, 3, 1, 4).reshape(-1)                    # [step, blk, h, i, e]

    img = torch.cat([frags(w1, col1), frags(w2, col2), frags(w3, col2)]).to(torch.bfloat16).contiguous()
    assert img.numel() == ops.GROUPER_FRAGS * 512
    return img


FUSED_GROUPER = os.environ.get("LDT_FUSED_GROUPER", "1") != "0"          # 0: the five-kernel chain (A/B runs, parity tests)


def run_grouper(G, pts, feat, groups, k):
    """LocalGrouper.forward (Compressor/layers.py:288-319) on the HIP kernels.  pts fp32 [B,n,3], feat fp32 [B,n,D]
    -> (centres [B,S,3], tokens fp32 [B*S, D], fps_idx, knn_idx): FPS + kNN grouping, normalisation, PreExtraction
    (Conv+BN+ReLU, residual Conv+BN+ReLU / Conv, ReLU) and the max over the k neighbours."""
    from ._lib import EPI_RELU_BF16
    B = pts.shape[0]
    fps_idx = ops.fps(pts, groups)
    centers = ops.gather_rows(pts, fps_idx)                                     # [B,S,3]
    knn_idx = ops.knn(pts, centers, k)
    global FUSED_GROUPER
    if FUSED_GROUPER and "wimg" in G and (k in (8, 16) or k % 32 == 0) and feat.shape[2] == 128:
        # grouped rows, the three pointwise layers and the max over neighbours in one kernel: no [B*S*k, .] tensor exists.
        # (Its fmaxf neighbour max + ReLU turn NaN activations into finite values; the chain's max-pool would propagate them.)
        try:
            tok = ops.grouper_mlp(feat.contiguous(), pts.contiguous(), fps_idx, knn_idx, G["alpha"], G["beta"], G["wimg"],
                                  G["b_pre1"], G["b_pre2"], G["b_pre3"])
            return centers, tok, fps_idx, knn_idx
        except LdtHipError as e:                                    # e.g. hipFuncSetAttribute refusing 138 KB of dynamic LDS
            import warnings
            FUSED_GROUPER = False                                   # once per process: the five-kernel chain computes the same tokens
            warnings.warn("ldt_amd: fused grouper kernel unavailable (%s); using the group_normalize + GEMM chain" % e)
    U = ops.group_normalize(feat, pts, fps_idx, knn_idx, G["alpha"], G["beta"], normalize=G["normalize"])
    h1 = ops.gemm_bf16(U, G["w_pre1"], G["b_pre1"], EPI_RELU_BF16)              # transfer: Conv+BN+ReLU
    r = ops.gemm_bf16(h1, G["w_pre2"], G["b_pre2"], EPI_RELU_BF16)              # net1: Conv+BN+ReLU
    h2 = ops.gemm_bf16(r, G["w_pre3"], G["b_pre3"], EPI_RELU_BF16, skip=h1)     # act(net2(.) + x)
    return centers, ops.maxpool(h2, B * groups, k), fps_idx, knn_idx            # max over the k neighbours


class MiniPointnet(_Holder):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.conv1 = nn.Conv1d(input_dim, 128, 1)
        self.conv2 = nn.Conv1d(128, 256, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(256)
        self.fc = nn.Linear(256, output_dim)


class Encoder(_Holder):
    def __init__(self, dim_in, p_dim, num_heads, norm, mlp_ratio=4.0, num_layers=1):
        super().__init__()
        self.atts = nn.ModuleList([ResidualBlock(dim_in, dim_in, p_dim, num_heads, norm, mlp_ratio)
                                   for _ in range(num_layers)])
        self.conv_out = FinalLayer(dim_in, dim_in, p_dim, norm)


class DecoderBlock(_Holder):
    def __init__(self, dim_in, dim_z, num_heads, norm, mlp_ratio=4.0, min_sigma=-30., act=None, c_dim=None):
        super().__init__()
        self.min_sigma = min_sigma
        self.att = ResidualBlock(dim_in, dim_in, c_dim, num_heads, norm, mlp_ratio, act=act)
        self.prior = nn.Sequential(nn.SiLU(), nn.Conv1d(dim_in, 2 * dim_z, 1))
        self.att1 = ResidualBlock(dim_in, dim_in, c_dim, num_heads, norm, mlp_ratio, act=act)
        self.ln = nn.Conv1d(dim_z, dim_in, 1)


def _fold_bn(conv, bn):
    """Conv1d(k=1) followed by eval-mode BatchNorm1d(eps) == one affine map: W' = s W, b' = s (b - mean) + beta."""
    s = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps))
    w = conv_w(conv).detach().float() * s[:, None]
    b = (conv.bias.detach().float() - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
    return w.contiguous(), b.contiguous()


def _bf16_panel(w):
    return ops.cast_pad_bf16(w.contiguous(), ops.pad64(w.shape[1]))


# ----------------------------------------------------------------------------- Compressor
class Compressor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.input_dim = cfg.input_dim
        self.max_outputs = cfg.max_outputs
        self.n_layers = cfg.n_layers
        self.z_dim = cfg.z_dim
        self.hidden_dim = cfg.hidden_dim
        self.num_heads = cfg.num_heads
        self.norm = cfg.norm
        self.z_scales = cfg.z_scales
        self.p_dim = cfg.p_dim
        self.outsize = cfg.outsize
        self.mlp_ratio = cfg.mlp_ratio
        self.min_sigma = cfg.min_sigma
        self.encoder_layers = cfg.encoder_layers
        self.norm_input = cfg.norm_input
        self.pre_group = cfg.pre_group
        self.class_condition = cfg.class_condition
        # (cfg.AdaLN is stored and never read upstream — Network.py:133; the Encoder blocks are built with their default AdaLN=True, :148-150 —
        #  so it is accepted and ignored here too; `decoder_act` = the activation behind the norms of the decoder blocks' no-condition branch,
        #  layers.py:224-226; `ActNorm: ~` drops the conv_in ActNorm, Network.py:121-123,200-201)
        self.decoder_act = cfg.decoder_act
        self.AdaLN = cfg.AdaLN
        # dropout (Network.py:115-116, 148-152) is the identity under eval(), the only mode the encode / decode paths run in
        # (trainer/Latent_SDE_Trainer.py:144-145): accepted, and checked against self.training at call time
        self.encoder_dropout_p = float(cfg.encoder_dropout_p or 0.)
        self.decoder_dropout_p = float(cfg.decoder_dropout_p or 0.)
        if cfg.class_condition and cfg.pos_embedding == "mlp":
            raise NotImplementedError("class_condition with pos_embedding=mlp: the reference adds a (B, p_dim) label embedding to a "
                                      "(B, p_dim, tokens) position condition (Network.py:197-198), which does not broadcast")
        self.input = nn.Conv1d(self.input_dim, self.hidden_dim, 1)
        self.ActNorm = cfg.ActNorm
        if self.ActNorm is not None:
            self.conv_in = ActNorm(self.hidden_dim, self.z_scales, feature_type=cfg.ActNorm)
        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()
        self.upsample = nn.ModuleList()
        self.group = LocalGrouper(self.hidden_dim, True, normalize=cfg.cluster_norm)
        if cfg.pos_embedding == "mlp":                                  # Network.py:133-136: a per-token position condition
            self.pos_embedding = MLP(dim_in=3, dim_hidden=self.p_dim, dim_out=self.p_dim, n_hidden=1)
        else:
            self.pos_embedding = MiniPointnet(3, self.p_dim)
        if cfg.class_condition:                                         # :137-142
            self.LabelEmbedding = LabelEmbedding(cfg.num_categorys, self.p_dim, self.p_dim)
            self.label_dim = self.p_dim
        else:
            self.label_dim = None
        for i in range(self.n_layers):
            self.encoder.append(Encoder(self.hidden_dim, self.p_dim, self.num_heads, norm=self.norm,
                                        num_layers=self.encoder_layers, mlp_ratio=self.mlp_ratio))
            self.decoder.append(DecoderBlock(self.hidden_dim, cfg.z_dim, self.num_heads, norm=self.norm,
                                             mlp_ratio=self.mlp_ratio, min_sigma=self.min_sigma, act=cfg.decoder_act,
                                             c_dim=self.label_dim))
        self.output = nn.Conv1d(self.hidden_dim, 3, 1)
        self.init_set = InitialSet(self.hidden_dim, self.max_outputs)
        if cfg.pre_group:                                               # Network.py:160-161 (created last, as upstream)
            self.pre_grouper = LocalGrouper(self.hidden_dim, True, normalize=cfg.cluster_norm)
        self._pack, self._pack_key = None, None
        self.decode_chunk = 512          # samples per decode pass (activations ~7 MB/sample at 2048 points: 3.6 GB of the 288)
        # True: consume the CPU generator exactly as the reference does (B randperms per InitialSet call even when all rows
        # are kept; posterior noise drawn with CPU randn and copied over) so that seeded runs stay stream-aligned with it.
        # False (default): no idle randperms, posterior noise from the device-side Philox keyed by ONE CPU draw — the CPU
        # draws + copies are ~2x the GPU time of an encode.  Explicit `post_noise=` / `given_eps=` work in both modes.
        self.reference_rng = False

    def _presence(self, B, num_points):
        """InitialSet's row subset (Compressor/ops.py:6-14): B CPU `randperm(max_outputs) < num_points` masks, or None when
        every row is kept.  The reference draws the B permutations even then (quirk Q9) — 10 us each, which is half of a
        1024-cloud decode on this path — so that burn is only reproduced under `reference_rng`."""
        if self.max_outputs is None or (num_points == self.max_outputs and not self.reference_rng):
            return None
        presence = [torch.randperm(self.max_outputs) < num_points for _ in range(B)]
        return torch.stack(presence, 0) if num_points != self.max_outputs else None

    def _draw_seed_eps(self, B, num_points):
        """The N(0,1) draws of a mixture InitialSet (Compressor/layers.py:38): on the CPU generator like upstream under
        `reference_rng`, else from the device Philox stream keyed by one CPU draw (4 MB per cloud at 2048 points)."""
        shape = [B, num_points, self.init_set.n_mixtures, self.hidden_dim]
        if self.reference_rng:
            return torch.randn(shape).to(self._device())
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return ops.philox_normal(tuple(shape), self._device(), seed, step=1 << 20)

    def init(self):
        """Network.py:163-165 — marks ActNorm initialised (call after loading a checkpoint)."""
        if self.ActNorm is not None:
            self.conv_in.init()

    # ------------------------------------------------------------------ packing
    def _device(self):
        return self.input.weight.device

    def packed(self):
        key = params_fingerprint(self) + tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._pack is not None and key == self._pack_key:
            return self._pack
        if self._device().type != "cuda":
            raise RuntimeError("Compressor parameters are on %s: the HIP path needs them on the GPU" % self._device())
        f32 = lambda t: t.detach().float().contiguous()
        with torch.no_grad():
            P = {"dec": [], "enc": []}
            for d in self.decoder:
                # DecoderBlock.forward (Network.py:80-83): att1's K/V source is ln(eps_j) and nothing else reads it, so
                # fc_kv(ln(eps)) is ONE fp32 linear z_dim -> 2C: W = Wkv . Wln, b = Wkv . b_ln + b_kv
                wkv, wln = f32(conv_w(d.att1.fc_kv)), f32(conv_w(d.ln))
                P["dec"].append({
                    "att1": pack_block(d.att1), "att": pack_block(d.att),
                    "w_zkv": ops.sgemm(wkv, wln.t().contiguous()),
                    "b_zkv": ops.sgemm(f32(d.ln.bias).view(1, -1), wkv, f32(d.att1.fc_kv.bias)).view(-1),
                    "w_prior": f32(conv_w(d.prior[1])), "b_prior": f32(d.prior[1].bias)})
            for e in self.encoder:
                P["enc"].append({"atts": [pack_block(a) for a in e.atts], "out": pack_final(e.conv_out)})
            P["w_out"], P["b_out"] = f32(conv_w(self.output)), f32(self.output.bias)
            P["w_input"], P["b_input"] = f32(conv_w(self.input)), f32(self.input.bias)
            if self.max_outputs is None:
                i = self.init_set
                P["mix"] = {"logits": f32(i.logits), "mu": f32(i.mu), "sig": f32(i.sig), "w1": f32(i.output[0].weight), "b1": f32(i.output[0].bias),
                            "w2": f32(i.output[2].weight), "b2": f32(i.output[2].bias)}
            else:
                P["prior"] = f32(self.init_set.prior)
            P["group"] = self.group.pack()
            if self.pre_group:
                P["pre_group"] = self.pre_grouper.pack()
            pe = self.pos_embedding
            if isinstance(pe, MLP):                                      # pos_embedding: mlp — Conv1d 3 -> p, GELU, Conv1d p -> p per centre
                P["w_pm1"], P["b_pm1"] = f32(conv_w(pe.fc[0][0])), f32(pe.fc[0][0].bias)
                P["w_pm2"], P["b_pm2"] = f32(conv_w(pe.out)), f32(pe.out.bias)
            else:                                                        # MiniPointnet (Network.py:86-101), fp32
                P["w_pe1"], P["b_pe1"] = _fold_bn(pe.conv1, pe.bn1)
                P["w_pe2"], P["b_pe2"] = _fold_bn(pe.conv2, pe.bn2)
                P["w_pefc"], P["b_pefc"] = f32(pe.fc.weight), f32(pe.fc.bias)
            if self.class_condition:
                le = self.LabelEmbedding
                P["label"] = {"emb": f32(le.label_emb.weight), "w1": f32(le.mlp[0].weight), "b1": f32(le.mlp[0].bias),
                              "w2": f32(le.mlp[2].weight), "b2": f32(le.mlp[2].bias)}
            if self.ActNorm is not None:
                P["an_shift"], P["an_logs"] = f32(self.conv_in.shift.reshape(-1)), f32(self.conv_in.log_scale.reshape(-1))
        self._pack, self._pack_key = P, key
        return P

    # ------------------------------------------------------------------ decode (Network.py:251-268)
    @torch.no_grad()
    def _no_training_dropout(self, what):
        if self.training and (self.encoder_dropout_p > 0 or self.decoder_dropout_p > 0):
            raise RuntimeError("Compressor.%s: dropout configured (%g / %g) and the module is in training mode — the HIP path is the "
                               "inference path, call eval()" % (what, self.encoder_dropout_p, self.decoder_dropout_p))

    def sample(self, shape, given_eps=None, keep_mask=None, seed_eps=None):
        """Top-down generation: given_eps (B, tokens, n_layers*z_dim) -> points (B, N, 3).  seed_eps: with a mixture InitialSet
        (max_outputs None), the (B, N, n_mixtures, hidden) draws of its seed rows (parity runs)."""
        self._no_training_dropout("sample")
        B, num_points = shape[0], shape[1]
        num_points = self.outsize if num_points is None else num_points
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("Compressor.sample: parameters on %s; the HIP path has no CPU fallback" % dev)
        if keep_mask is None:
            keep_mask = self._presence(B, num_points)
        if given_eps is None:                                            # Network.py:259-260
            given_eps = torch.randn((B, self.z_scales, self.n_layers * self.z_dim)).to(dev)
        eps = given_eps.to(dev, torch.float32).contiguous()
        P = self.packed()
        out = torch.empty((B, num_points, 3), dtype=torch.float32, device=dev)
        for b0 in range(0, B, self.decode_chunk):
            b1 = min(B, b0 + self.decode_chunk)
            km = None if keep_mask is None else keep_mask[b0:b1]
            se = None if seed_eps is None else seed_eps[b0:b1]
            out[b0:b1] = self._decode_chunk(P, eps[b0:b1], num_points, km, se)
        return self.postprocess(out)

    def _initial_set(self, P, Bc, num_points, keep_mask, seed_eps=None):
        """Compressor/layers.py:26-42: the learned prior rows, token-major fp32 [Bc*N, C]; or (max_outputs None) rows drawn from
        the learned mixture — `seed_eps` (Bc, N, n_mixtures, C) replaces the N(0,1) draw of :38 — through the `output` MLP."""
        if self.max_outputs is None:
            M = P["mix"]
            dev = M["mu"].device
            if seed_eps is None:
                seed_eps = self._draw_seed_eps(Bc, num_points)
            e = seed_eps.to(dev, torch.float32).contiguous().view(Bc * num_points, self.init_set.n_mixtures, self.hidden_dim)
            x = ops.mixture_seed(e, M["sig"], M["mu"], M["logits"])
            h = ops.sgemm(x, M["w1"], M["b1"], act_out=ACT_SILU)
            return ops.sgemm(h, M["w2"], M["b2"])
        prior = P["prior"]
        if keep_mask is None:
            return prior.unsqueeze(0).expand(Bc, -1, -1).reshape(Bc * num_points, -1).contiguous()
        km = keep_mask.to(prior.device)
        return torch.stack([prior[km[b]] for b in range(Bc)], 0).reshape(Bc * num_points, -1).contiguous()

    def _decoder_level(self, Pd, o, eps_j, Bc, N, T, c=None, o_bf16=None, q_pre=None, next_P=None):
        """DecoderBlock.forward (Network.py:80-83): o <- att1(o, ln(eps_j), c); c = label embedding in a class-conditional
        forward, None in `sample` (which never passes one, :263-264: the block then runs its plain-LayerNorm branch)."""
        kv = ops.sgemm(eps_j, Pd["w_zkv"], Pd["b_zkv"], out_bf16=True)           # fc_kv(Conv1d z_dim -> C) on T tokens, composed
        return residual_block(Pd["att1"], o, Bc, N, kv_pre=kv, Nk=T, c=c, x_bf16_out=o_bf16, q_pre=q_pre, next_P=next_P)

    def _decode_chunk(self, P, eps, N, keep_mask, seed_eps=None):
        Bc, T, _ = eps.shape
        o = self._initial_set(P, Bc, N, keep_mask, seed_eps)
        e2 = eps.view(Bc * T, -1)
        q = None
        for j in range(self.n_layers):                                             # reversed(self.decoder), :263
            Pd = P["dec"][self.n_layers - 1 - j]
            # level j's last kernel also computes level j + 1's LN1 + fc_q on the rows it has just written (no condition here)
            nxt = P["dec"][self.n_layers - 2 - j]["att1"] if j + 1 < self.n_layers else None
            r = self._decoder_level(Pd, o, e2[:, self.z_dim * j: self.z_dim * (j + 1)], Bc, N, T, q_pre=q, next_P=nxt)
            q = r[1] if nxt is not None else None
        return ops.sgemm(o, P["w_out"], P["b_out"]).view(Bc, N, 3)                 # Conv1d C -> 3, :266

    @staticmethod
    def postprocess(x):
        if x.shape[-1] != 3:
            raise NotImplementedError("only xyz outputs (ShapeNet) are on the shipped path")   # Network.py:271-279
        return x

    # aliases named by BASELINE.json's north-star
    def decode(self, given_eps, num_points=None):
        return self.sample((given_eps.shape[0], num_points), given_eps=given_eps)

    def encode(self, x, **kw):
        return self.forward(x, **kw)["all_eps"]

    # ------------------------------------------------------------------ encode (Network.py:188-249)
    @torch.no_grad()
    def forward(self, x, num_points=None, label=None, *, post_noise=None, want_stats=False, seed_eps=None):
        """Bidirectional inference: x (B, N, 3) -> dict with 'all_eps' (B, tokens, n_layers*z_dim) and the
        reconstruction 'set' (B, N, 3).  `post_noise`: optional list of n_layers tensors (B, tokens, z_dim) replacing
        the N(0,1) draws of `sample(mu, logvar)` (Network.py:26-29).  Training-only entries of the reference dict
        ('kls', 'all_logqz') are not produced; 'posteriors' holds token-major (mu, logvar) when want_stats."""
        self._no_training_dropout("forward")
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("Compressor.forward: parameters on %s; the HIP path has no CPU fallback" % dev)
        B, N, _ = x.shape
        T, D, z, L = self.z_scales, self.hidden_dim, self.z_dim, self.n_layers
        npts = self.outsize if num_points is None else num_points
        # reference order on the CPU generator: B randperms (InitialSet, :215) then one randn per level (:220)
        keep_mask = self._presence(B, npts)
        if self.max_outputs is None and seed_eps is None:                 # (the mixture's seed rows are drawn first, :215 before :220)
            seed_eps = self._draw_seed_eps(B, npts)
        if post_noise is None:
            if self.reference_rng:
                post_noise = [torch.randn((B, z, T)).transpose(1, 2) for _ in range(L)]
            else:                                                    # one CPU draw keys a device-side Philox stream per level
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                post_noise = [ops.philox_normal((B, T, z), dev, seed, step=j) for j in range(L)]
        P = self.packed()
        pts = x.to(dev, torch.float32).contiguous()
        if self.norm_input:
            pts = ops.norm_points(pts)                                              # norm_pts (:170-174, :189-190)
        # ---- bottom_up: input conv, FPS + kNN grouping, PreExtraction, pos embedding, ActNorm, encoder stages
        feat = ops.sgemm(pts.view(B * N, 3), P["w_input"], P["b_input"])            # Conv1d 3 -> D  (:192)
        n_cur = N
        if self.pre_group:                                                          # :193-194: 256 groups of 32 neighbours first
            pts, feat, _, _ = run_grouper(P["pre_group"], pts, feat.view(B, N, D), 256, 32)
            n_cur = 256
        k = n_cur // T * 2                                                          # :195
        centers, tok, fps_idx, knn_idx = run_grouper(P["group"], pts, feat.view(B, n_cur, D), T, k)
        tok_pre = tok.clone() if want_stats else None
        per_tok = "w_pm1" in P
        if per_tok:                                                                 # pos_embedding: mlp (:133-134): one row per token
            from ._lib import ACT_GELU
            pos = ops.sgemm(ops.sgemm(centers.view(B * T, 3), P["w_pm1"], P["b_pm1"], act_out=ACT_GELU), P["w_pm2"], P["b_pm2"])
        else:
            c1 = ops.sgemm(centers.view(B * T, 3), P["w_pe1"], P["b_pe1"], act_out=ACT_RELU)
            c2 = ops.sgemm(c1, P["w_pe2"], P["b_pe2"], act_out=ACT_RELU)
            pos = ops.sgemm(ops.maxpool(c2, B, T), P["w_pefc"], P["b_pefc"])        # [B, p_dim]
        l_emb = None
        if label is not None and self.class_condition:                              # :240-241, :197-198 (ignored otherwise, as upstream)
            PL = P["label"]
            e = PL["emb"][label.to(dev).long()].contiguous()
            l_emb = ops.sgemm(ops.sgemm(e, PL["w1"], PL["b1"], act_out=ACT_SILU), PL["w2"], PL["b2"])
            pos = pos + l_emb
        if self.ActNorm is not None:
            ops.actnorm_(tok, P["an_shift"], P["an_logs"], B)
        enc_out = []
        for Pe in P["enc"]:                                                         # Network.py:203-205, 41-45
            for Pa in Pe["atts"]:
                residual_block(Pa, tok, B, T, y_bf16=ops.cast_pad_bf16(tok, ops.pad64(D)), Nk=T, c=pos, per_token=per_tok)
            enc_out.append(final_layer(Pe["out"], tok, B, T, pos, per_token=per_tok))
        # ---- top_down: posterior per level + decoder block
        o = self._initial_set(P, B, npts, keep_mask, seed_eps)
        all_eps = torch.empty((B * T, L * z), dtype=torch.float32, device=dev)
        stats = []
        q_dec = None                                                                # next decoder level's query projection (fused)
        # bf16 image of o for the next level's att(x, o): written by the decoder block's last kernel, not by a pass of its own
        o_bf = torch.empty((B * npts, D), dtype=torch.bfloat16, device=dev) if (L > 1 and D % 64 == 0) else None
        for j in range(L):
            Pd = P["dec"][L - 1 - j]
            xj = enc_out[-j - 1]                                                    # (each stage output is consumed once: updated in place)
            if j == 0:
                y, nk = ops.cast_pad_bf16(xj, ops.pad64(D)), T                       # compute_posterior(x, None): att(x, x)
            else:                                                                   # att(x, o): K/V = 2048 decoded points
                y, nk = (o_bf if o_bf is not None else ops.cast_pad_bf16(o, ops.pad64(D))), npts
            residual_block(Pd["att"], xj, B, T, y_bf16=y, Nk=nk, c=l_emb)
            post = ops.sgemm(xj, Pd["w_prior"], Pd["b_prior"], act_in=ACT_SILU)      # SiLU -> Conv1d D -> 2z
            nz = post_noise[j].to(dev, torch.float32).contiguous().view(B * T, z)
            ej = all_eps[:, z * j: z * (j + 1)]
            stats.append(ops.reparam(post, nz, ej, self.min_sigma, 10., want_stats))
            nxt = P["dec"][L - 2 - j]["att1"] if (j + 1 < L and l_emb is None) else None
            r = self._decoder_level(Pd, o, ej, B, npts, T, c=l_emb, o_bf16=o_bf if j + 1 < L else None, q_pre=q_dec, next_P=nxt)
            q_dec = r[1] if nxt is not None else None
        out = ops.sgemm(o, P["w_out"], P["b_out"]).view(B, npts, 3)
        res = {"set": self.postprocess(out), "all_eps": all_eps.view(B, T, L * z), "max": tok.max(),
               "posteriors": [(all_eps.view(B, T, L * z)[..., z * j: z * (j + 1)],) + tuple(
                   None if t is None else t.view(B, T, z) for t in stats[j]) for j in range(L)],
               "kls": None, "all_logqz": None,
               "fps_idx": fps_idx, "knn_idx": knn_idx, "centers": centers, "tokens": tok_pre}
        return res
