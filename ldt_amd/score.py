"""`Score` — the eps-prediction Transformer (reference: model/scorenet/score.py:47-151), MI355X path.

Same constructor, parameter tree and `forward(x, t, label=None, condition=None)` contract as the
reference class; the arithmetic is one `ldt_score_forward` call into libldt_hip.so (24 x [LN+AdaLN ->
QKV GEMM -> fused attention -> out-proj+gate+residual -> LN+AdaLN -> MLP-up+GELU -> MLP-down+gate+residual]).

Host-side responsibilities kept in Python:
  * weight packing: Conv1d (out,in,1) fp32 -> bf16 [N][K] panels, fc_q|fc_kv concatenated to one
    [3C][C] operand (self-attention: both read the same modulated input, layers.py:184-189); repacked
    when the parameters change (EMA swap, tools/utils.py:80-101).
  * AdaLN tables: c = TimeEmbedding(t) and every block's adaLN Linear(SiLU(c)) (layers.py:172,214,
    238,244) are evaluated in fp32 for a whole vector of times at once (`time_table`).  In unconditional
    sampling t is one scalar per step (diffusion_continuous.py:243-244), so the sampler builds the table
    for all N steps in one batched call instead of 25 GEMVs per step (SURVEY hard part 3).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_NONE, ACT_SILU, MAX_BLOCKS, ScorePlan, check, lib
from .layers import FinalLayer, LabelEmbedding, ResidualBlock, TimeEmbedding, conv_w, params_fingerprint


class Score(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.z_dim = cfg.z_dim
        self.out_dim = self.z_dim
        self.z_scale = cfg.z_scale
        self.hidden_size = cfg.hidden_size
        self.num_heads = cfg.num_heads
        self.condition = cfg.condition
        self.num_steps = cfg.num_steps
        self.norm = cfg.norm
        self.t_dim = cfg.t_dim
        self.num_blocks = cfg.num_blocks
        self.unet = cfg.unet
        self.AdaLN = cfg.AdaLN
        # norm other than layer_norm (tools/utils.py:168-181: group_norm, None): the blocks run host-driven over the HIP kernels like the
        # U-Net variant (the fused C++ forward and its LN folding are LayerNorm code); `batch_norm` is refused by make_norm with the reason
        self.host_blocks = bool(self.unet) or not (isinstance(self.norm, str) and self.norm.lower() == "layer_norm")
        if self.condition:
            from .condition import ConditionNet                     # built first, as upstream (score.py:64-65)
            self.c_net = ConditionNet(self.hidden_size, self.t_dim, patch_size=self.z_scale)
        # dropout (score.py:61; layers.py:179,199,129) is the identity under eval(), the only mode the sampling path runs in
        # (Latent_SDE_Trainer.py:144): it is accepted and checked against self.training in forward
        self.dropout = float(getattr(cfg, "dropout", 0.) or 0.)
        if not self.AdaLN:
            raise NotImplementedError("AdaLN: False blocks (layers.py:221-223) are not built: upstream adds pos_embedding(c) of shape (B, 1, C) to "
                                      "the channel-first (B, C, N) activations without the transpose its AdaLN branch applies, which only "
                                      "broadcasts when tokens == hidden_size — no shipped YAML sets it")
        if self.num_blocks > MAX_BLOCKS:
            raise ValueError("num_blocks > %d" % MAX_BLOCKS)
        D = self.hidden_size
        mk = lambda din, dout=None: ResidualBlock(din, din, self.t_dim, self.num_heads, norm=self.norm, dim_out=dout,
                                                  act=cfg.act, AdaLN=self.AdaLN)
        if self.unet:                                               # score.py:67-83: up / mid / down with skip concats
            self.Transformer_Up = nn.ModuleList([mk(D) for _ in range(self.num_blocks // 2)])
            self.Transformer_Mid = mk(D)
            self.Transformer_Down = nn.ModuleList([mk(2 * D, D) for _ in range(self.num_blocks // 2)])
        else:
            self.Transformer = nn.ModuleList([mk(D) for _ in range(self.num_blocks)])
        if cfg.num_categorys > 1:
            self.LabelEmbedding = LabelEmbedding(cfg.num_categorys, self.t_dim, self.t_dim)
        else:
            self.label_dim = None
        self.ln_in = nn.Conv1d(self.z_dim, self.hidden_size, 1)
        self.TimeEmbedding = TimeEmbedding(self.t_dim // 4, self.t_dim)
        self.ln_out = FinalLayer(self.hidden_size, self.z_dim, self.t_dim, self.norm)
        self._pack = None
        self._pack_key = None
        self._ws = {}
        self._freq = None
        self._cond_cache = {}
        self._fold_disabled = False      # set by fold_probe when a row's mean^2 / variance exceeds FOLD_MAX_MEAN_RATIO
        self.fold_ratio_seen = 0.0       # largest mean^2 / variance the probes have measured (diagnostic)
        self._fold_pending = None        # device scalar the last sampling loop's in-loop monitor wrote; read by collect_fold_ratio()

    # ------------------------------------------------------------------ packed weights
    @property
    def n_mod(self):
        return self.num_blocks * 6 * self.hidden_size + 2 * self.hidden_size

    def _device(self):
        return self.ln_in.weight.device

    def packed(self):
        """bf16 operand panels + fp32 biases, rebuilt when any parameter changed (EMA swap => repack)."""
        key = params_fingerprint(self)
        if self._pack is not None and key == self._pack_key:
            return self._pack
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("Score parameters are on %s: the HIP path needs them on the GPU (.to('cuda'))" % dev)
        if self.unet:
            from .blocks import pack_block, pack_final
            with torch.no_grad():
                zp = ops.pad64(self.z_dim)
                P = {"w_in": ops.cast_pad_bf16(conv_w(self.ln_in).float().contiguous(), zp),
                     "b_in": self.ln_in.bias.detach().float().contiguous(),
                     "up": [pack_block(b) for b in self.Transformer_Up], "mid": pack_block(self.Transformer_Mid),
                     "down": [pack_block(b) for b in self.Transformer_Down], "final": pack_final(self.ln_out)}
            self._pack, self._pack_key = P, key
            return P
        if self.host_blocks:
            from .blocks import pack_block, pack_final
            with torch.no_grad():
                P = {"w_in": ops.cast_pad_bf16(conv_w(self.ln_in).float().contiguous(), ops.pad64(self.z_dim)),
                     "b_in": self.ln_in.bias.detach().float().contiguous(),
                     "blocks": [pack_block(b) for b in self.Transformer], "final": pack_final(self.ln_out)}
            self._pack, self._pack_key = P, key
            return P
        with torch.no_grad():
            P = {"w_qkv": [], "b_qkv": [], "w_o": [], "b_o": [], "w_up": [], "b_up": [], "w_dn": [], "b_dn": []}
            zp = ops.pad64(self.z_dim)
            P["w_in"] = ops.cast_pad_bf16(conv_w(self.ln_in).float().contiguous(), zp)
            P["b_in"] = self.ln_in.bias.detach().float().contiguous()
            for blk in self.Transformer:
                wqkv = torch.cat([conv_w(blk.fc_q), conv_w(blk.fc_kv)], 0).float().contiguous()
                P["w_qkv"].append(ops.cast_pad_bf16(wqkv, wqkv.shape[1]))
                P["b_qkv"].append(torch.cat([blk.fc_q.bias, blk.fc_kv.bias]).detach().float().contiguous())
                for nm, m in (("o", blk.fc_o), ("up", blk.mlp.fc[0][0]), ("dn", blk.mlp.out)):
                    w = conv_w(m).float().contiguous()
                    P["w_" + nm].append(ops.cast_pad_bf16(w, w.shape[1]))
                    P["b_" + nm].append(m.bias.detach().float().contiguous())
            w = conv_w(self.ln_out.ln).float().contiguous()
            P["w_out"] = ops.cast_pad_bf16(w, w.shape[1])
            P["b_out"] = self.ln_out.ln.bias.detach().float().contiguous()
        self._pack, self._pack_key = P, key
        self._cond_cache = {}
        return P

    def stacked_adaln(self):
        """Every adaLN Linear stacked in mod-row order ([n_mod][t_dim] fp32 + [n_mod]) for the one-launch per-step
        AdaLN of the conditional sampler (ldt_cond_args.w_ada); built on first use, dropped on repack."""
        P = self.packed()
        if "w_ada" not in P:
            with torch.no_grad():
                lins = [blk.adaLN[1] for blk in self.Transformer] + [self.ln_out.adaLN[1]]
                P["w_ada"] = torch.cat([l.weight.detach().float() for l in lins], 0).contiguous()
                P["b_ada"] = torch.cat([l.bias.detach().float() for l in lins], 0).contiguous()
        return P["w_ada"], P["b_ada"]

    def stacked_adaln_bf16(self):
        """The stacked adaLN weights as a bf16 panel [n_mod][t_dim] (ldt_cond_args.w_ada_bf16): the per-step per-sample AdaLN rows of
        the conditional sampler are HBM-bound on their weights (BASELINE configs[4]: 604 MB fp32 per step, 163 us = 9 % of a step;
        80 us as bf16 — profiles/r05_adaln_bf16_probe.txt), and bf16 operands with fp32 accumulation are what every token GEMM of the
        model already runs on.  LDT_ADALN_BF16=0 keeps the fp32 SGEMM (the rows then equal the reference's fp32 Linear to 1e-12).
        -> (panel, bias), or None when the MFMA row GEMM does not take this width (t_dim % 64 != 0: the fp32 SGEMM takes any).  The fp32
        stack is dropped once the panel exists (the loop needs one of the two, include/ldt_hip.h ldt_cond_args)."""
        if self.t_dim % 64 != 0:
            return None
        P = self.packed()
        if "w_ada_bf16" not in P:
            w_ada, b_ada = self.stacked_adaln()
            P["w_ada_bf16"] = ops.cast_pad_bf16(w_ada, w_ada.shape[1])
            P["b_ada_only"] = b_ada
            P.pop("w_ada", None)                                         # 604 MB at the shipped width; rebuilt on demand by stacked_adaln()
        return P["w_ada_bf16"], P["b_ada_only"]

    def time_embedding(self, t):
        """TimeEmbedding(t) only: t [n] -> c [n, t_dim] fp32 (model/layers.py:38-41)."""
        te = self.TimeEmbedding.mlp
        e = ops.sinusoid(t.contiguous(), self._frequencies())
        h = ops.sgemm(e, te[0].weight, te[0].bias, act_out=ACT_SILU)
        return ops.sgemm(h, te[2].weight, te[2].bias)

    def condition_embedding(self, label=None, condition=None):
        """-> (extra [B,t_dim] or None, kv_cond {block: K|V} or None, cond_tokens) for forward()/the fused loop."""
        if isinstance(condition, dict):                              # score.py:129-131: raw ViPC inputs -> ConditionNet
            if not hasattr(self, "c_net"):
                raise ValueError("a raw condition dict needs cfg.score.condition=True (ConditionNet, score.py:64-65)")
            condition = self.c_net(condition)
        pts_cond, img_cond = (None, 0.) if condition is None else condition
        extra = None
        if label is not None:
            extra = self.label_embedding(label)                      # a label wins over the image condition (score.py:135)
        elif torch.is_tensor(img_cond):
            extra = img_cond.to(self._device(), torch.float32).contiguous()
        kv, S = (None, 0)
        if torch.is_tensor(pts_cond):
            kv, S = self.project_condition(pts_cond)
        return extra, kv, S

    def _cross_panels(self, l):
        """bf16 fc_q and fc_kv panels of block l as separate operands (cross-attention: layers.py:186-189)."""
        P = self.packed()
        if ("w_q", l) not in P:
            blk = self.Transformer[l]
            with torch.no_grad():
                wq = conv_w(blk.fc_q).float().contiguous(); wkv = conv_w(blk.fc_kv).float().contiguous()
                P[("w_q", l)] = ops.cast_pad_bf16(wq, wq.shape[1]); P[("b_q", l)] = blk.fc_q.bias.detach().float().contiguous()
                P[("w_kv", l)] = ops.cast_pad_bf16(wkv, wkv.shape[1]); P[("b_kv", l)] = blk.fc_kv.bias.detach().float().contiguous()
        return P[("w_q", l)], P[("b_q", l)], P[("w_kv", l)], P[("b_kv", l)]

    def project_condition(self, pts_cond):
        """pts_cond (B, hidden, S) channels-first (what ConditionNet returns, score.py:41-44) -> per even block the
        K|V rows [B*S, 2*hidden] bf16.  Step-invariant: computed once and cached per condition tensor.  The cache entry
        holds a reference to that tensor, so its storage cannot be freed and handed to the next batch's condition (same
        address, same version) while the entry is alive."""
        key = (pts_cond.data_ptr(), pts_cond._version, tuple(pts_cond.shape))
        self.packed()
        if key not in self._cond_cache:
            from ._lib import EPI_BF16
            B, C, S = pts_cond.shape
            y = pts_cond.to(self._device(), torch.float32).transpose(1, 2).contiguous().view(B * S, C)   # token-major raw y
            yb = ops.cast_pad_bf16(y, C)
            kv = {}
            for l in range(0, self.num_blocks, 2):                                  # score.py:149: idx % 2 == 0
                _, _, wkv, bkv = self._cross_panels(l)
                kv[l] = ops.gemm_bf16(yb, wkv, bkv, EPI_BF16)
            self._cond_cache = {key: (kv, S, pts_cond)}
        return self._cond_cache[key][:2]

    def _workspace(self, B, T, slot=0):
        """Activation buffers of a (B, T) batch; `slot` separates the sub-batches that run concurrently on their own streams."""
        k = (B, T, self._device(), slot)
        if k not in self._ws:
            dev, M, D = self._device(), B * T, self.hidden_size
            bf = dict(dtype=torch.bfloat16, device=dev)
            self._ws = {kk: v for kk, v in self._ws.items() if kk[:3] == k[:3]}     # keep only this shape's slots
            self._ws[k] = {
                "xin": torch.zeros((M, ops.pad64(self.z_dim)), **bf),
                "X": torch.empty((M, D), dtype=torch.float32, device=dev),
                "Hb": torch.empty((M, D), **bf), "QKV": torch.empty((M, 3 * D), **bf),
                "Ob": torch.empty((M, D), **bf), "U": torch.empty((M, self.Transformer[0].mlp.out.in_channels), **bf),
                "stats": torch.empty((max(D // 32, 1), M, 2), dtype=torch.float32, device=dev),   # [D/256] (256-tile kernels) or [D/32] (small-batch kernels) partials per row
            }
        return self._ws[k]

    def plan(self, B, T, mod, mod_step_stride, mod_sample_stride, kv_cond=None, cond_tokens=0, fold=None, slot=0, gemm_wgs=0, monitor=None,
             monitor_every=0):
        """ctypes `ldt_score_plan` for a (B,T) batch reading AdaLN rows from `mod`; kv_cond: {block: K|V rows};
        fold: the `fold_table(mod)` of a batch-shared `mod` (enables the LN-folded GEMM epilogues); slot / gemm_wgs: workspace
        set and persistent-grid cap of a sub-batch that shares the GPU with another stream (diffusion.py, `streams`)."""
        P, W = self.packed(), self._workspace(B, T, slot)
        p = ScorePlan()
        p.gemm_wgs = gemm_wgs
        p.hidden, p.heads, p.blocks = self.hidden_size, self.num_heads, self.num_blocks
        p.z_dim, p.z_pad, p.mlp_hidden = self.z_dim, ops.pad64(self.z_dim), W["U"].shape[1]
        p.tokens, p.batch = T, B
        p.w_in, p.b_in = P["w_in"].data_ptr(), P["b_in"].data_ptr()
        for l in range(self.num_blocks):
            for nm in ("qkv", "o", "up", "dn"):
                getattr(p, "w_" + nm)[l] = P["w_" + nm][l].data_ptr()
                getattr(p, "b_" + nm)[l] = P["b_" + nm][l].data_ptr()
        p.w_out, p.b_out = P["w_out"].data_ptr(), P["b_out"].data_ptr()
        p.mod, p.mod_step_stride, p.mod_sample_stride = mod.data_ptr(), mod_step_stride, mod_sample_stride
        for nm in ("xin", "X", "Hb", "QKV", "Ob", "U"):
            setattr(p, nm, W[nm].data_ptr())
        if kv_cond:
            p.cond_tokens = cond_tokens
            for l, kv in kv_cond.items():
                wq, bq, _, _ = self._cross_panels(l)
                p.w_q[l], p.b_q[l], p.kv_cond[l] = wq.data_ptr(), bq.data_ptr(), kv.data_ptr()
        if fold is not None:
            if mod_sample_stride != 0:
                raise ValueError("LN folding needs batch-shared modulation (mod_sample_stride == 0)")
            p.fold, p.fold_step_stride, p.stats = fold.data_ptr(), fold.stride(0), W["stats"].data_ptr()
            if monitor is not None:
                p.fold_monitor = monitor.data_ptr()
                p.fold_monitor_every = int(monitor_every)               # ldt_sample_loop: monitored steps (0 = every step)
        p._keep = (P, W, mod, kv_cond, fold, monitor)   # keep the buffers alive as long as the plan
        return p

    # ------------------------------------------------------------------ AdaLN tables
    def _frequencies(self):
        """model/layers.py:29-30 evaluated with the reference's own fp32 ops (quirk Q5), cached on device."""
        if self._freq is None or self._freq.device != self._device():
            half = self.TimeEmbedding.t_emb_dim // 2
            s = np.log(10000) / (half - 1)
            self._freq = torch.exp(torch.arange(half) * -s).to(self._device())
        return self._freq

    def time_table(self, t, extra_c=None):
        """t [n] fp32 on device -> (c [n,t_dim], mod [n,n_mod]) in fp32.
        c = TimeEmbedding(t) (+ extra_c: label / image-condition embedding, score.py:135);
        mod[:, l*6D:(l+1)*6D] = adaLN_l(SiLU(c)); the last 2D columns are FinalLayer's (shift, scale)."""
        te = self.TimeEmbedding.mlp
        e = ops.sinusoid(t.contiguous(), self._frequencies())
        h = ops.sgemm(e, te[0].weight, te[0].bias, act_out=ACT_SILU)
        c = ops.sgemm(h, te[2].weight, te[2].bias)
        if extra_c is not None:
            c = ops.add_f32(c, extra_c.expand_as(c))
        D = self.hidden_size
        mod = torch.empty((t.numel(), self.n_mod), dtype=torch.float32, device=t.device)
        for l, blk in enumerate(self.Transformer):
            lin = blk.adaLN[1]
            ops.sgemm(c, lin.weight, lin.bias, act_in=ACT_SILU, out=mod[:, l * 6 * D:(l + 1) * 6 * D])
        lin = self.ln_out.adaLN[1]
        ops.sgemm(c, lin.weight, lin.bias, act_in=ACT_SILU, out=mod[:, self.num_blocks * 6 * D:])
        return c, mod

    def can_fold(self, B, T, gemm_wgs=0):
        """LN folding pays (a) when every GEMM of the block runs whole 256x256 tiles and the residual GEMMs' tiles fill at least 5/8 of the
        workgroups they may use (all 256 CUs, or a sub-batch stream's share `gemm_wgs`) — the rule by which ldt_gemm_launch prefers the 256^2
        kernel; at M = 8192 on the whole chip the smaller tiles + LayerNorm launches measured 3 % faster — and (b) for small batches (the shipped
        32-token config: M = 2048) when the mid-size tile kernels (csrc/gemm_mid.hip) take all four GEMMs of the block in their folded forms:
        there the loader waves form the row statistics while the ring fills (98.5 -> 92.5 us per block, profiles/r04_t32_kernel_sequence*.txt).
        The C++ forward makes the same decision (`ldt_score_lnfold_route`); this method asks it.
        LDT_LN_FOLD=0 disables folding, =2 forces it wherever a route exists; LDT_LN_FOLD_SMALL=0 disables (b) only (A/B runs)."""
        self.collect_fold_ratio()
        mode = int(os.environ.get("LDT_LN_FOLD", "1"))
        D, M = self.hidden_size, B * T
        F = self.Transformer[0].mlp.out.in_channels if not self.unet else 0
        if (self._fold_disabled and mode != 2) or mode == 0 or self.host_blocks:
            return False
        route = int(lib().ldt_score_lnfold_route(M, D, F, gemm_wgs))
        if route == 2:
            return mode == 2 or bool(int(os.environ.get("LDT_LN_FOLD_SMALL", "1")))
        if route == 1:
            return True
        # (mode 2 used to force the 256-tile route below the 5/8 rule: still possible when M is a multiple of 256 and no small route exists)
        return mode == 2 and M % 256 == 0 and D % 256 == 0 and D <= 1024 and F % 256 == 0

    # The folded projections round x (1 + scale) to bf16 BEFORE the row mean is removed: their operand-rounding error is
    # (1 + mean^2 / variance) x the LayerNorm kernel's (tests/test_gpu_kernels.py::test_gemm_lnfold_error_law_vs_row_mean:
    # 1.3e-6 relative MSE per projection at mean = 0, 2.2e-5 at |mean| = 4 std, 8.5e-5 at 8 std).  Past this bound
    # (|mean| > 4 std on any row of any block) the sampler uses the LayerNorm kernels, which do not have the term.
    FOLD_MAX_MEAN_RATIO = 16.0

    def fold_probe(self, x, step_index, mod, fold):
        """One monitored Score forward on latents `x` (B, T, z) at row `step_index` of the batch-shared tables: returns the
        largest mean^2 / variance over every row of every folded LayerNorm input (ldt_score_plan.fold_monitor) and disables
        folding for this model when it exceeds FOLD_MAX_MEAN_RATIO (LDT_LN_FOLD=2 keeps it forced on)."""
        B, T, _ = x.shape
        mon = torch.zeros(1, dtype=torch.float32, device=x.device)
        step = torch.full((1,), int(step_index), dtype=torch.int32, device=x.device)
        plan = self.plan(B, T, mod, self.n_mod, 0, fold=fold, slot=0, monitor=mon)   # (runs before the loop on the same stream: shares its workspace)
        out = torch.empty_like(x)
        check(lib().ldt_score_forward(ctypes.byref(plan), x.contiguous().data_ptr(), out.data_ptr(), step.data_ptr(), ops.stream_ptr()),
              "ldt_score_forward")
        return self.note_fold_ratio(float(mon.item()))

    def note_fold_ratio(self, ratio):
        """Record a measured max mean^2 / variance of the folded LayerNorm inputs (a probe forward, or the running maximum the sampling
        loop's in-loop monitor wrote) and switch this model to the LayerNorm kernels when it exceeds FOLD_MAX_MEAN_RATIO."""
        self.fold_ratio_seen = max(self.fold_ratio_seen, ratio)
        if ratio > self.FOLD_MAX_MEAN_RATIO and not self._fold_disabled:
            import warnings
            self._fold_disabled = True
            warnings.warn("ldt_amd.Score: a LayerNorm input row has mean^2 / variance = %.1f (> %.0f): the LN-folded GEMM epilogues "
                          "would lose accuracy there; using the LayerNorm kernels from now on (LDT_LN_FOLD=2 forces folding)"
                          % (ratio, self.FOLD_MAX_MEAN_RATIO))
        return ratio

    def defer_fold_ratio(self, monitor):
        """The sampling loop hands over the device scalar its in-loop monitor accumulates into INSTEAD of reading it: a read is a device
        synchronisation, and at the end of `sample()` it would serialise the next call's host work (the CPU draw of x0: 100-150 ms for a
        512-shape global batch, tools/dbg/x0_draw_cost.py) behind the ~0.1 s of steps still queued on the GPU.  The value is collected at the
        next decision that depends on it (can_fold) — i.e. after the next call's x0 has been drawn."""
        self.collect_fold_ratio()
        self._fold_pending = monitor

    def collect_fold_ratio(self):
        """Read (synchronising) the pending in-loop monitor value, if any; -> the largest ratio seen so far."""
        if self._fold_pending is not None:
            mon, self._fold_pending = self._fold_pending, None
            self.note_fold_ratio(float(mon.item()))
        return self.fold_ratio_seen

    def fold_table(self, mod):
        """Per-step S / C vectors of the LN-folded projections (include/ldt_hip.h, ldt_gemm_resid_lnstats), fp32
        [n, blocks * (6D + 2F)], block l = [S_qkv 3D | C_qkv 3D | S_up F | C_up F]:
            S[n] = sum_k (1 + scale[k]) W[n][k],   C[n] = sum_k shift[k] W[n][k] + b[n]
        with (shift, scale) the rows of `mod` feeding that block's LayerNorm (layers.py:214,218-219) and W the bf16
        panel the GEMM multiplies by (widened to fp32), so the mean term cancels against what the MFMAs accumulate."""
        P = self.packed()
        D, F = self.hidden_size, self.Transformer[0].mlp.out.in_channels
        n = mod.shape[0]
        fb = 6 * D + 2 * F
        fold = torch.empty((n, self.num_blocks * fb), dtype=torch.float32, device=mod.device)
        ones = torch.ones((1, D), dtype=torch.float32, device=mod.device)
        for l in range(self.num_blocks):
            m0, f0 = l * 6 * D, l * fb
            for (w, b, sh, sc, off) in ((P["w_qkv"][l], P["b_qkv"][l], m0, m0 + D, f0),
                                        (P["w_up"][l], P["b_up"][l], m0 + 3 * D, m0 + 4 * D, f0 + 6 * D)):
                wf = ops.widen_bf16(w)
                N = wf.shape[0]
                ops.sgemm(mod[:, sc:sc + D], wf, ops.sgemm(ones, wf).view(-1), out=fold[:, off:off + N])   # S (row sums as the bias)
                ops.sgemm(mod[:, sh:sh + D], wf, b, out=fold[:, off + N:off + 2 * N])               # C
        return fold

    # ------------------------------------------------------------------ reference API
    def label_embedding(self, label):
        """LabelEmbedding (model/layers.py:44-52): mlp(Embedding[label]) -> (B, t_dim)."""
        if not hasattr(self, "LabelEmbedding"):
            raise ValueError("label given but cfg.score.num_categorys <= 1 (no LabelEmbedding, score.py:103-106)")
        le = self.LabelEmbedding
        e = le.label_emb.weight.detach()[label.to(self._device()).long()].float().contiguous()      # row gather
        h = ops.sgemm(e, le.mlp[0].weight, le.mlp[0].bias, act_out=ACT_SILU)
        return ops.sgemm(h, le.mlp[2].weight, le.mlp[2].bias)

    @torch.no_grad()
    def forward(self, x, t, label=None, condition=None):
        """x (bs, tokens, z_dim), t (bs,) -> predicted noise (bs, tokens, z_dim)   [score.py:117-151]

        label: (bs,) class ids (LabelEmbedding, num_categorys > 1).  condition: the EMBEDDED pair the reference's
        ConditionNet returns — (pts_condition (bs, hidden, S) or None, img_condition (bs, t_dim) or 0.) — cross-
        attended on even blocks / added to the time embedding (score.py:135,148-149; a label wins over the image
        condition by the reference's operator precedence).  A raw dict {'img':…, 'pts':…} goes through `self.c_net`
        (cfg.score.condition=True) first, as upstream (:129-131)."""
        if not x.is_cuda:
            raise RuntimeError("Score.forward: x is on %s; the HIP path has no CPU fallback" % x.device)
        if self.training and self.dropout > 0:
            raise RuntimeError("Score.forward: dropout=%g needs eval() — the HIP path is the inference (sampling) path" % self.dropout)
        B, T, z = x.shape
        assert z == self.z_dim
        x = x.contiguous().float()
        if self.unet:
            return self._forward_unet(x, t, label, condition)
        if self.host_blocks:
            return self._forward_host_blocks(x, t, label, condition)
        extra, kv, S = self.condition_embedding(label, condition)
        _, mod = self.time_table(t.to(x).float(), extra_c=extra)
        plan = self.plan(B, T, mod, 0, self.n_mod, kv_cond=kv, cond_tokens=S)      # per-sample AdaLN rows
        out = torch.empty_like(x)
        check(lib().ldt_score_forward(ctypes.byref(plan), x.data_ptr(), out.data_ptr(), None, ops.stream_ptr()),
              "ldt_score_forward")
        return out

    @torch.no_grad()
    def forward_shared_t(self, x, t, fold=None):
        """One Score evaluation with every sample at the SAME time `t` (a float) — what each step of the fused sampling
        loop runs (diffusion_continuous.py:243-244 builds vec_t = ones * t): batch-shared AdaLN rows and, where
        `can_fold(B, T)` holds (or fold=True forces it), the LN-folded GEMM epilogues.  Same plan and kernels as
        `ldt_sample_loop`'s step; used by the full-size parity tests and bench.py's per-kernel timing."""
        if not x.is_cuda:
            raise RuntimeError("Score.forward_shared_t: x is on %s; the HIP path has no CPU fallback" % x.device)
        if self.host_blocks:
            raise NotImplementedError("forward_shared_t: the U-Net / non-LayerNorm variants are host-driven (forward)")
        B, T, z = x.shape
        assert z == self.z_dim
        x = x.contiguous().float()
        _, mod = self.time_table(torch.tensor([float(t)], dtype=torch.float32, device=x.device))
        use_fold = self.can_fold(B, T) if fold is None else bool(fold)
        plan = self.plan(B, T, mod, self.n_mod, 0, fold=self.fold_table(mod) if use_fold else None)
        out = torch.empty_like(x)
        check(lib().ldt_score_forward(ctypes.byref(plan), x.data_ptr(), out.data_ptr(), None, ops.stream_ptr()),
              "ldt_score_forward")
        return out

    @torch.no_grad()
    def forward_table_row(self, x, step_index, mod, fold=None):
        """One Score evaluation reading row `step_index` of a MULTI-step batch-shared AdaLN table `mod` [n, n_mod] (and of its
        LN-folding table `fold` [n, ...] when given) — exactly how step `step_index` of `ldt_sample_loop` addresses them
        (base + step * stride, the step in device memory).  The parity tests use it to check the 1000-row tables the headline
        config builds at rows far from 0 (tests/test_gpu_fullsize.py::test_fullsize_teacher_forced_rows_of_1000_step_tables)."""
        if not x.is_cuda:
            raise RuntimeError("Score.forward_table_row: x is on %s; the HIP path has no CPU fallback" % x.device)
        B, T, z = x.shape
        assert z == self.z_dim and mod.shape[1] == self.n_mod and 0 <= int(step_index) < mod.shape[0]
        x = x.contiguous().float()
        step = torch.full((1,), int(step_index), dtype=torch.int32, device=x.device)
        plan = self.plan(B, T, mod, self.n_mod, 0, fold=fold)
        out = torch.empty_like(x)
        check(lib().ldt_score_forward(ctypes.byref(plan), x.data_ptr(), out.data_ptr(), step.data_ptr(), ops.stream_ptr()),
              "ldt_score_forward")
        return out

    def _forward_host_blocks(self, x, t, label, condition):
        """The plain block stack with `norm` other than layer_norm (score.py:117-151 with get_norm's group_norm / None, tools/utils.py:168-181):
        host-driven, every block the kernel chain of ldt_amd/blocks.py (norm kernel -> GEMMs -> attention -> GEMMs); cross-attention to the
        point condition on the even blocks (:148-149) with K / V from the RAW condition rows."""
        from ._lib import EPI_F32
        from .blocks import final_layer, residual_block
        if isinstance(condition, dict):
            if not hasattr(self, "c_net"):
                raise ValueError("a raw condition dict needs cfg.score.condition=True (ConditionNet, score.py:64-65)")
            condition = self.c_net(condition)
        pts_cond, img_cond = (None, 0.) if condition is None else condition
        B, T, _ = x.shape
        P = self.packed()
        c = self.time_embedding(t.to(x).float())
        if label is not None:
            c = ops.add_f32(c, self.label_embedding(label))
        elif torch.is_tensor(img_cond):
            c = ops.add_f32(c, img_cond.to(x).expand_as(c).contiguous())
        y, S = None, None
        if torch.is_tensor(pts_cond):                                # (B, hidden, S) channels-first -> token-major raw rows, bf16
            S = pts_cond.shape[2]
            yt = pts_cond.to(self._device(), torch.float32).transpose(1, 2).contiguous().view(B * S, self.hidden_size)
            y = ops.cast_pad_bf16(yt, self.hidden_size)
        xin = ops.cast_pad_bf16(x.view(B * T, self.z_dim), ops.pad64(self.z_dim))
        h = ops.gemm_bf16(xin, P["w_in"], P["b_in"], EPI_F32)                       # ln_in
        for l, Pb in enumerate(P["blocks"]):
            if y is not None and l % 2 == 0:
                residual_block(Pb, h, B, T, y_bf16=y, Nk=S, c=c)
            else:
                residual_block(Pb, h, B, T, c=c)
        return final_layer(P["final"], h, B, T, c).view(B, T, self.z_dim)

    def _forward_unet(self, x, t, label, condition):
        """`unet: True` variant (score.py:138-146): num_blocks//2 up blocks whose outputs are kept, a mid block, then
        num_blocks//2 down blocks on cat(x, skip) (width 2*hidden -> hidden, conv shortcut, adaLN1/adaLN2).  Host-driven:
        each block is the same sequence of HIP kernels as the Compressor's blocks (ldt_amd/blocks.py)."""
        from ._lib import EPI_F32
        from .blocks import final_layer, residual_block
        pts_cond, img_cond = (None, 0.) if condition is None else (self.c_net(condition) if isinstance(condition, dict) else condition)
        if torch.is_tensor(pts_cond):
            raise NotImplementedError("unet + point condition: the reference passes the (B,hidden,S) condition as K/V source to "
                                      "down blocks whose fc_kv expects 2*hidden channels (score.py:80,146) and fails there")
        B, T, _ = x.shape
        P = self.packed()
        c = self.time_embedding(t.to(x).float())
        if label is not None:
            c = ops.add_f32(c, self.label_embedding(label))
        elif torch.is_tensor(img_cond):
            c = ops.add_f32(c, img_cond.to(x).expand_as(c))
        xin = ops.cast_pad_bf16(x.view(B * T, self.z_dim), ops.pad64(self.z_dim))
        h = ops.gemm_bf16(xin, P["w_in"], P["b_in"], EPI_F32)                       # ln_in
        skips = [h.clone()]
        for Pb in P["up"]:
            residual_block(Pb, h, B, T, c=c)
            skips.append(h.clone())
        residual_block(P["mid"], h, B, T, c=c)
        for Pb in P["down"]:
            h = residual_block(Pb, torch.cat((h, skips.pop()), dim=1), B, T, c=c)   # channels: [x | skip]  (:145)
        return final_layer(P["final"], h, B, T, c).view(B, T, self.z_dim)
