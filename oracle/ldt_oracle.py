"""TEST INFRASTRUCTURE ONLY — CPU restatement (the parity oracle) of LDT's sampling hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product (`ldt_amd/`) never does.

What it is: a from-scratch, *functional*, token-major PyTorch-fp32 restatement of
the reference algorithm.  Every function takes a plain
`state_dict` (reference parameter names/shapes: Conv1d weights are (out,in,1))
and cites the reference file:line it follows (paths relative to /root/reference).

Pinning: `tests/test_oracle_golden.py` checks every function here against golden
vectors captured from the *imported reference itself* by `oracle/gen_golden.py`
(fixtures under tests/golden/).  Exception — **FPS parity unpinned**: the
reference calls the third-party `pointnet2_ops` CUDA kernel (not vendored, not in
this image; call site model/Compressor/layers.py:106); we restate the algorithm of
its vendored twin model/functional/src/sampling/sampling.cu:86-167 instead.

Layout convention: the reference is channels-first (B,C,T) with 1x1 Conv1d; here
everything is token-major X[B,T,C] and a Conv1d weight W(out,in,1) is X @ W[:,:,0].T + b.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- helpers


def _w(sd, name):
    """Weight as a 2-D (out,in) matrix whether it is Linear (out,in) or Conv1d (out,in,1)."""
    w = sd[name]
    return w[:, :, 0] if w.dim() == 3 else w


def linear(sd, prefix, x):
    """1x1 Conv1d / Linear: x[..., in] -> [..., out]."""
    return F.linear(x, _w(sd, prefix + ".weight"), sd.get(prefix + ".bias"))


def layer_norm(x, weight=None, bias=None):
    """tools/utils.py:127-133: nn.LayerNorm(C, eps=1e-6) over channels."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, 1e-6)


def apply_norm(sd, prefix, x, kind="layer_norm", affine=False):
    """tools/utils.py:168-181 get_norm applied the way the blocks apply it (model/layers.py:163-164,218-219,224-226,235,244) on token-major
    x [B, N, C]: `layer_norm` (keys <prefix>.norm.{weight,bias} only when `affine`: blocks / final layers WITHOUT a condition),
    `group_norm` (nn.GroupNorm(min(C // 4, 16), C, eps=1e-6) on the channels-first (B, C, N) tensor, keys <prefix>.{weight,bias}, always affine),
    None (Identity).  (`batch_norm` is undefined upstream: its wrapper normalises the token axis of a (B, N, C) tensor with C running statistics.)"""
    if kind is None:
        return x
    k = str(kind).lower()
    if k == "layer_norm":
        return layer_norm(x, sd.get(prefix + ".norm.weight") if affine else None, sd.get(prefix + ".norm.bias") if affine else None)
    if k == "group_norm":
        C = x.shape[-1]
        return F.group_norm(x.transpose(1, 2), min(C // 4, 16), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-6).transpose(1, 2)
    raise NotImplementedError("norm %r" % (kind,))


def modulate(x, shift, scale):
    """model/layers.py:136-137 (shift/scale broadcast over tokens)."""
    return x * (1 + scale) + shift


# ----------------------------------------------------------------------------- time embedding


def sinusoid(t, dim):
    """model/layers.py:20-36 (calc_t_emb): continuous t, [sin | cos], fp32 frequency product."""
    half = dim // 2
    s = np.log(10000) / (half - 1)           # python/fp64 scalar, :29
    f = torch.exp(torch.arange(half) * -s)   # int64 * python float -> fp32 product, :30
    e = t.unsqueeze(1) * f                   # :33
    return torch.cat((torch.sin(e), torch.cos(e)), 1)


def time_embedding(sd, prefix, t, t_emb_dim):
    """model/layers.py:14-41: Linear -> SiLU -> Linear on the sinusoid."""
    e = sinusoid(t, t_emb_dim)
    h = F.silu(linear(sd, prefix + ".mlp.0", e))
    return linear(sd, prefix + ".mlp.2", h)


# ----------------------------------------------------------------------------- attention / block


def attention(sd, prefix, xq, xkv, num_heads):
    """model/layers.py:183-200 compute_attention.

    xq [B,N,C] (query source), xkv [B,M,Ckv].  K = first C output channels of
    fc_kv, V = last C (:189).  Head merge is a RAW reinterpret of the contiguous
    (B,H,N,Dh) result as (B,N,C) (:197, quirk Q1)."""
    B, N, _ = xq.shape
    q = linear(sd, prefix + ".fc_q", xq)
    kv = linear(sd, prefix + ".fc_kv", xkv)
    C = q.shape[-1]
    M = kv.shape[1]
    dh = C // num_heads
    k, v = kv[..., :C], kv[..., C:]
    qh = q.reshape(B, N, num_heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(B, M, num_heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, M, num_heads, dh).permute(0, 2, 1, 3)
    w = (qh @ kh.transpose(-2, -1)) * (dh ** -0.5)
    w = w.softmax(dim=-1)
    o = (w @ vh).contiguous()                  # (B,H,N,Dh) contiguous
    o = o.reshape(B, N, C)                     # raw reinterpret, no head permute-back
    return linear(sd, prefix + ".fc_o", o)


def mlp(sd, prefix, x):
    """model/layers.py:110-133 with n_hidden=1: Conv -> GELU(erf) -> Conv."""
    return linear(sd, prefix + ".out", F.gelu(linear(sd, prefix + ".fc.0.0", x)))


def block_act(name):
    """tools/utils.py:104-124 get_activation as a function (eval mode: rrelu = its mean slope; unknown names are ReLU there)."""
    if name is None:
        return lambda v: v
    n = str(name).lower()
    table = {"gelu": F.gelu, "silu": F.silu, "swish": F.silu, "selu": F.selu, "hardswish": F.hardswish,
             "leakyrelu": lambda v: F.leaky_relu(v, 0.01), "leakyrelu0.2": lambda v: F.leaky_relu(v, 0.2),
             "rrelu": lambda v: F.rrelu(v, training=False)}
    return table.get(n, F.relu)


def residual_block(sd, prefix, x, y, c, num_heads, act=None, norm="layer_norm"):
    """model/layers.py:202-229.

    c is not None  -> AdaLN branch (:212-219), LayerNorm without affine.
       y is None   -> K/V from the modulated-normalised x (Score, :184-185)
       y given     -> K/V from RAW y (Compressor Encoder passes y=x, quirk Q2)
       dim_in != dim_out (U-Net down block: `<prefix>.shortcut` exists) -> adaLN1 gives (shift, scale) of width dim_in,
       adaLN2 gives (gate_msa, shift_mlp, scale_mlp, gate_mlp) of width dim_out, the skip path is the 1x1 conv (:216-218)
    c is None      -> no-condition branch (:224-226), LayerNorm WITH affine, act=Identity
                      (decoder_act: ~ in the shipped config)."""
    if c is not None and prefix + ".shortcut.weight" in sd:
        sh1, sc1 = linear(sd, prefix + ".adaLN1.1", F.silu(c))[:, None, :].chunk(2, dim=-1)
        g1, sh2, sc2, g2 = linear(sd, prefix + ".adaLN2.1", F.silu(c))[:, None, :].chunk(4, dim=-1)
        h = modulate(apply_norm(sd, prefix + ".norm1", x, norm), sh1, sc1)
        x = linear(sd, prefix + ".shortcut", x) + g1 * attention(sd, prefix, h, h if y is None else y, num_heads)
        x = x + g2 * mlp(sd, prefix + ".mlp", modulate(apply_norm(sd, prefix + ".norm2", x, norm), sh2, sc2))
    elif c is not None:
        m = linear(sd, prefix + ".adaLN.1", F.silu(c))                   # [B,6C], or [B,T,6C] for a per-token condition (:210)
        m = m[:, None, :] if m.dim() == 2 else m
        sh1, sc1, g1, sh2, sc2, g2 = m.chunk(6, dim=-1)
        h = modulate(apply_norm(sd, prefix + ".norm1", x, norm), sh1, sc1)
        x = x + g1 * attention(sd, prefix, h, h if y is None else y, num_heads)
        x = x + g2 * mlp(sd, prefix + ".mlp", modulate(apply_norm(sd, prefix + ".norm2", x, norm), sh2, sc2))
    else:                                # (a block BUILT with a condition but called without one has no affine: :172-173)
        fa = block_act(act)              # :224-226: self.act behind both norms (`decoder_act`; Identity in the shipped configs)
        h = fa(apply_norm(sd, prefix + ".norm1", x, norm, affine=True))
        x = x + attention(sd, prefix, h, h if y is None else y, num_heads)
        h = fa(apply_norm(sd, prefix + ".norm2", x, norm, affine=True))
        x = x + mlp(sd, prefix + ".mlp", h)
    return x


def final_layer(sd, prefix, x, c, norm="layer_norm"):
    """model/layers.py:232-248: chunk order is (shift, scale)."""
    m = linear(sd, prefix + ".adaLN.1", F.silu(c))
    m = m[:, None, :] if m.dim() == 2 else m
    sh, sc = m.chunk(2, dim=-1)
    return linear(sd, prefix + ".ln", modulate(apply_norm(sd, prefix + ".norm", x, norm), sh, sc))


# ----------------------------------------------------------------------------- Score


def score_forward(sd, cfg, x, t, label_emb=None, condition=None, trace=None):
    """model/scorenet/score.py:117-151 (both the plain stack and, when cfg.unet, the up/mid/down variant :138-146).

    x [B,T,z], t [B].  condition = (pts_cond [B,S,hidden] token-major or None, img_cond [B,t_dim] or 0.)
    label_emb: already-embedded label [B,t_dim] (LabelEmbedding, layers.py:44-52) — label wins over
    the image condition (operator precedence at score.py:135)."""
    pts_cond, img_cond = (None, 0.) if condition is None else condition
    t_emb = time_embedding(sd, "TimeEmbedding", t, cfg.t_dim // 4)
    c = t_emb + label_emb if label_emb is not None else t_emb + img_cond
    h = linear(sd, "ln_in", x)
    nk = getattr(cfg, "norm", "layer_norm")                          # score.py:58 -> every block's and the final layer's get_norm
    if trace is not None:
        trace.append(("c", c.clone()))
        trace.append(("ln_in", h.clone()))
    if getattr(cfg, "unet", False):
        skips = [h]
        for i in range(cfg.num_blocks // 2):
            h = residual_block(sd, "Transformer_Up.%d" % i, h, pts_cond, c, cfg.num_heads, norm=nk)      # every layer gets y (:141)
            skips.append(h)
        h = residual_block(sd, "Transformer_Mid", h, pts_cond, c, cfg.num_heads, norm=nk)
        for i in range(cfg.num_blocks // 2):
            h = torch.cat((h, skips.pop()), dim=-1)                                            # channels [x | skip] (:145)
            h = residual_block(sd, "Transformer_Down.%d" % i, h, pts_cond, c, cfg.num_heads, norm=nk)
        return final_layer(sd, "ln_out", h, c, norm=nk)
    for i in range(cfg.num_blocks):
        y = pts_cond if (i % 2 == 0) else None
        h = residual_block(sd, "Transformer.%d" % i, h, y, c, cfg.num_heads, norm=nk)
        if trace is not None:
            trace.append(("block%d" % i, h.clone()))
    return final_layer(sd, "ln_out", h, c, norm=nk)


# ----------------------------------------------------------------------------- VPSDE + samplers


class VPSDE:
    """diffusion/diffusion_continuous.py:626-678 (DiffusionVPSDE), discrete sample mode."""

    def __init__(self, sde_cfg):
        self.beta_start = sde_cfg.beta_start
        self.beta_end = sde_cfg.beta_end
        self.sigma2_0 = sde_cfg.sigma2_0
        self.N = sde_cfg.sample_N
        # :649-653 — fp64 linspace cast to fp32
        self.betas = torch.from_numpy(
            np.linspace(self.beta_start / self.N, self.beta_end / self.N, self.N, dtype=np.float64)).to(torch.float32)
        self.alpha = 1.0 - self.betas
        self.alphas_cump = self.alpha.cumprod(dim=0)

    def g2(self, t):  # :658-659
        return self.beta_start + (self.beta_end - self.beta_start) * t

    def f(self, t):  # :655-656
        return -0.5 * self.g2(t)

    def var(self, t):  # :664-666 (same op order => same fp32 rounding)
        return 1.0 - (1.0 - self.sigma2_0) * torch.exp(
            -self.beta_start * t - 0.5 * (self.beta_end - self.beta_start) * t * t)

    def std(self, t):  # :668-669
        return torch.sqrt(self.var(t))

    def e2int_f(self, t):  # :671-672
        return torch.exp(-0.5 * self.beta_start * t - 0.25 * (self.beta_end - self.beta_start) * t * t)


class SubVPSDE:
    """diffusion/diffusion_continuous.py:681-729 (DiffusionSubVPSDE); no betas table (ancestral / ddim raise AttributeError upstream too)."""

    def __init__(self, sde_cfg):
        self.beta_start = sde_cfg.beta_start
        self.beta_end = sde_cfg.beta_end
        self.sigma2_0 = sde_cfg.sigma2_0

    def beta(self, t):  # :715-717
        return self.beta_start + (self.beta_end - self.beta_start) * t

    def f(self, t):  # :699-700
        return -0.5 * self.beta(t)

    def g2(self, t):  # :702-703
        return self.beta(t) * (1.0 - torch.exp(-2.0 * self.beta_start * t - (self.beta_end - self.beta_start) * t * t))

    def var(self, t):  # :705-707
        int_term = torch.exp(-self.beta_start * t - 0.5 * (self.beta_end - self.beta_start) * t * t)
        return torch.square(1.0 - int_term) + self.sigma2_0 * int_term

    def std(self, t):
        return torch.sqrt(self.var(t))

    def e2int_f(self, t):  # :709-710
        return torch.exp(-0.5 * self.beta_start * t - 0.25 * (self.beta_end - self.beta_start) * t * t)


class VESDE:
    """diffusion/diffusion_continuous.py:732-766 (DiffusionVESDE): dz = sqrt(beta(t)) dW."""

    def __init__(self, sde_cfg):
        self.sigma2_min = sde_cfg.sigma2_min
        self.sigma2_max = sde_cfg.sigma2_max
        self.sigma2_0 = sde_cfg.sigma2_0
        assert self.sigma2_min == self.sigma2_0                      # :741

    def f(self, t):  # :743-744
        return torch.zeros_like(t)

    def g2(self, t):  # :746-747
        return self.sigma2_min * np.log(self.sigma2_max / self.sigma2_min) * ((self.sigma2_max / self.sigma2_min) ** t)

    def var(self, t):  # :749-750
        return self.sigma2_min * ((self.sigma2_max / self.sigma2_min) ** t) - self.sigma2_min + self.sigma2_0

    def std(self, t):
        return torch.sqrt(self.var(t))

    def e2int_f(self, t):  # :752-753
        return torch.ones_like(t)


class GeometricSDE:
    """diffusion/diffusion_continuous.py:595-623 (DiffusionGeometric): VP drift, geometric progression of the variance."""

    def __init__(self, sde_cfg):
        self.sigma2_min = sde_cfg.sigma2_min
        self.sigma2_max = sde_cfg.sigma2_max
        self.sigma2_0 = sde_cfg.sigma2_0

    def f(self, t):  # :606-607
        return -0.5 * self.g2(t)

    def g2(self, t):  # :609-612
        sigma2_geom = self.sigma2_min * ((self.sigma2_max / self.sigma2_min) ** t)
        log_term = np.log(self.sigma2_max / self.sigma2_min)
        return sigma2_geom * log_term / (1.0 - self.sigma2_0 + self.sigma2_min - sigma2_geom)

    def var(self, t):  # :614-615
        return self.sigma2_min * ((self.sigma2_max / self.sigma2_min) ** t) - self.sigma2_min + self.sigma2_0

    def std(self, t):
        return torch.sqrt(self.var(t))

    def e2int_f(self, t):  # :617-619
        return torch.sqrt(1.0 + self.sigma2_min * (1.0 - (self.sigma2_max / self.sigma2_min) ** t) / (1.0 - self.sigma2_0))


def make_sde(sde_cfg):
    """diffusion/diffusion_continuous.py:18-29 (make_diffusion)."""
    return {"vpsde": VPSDE, "sub_vpsde": SubVPSDE, "vesde": VESDE, "geometric_sde": GeometricSDE}[sde_cfg.sde_type](sde_cfg)


def score_fn_from_model(sde, model_fn):
    """trainer/Latent_SDE_Trainer.py:57-61: params -> (score, params)."""
    def fn(t, x):
        params = model_fn(x, t)
        var = sde.var(t)[:, None, None]
        return -params / torch.sqrt(var), params
    return fn


def sample_discrete(sde, score_fn, x0, noises, N, predictor="ancestral", time_eps=1e-6,
                    denoise=True, probability_flow=False, record=None, max_steps=None,
                    corrector=None, corrector_steps=1, snr=0.01, print_steps=None, progress=None):
    """diffusion/diffusion_continuous.py:133-258,318-338 (pc_sampling).

    x0 [B,T,z] is the initial N(0,1) draw (:237); `noises` are the randn_like draws in consumption order: one per
    predictor call (:160 etc.; the last one is drawn but unused when denoise, quirk Q8) followed by
    `corrector_steps` per corrector call.  corrector: None, 'ancestral' (AncestralCorrector :212-229, alpha = 1
    by quirk Q11) or 'langevin' (LangevinCorrector :193-210: its (B,) step size is indexed [:, None] against
    (B,T,z) latents, which broadcasts only when B == 1 or B == T — kept as is, torch raises otherwise).
    PNDM (:260-316) is `sample_pndm` below.
    record: optional list that receives (x_in, params, x_mean, x_out) per predictor step.
    max_steps: stop after that many steps (bench.py's bounded CPU-baseline sample).
    print_steps: trajectory dump of :239-257 (returns the list).
    progress: optional callable(step index) invoked after every step (long CPU runs report that they are alive)."""
    T = 1.0
    B = x0.shape[0]
    x = x0
    timesteps = torch.linspace(T, time_eps, N)                      # :238
    x_mean = x
    it = iter(noises)
    if print_steps is not None:
        out_list = [x]
        every = (N - 1) // (print_steps - 2)
    for i in range(N if max_steps is None else min(N, max_steps)):
        t = torch.ones((B,)) * timesteps[i]                         # :243-244
        z = next(it)
        if predictor == "ancestral":                                # :152-162
            idx = (t * (N - 1) / T).long()
            beta = sde.betas[idx]
            score, params = score_fn(t, x)
            x_mean = (x + beta[:, None, None] * score) / torch.sqrt(1. - beta)[:, None, None]
            x_new = x_mean + torch.sqrt(beta)[:, None, None] * z
        elif predictor == "reversediffusion":                       # :141-150
            dt = torch.tensor((1 - time_eps) / N)
            f, g2 = sde.f(t)[:, None, None] * x, sde.g2(t)[:, None, None]
            score, params = score_fn(t, x)
            dx = (f - g2 * score * (0.5 if probability_flow else 1.)) * dt
            g = torch.zeros_like(g2) if probability_flow else torch.sqrt(g2)
            x_mean = x - dx
            x_new = x_mean + g * z * torch.sqrt(dt)
        elif predictor == "eulermaruyama":                          # :182-191
            dt = -1. / N
            f, g2 = sde.f(t)[:, None, None] * x, sde.g2(t)[:, None, None]
            score, params = score_fn(t, x)
            f = f - g2 * score * (0.5 if probability_flow else 1.)
            x_mean = x + f * dt
            g2 = torch.zeros(1) if probability_flow else g2
            x_new = x_mean + torch.sqrt(g2) * np.sqrt(-dt) * z
        elif predictor == "ddim":                                   # :164-180
            idx = (t * (N - 1) / T).long()
            at = sde.alphas_cump[idx][:, None, None]
            if idx[0] - 1 < 0:
                at_next = torch.ones_like(at)
            else:
                at_next = sde.alphas_cump[idx - 1][:, None, None]
            _, params = score_fn(t, x)
            x_mean = at_next.sqrt() * (x - (1 - at).sqrt() * params) / at.sqrt() + (1 - at_next).sqrt() * params
            x_new = x_mean + 0 * z
        else:
            raise NotImplementedError("preditor not Implemented")   # :328
        if record is not None:
            record.append((x, params, x_mean, x_new))
        x = x_new
        if corrector == "ancestral":                                # :212-229
            std = sde.std(t)
            for _ in range(corrector_steps):
                grad, params = score_fn(t, x)
                zc = next(it)
                step_size = (snr * std) ** 2 * 2 * torch.ones_like(t)
                x_mean = x + step_size[:, None, None] * grad
                x = x_mean + zc * torch.sqrt(step_size * 2)[:, None, None]
        elif corrector == "langevin":                               # :193-210 (alpha = ones_like(t): the class test at :195 is never true)
            alpha = torch.ones_like(t)
            for _ in range(corrector_steps):
                grad, params = score_fn(t, x)
                zc = next(it)
                grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
                noise_norm = torch.norm(zc.reshape(zc.shape[0], -1), dim=-1).mean()
                step_size = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
                x_mean = x + step_size[:, None] * grad              # (B,1) against (B,T,z): B == 1 or B == T only
                x = x_mean + torch.sqrt(step_size * 2)[:, None] * zc
        elif corrector is not None:
            raise NotImplementedError("corrector not Implemented")  # :335
        if print_steps is not None and (i + 1) % every == 0:
            out_list.append(x_mean)
        if progress is not None:
            progress(i)
    if print_steps is not None:
        out_list.append(x_mean if denoise else x)
        return out_list
    return x_mean if denoise else x


def sample_pndm(sde_cfg, score_fn, x0, time_eps):
    """diffusion/diffusion_continuous.py:260-316 (predictor == "pndm"): pseudo linear multistep sampling over
    sde_cfg.sample_N steps on sde_cfg.train_N training levels, three Runge-Kutta warm-up steps.  x0 [B,T,z] is the
    initial draw (:309).  transfer() views the gathered alphas as (B,1) against (B,T,z) latents (:267-268): runs only
    when B == 1 or B == T, as upstream.  The last step reads timesteps[-1] (t_next = 0 -> index -1, :307) — kept."""
    N, train_N = sde_cfg.sample_N, sde_cfg.train_N
    betas = torch.from_numpy(np.linspace(sde_cfg.beta_start / train_N, sde_cfg.beta_end / train_N, train_N,
                                         dtype=np.float64)).to(torch.float32)       # :310-313
    alphas_cump = torch.cat((torch.ones(1), (1.0 - betas).cumprod(dim=0)))          # :314-315
    timesteps = torch.linspace(time_eps, 1.0, N * 2)                                 # :262
    B = x0.shape[0]

    def transfer(x, t, t_next, et):                                                  # :263-274
        t = (train_N * (t - time_eps) + 1).long()
        t_next = (train_N * (t_next - time_eps) + 1).long()
        at = alphas_cump[t].view(-1, 1)
        at_next = alphas_cump[t_next].view(-1, 1)
        x_delta = (at_next - at) * ((1 / (at.sqrt() * (at.sqrt() + at_next.sqrt()))) * x - 1 / (at.sqrt() * (
            ((1 - at_next) * at).sqrt() + ((1 - at) * at_next).sqrt())) * et)
        return x + x_delta

    def at_time(i):
        return timesteps[i].view(-1).expand(B)

    sample, ets = x0, []
    for t in range(N, 0, -1):                                                        # :316-317
        t_next = t - 1
        if len(ets) > 2:                                                             # :296-300
            _, e = score_fn(at_time(t * 2 - 1), sample)
            ets.append(e)
            noise = (1 / 24) * (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4])
        else:                                                                        # runge_kutta :276-292
            t1, t2, t3 = at_time(t * 2 - 1), at_time(int((t + t_next) / 2 * 2) - 1), at_time(int(t_next * 2) - 1)
            _, e1 = score_fn(t1, sample)
            ets.append(e1)
            _, e2 = score_fn(t2, transfer(sample, t1, t2, e1))
            _, e3 = score_fn(t2, transfer(sample, t1, t2, e2))
            _, e4 = score_fn(t3, transfer(sample, t1, t3, e3))
            noise = (1 / 6) * (e1 + 2 * e2 + 2 * e3 + e4)
        sample = transfer(sample, at_time(t * 2 - 1), at_time(t_next * 2 - 1), noise)   # :304-307
    return sample


# ----------------------------------------------------------------------------- Compressor: decode


def initial_set(sd, B, num_points=None, keep_mask=None, seed_eps=None):
    """model/Compressor/layers.py:26-42.  max_outputs set: learned prior rows, token-major [B,N,D];
    keep_mask [B,max_outputs] bool selects rows (index order) when num_points < max_outputs;
    with num_points == max_outputs every row is kept in order (quirk Q9).
    max_outputs None (no 'init_set.prior' in the state_dict): seed_eps [B,N,n_mixtures,D] are the N(0,1) draws of :38,
    x = sum_m (eps sig + mu) softmax(logits)_m, then the `output` MLP (:39-41)."""
    if "init_set.prior" not in sd:
        w = torch.softmax(sd["init_set.logits"], dim=0)
        x = ((seed_eps * sd["init_set.sig"][None, None] + sd["init_set.mu"][None, None]) * w[None, None, :, None]).sum(2)
        return linear(sd, "init_set.output.2", F.silu(linear(sd, "init_set.output.0", x)))
    prior = sd["init_set.prior"]
    if keep_mask is None:
        return prior[None].expand(B, -1, -1)
    return torch.stack([prior[keep_mask[b]] for b in range(B)], 0)


def decoder_block(sd, prefix, o, eps_j, num_heads, c=None, act=None, norm="layer_norm"):
    """model/Compressor/Network.py:80-83: o <- att1(o, ln(eps_j), c) (K/V raw; c: label embedding under class_condition)."""
    z = linear(sd, prefix + ".ln", eps_j)
    return residual_block(sd, prefix + ".att1", o, z, c, num_heads, act=act, norm=norm)


def compressor_decode(sd, cfg, given_eps, keep_mask=None, seed_eps=None):
    """model/Compressor/Network.py:251-268 Compressor.sample: given_eps [B,T,n_layers*z_dim] -> [B,N,3]."""
    B = given_eps.shape[0]
    o = initial_set(sd, B, keep_mask=keep_mask, seed_eps=seed_eps)
    for j in range(cfg.n_layers):
        blk = "decoder.%d" % (cfg.n_layers - 1 - j)                 # reversed(self.decoder), :263
        e_j = given_eps[:, :, cfg.z_dim * j: cfg.z_dim * (j + 1)]  # split along channels, :261-262
        o = decoder_block(sd, blk, o, e_j, cfg.num_heads, act=getattr(cfg, "decoder_act", None), norm=getattr(cfg, "norm", "layer_norm"))
    return linear(sd, "output", o)                                  # postprocess = identity for xyz, :271-275


# ----------------------------------------------------------------------------- Compressor: encode


# Default of the near-origin rule: ON, as in upstream pointnet2_ops (the library every FPS call of the reference goes through:
# Compressor/layers.py:106, completion valsample :182-183); False = the vendored twin's behaviour (sampling.cu has no such rule).
FPS_SKIP_NEAR_ORIGIN = True


def fps(xyz, m, skip_near_origin=None):
    """Farthest point sampling, restating model/functional/src/sampling/sampling.cu:86-167
    (the vendored twin of pointnet2_ops' kernel; **parity unpinned**, see module header).
    Start index 0; distances init 1e38 (sampling.cpp:53-54); running min; argmax; ties go to the
    smaller (k % 512, k // 512) as the 512-thread strided scan + pairwise tree does (:141-158).
    Squared distance is (dx*dx + dy*dy) + dz*dz in fp32 without FMA contraction.
    skip_near_origin: what upstream pointnet2_ops (the library the reference really calls, model/Compressor/layers.py:106;
    not vendored — behaviour restated from its published kernel, unverifiable here) does in addition: points with
    |p|^2 <= 1e-3 neither update their distance nor can be selected (argmax starts from (best = -1, index 0))."""
    if skip_near_origin is None:
        skip_near_origin = FPS_SKIP_NEAR_ORIGIN
    xyz = np.ascontiguousarray(xyz.detach().cpu().numpy(), dtype=np.float32)
    b, n, _ = xyz.shape
    out = np.zeros((b, m), dtype=np.int64)
    k = np.arange(n)
    tie_rank = (k % 512) * (n // 512 + 1) + k // 512
    for bi in range(b):
        p = xyz[bi]
        dist = np.full((n,), np.float32(1e38), dtype=np.float32)
        live = np.ones((n,), dtype=bool)
        if skip_near_origin:
            live = ((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]) > np.float32(1e-3)
        old = 0
        for j in range(1, m):
            dx = p[:, 0] - p[old, 0]
            dy = p[:, 1] - p[old, 1]
            dz = p[:, 2] - p[old, 2]
            d = (dx * dx + dy * dy) + dz * dz
            dist = np.where(live, np.minimum(d, dist), dist)
            if not live.any():
                old = 0
            else:
                best = dist[live].max()
                cand = np.nonzero((dist == best) & live)[0]
                old = int(cand[np.argmin(tie_rank[cand])])
            out[bi, j] = old
    return torch.from_numpy(out)


def square_distance(src, dst):
    """model/Compressor/layers.py:65-84: -2ab + |a|^2 + |b|^2 (expanded form, in this op order)."""
    B, N, _ = src.shape
    M = dst.shape[1]
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def knn(k, xyz, centers):
    """model/Compressor/layers.py:87-98: k smallest squared distances, unordered."""
    d = square_distance(centers, xyz)
    return torch.topk(d, k, dim=-1, largest=False, sorted=False)[1]


def gather(points, idx):
    """model/Compressor/layers.py:46-62 index_points: points [B,N,C], idx [B,...] -> [B,...,C]."""
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def batch_norm_eval(sd, prefix, x):
    """nn.BatchNorm1d in eval mode on channel-last x (eps=1e-5)."""
    return F.batch_norm(x.reshape(-1, x.shape[-1]), sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, 1e-5).reshape(x.shape)


def local_grouper(sd, prefix, xyz, feat, groups, k, fps_idx=None, knn_idx=None, normalize="anchor"):
    """model/Compressor/layers.py:288-319 (normalize='anchor' or 'center', use_xyz=True).
    xyz [B,N,3], feat [B,N,D] -> centres [B,S,3], tokens [B,S,D]; also returns the index sets."""
    B, N, _ = xyz.shape
    if fps_idx is None:
        fps_idx = fps(xyz, groups)                                   # cluster :106
    new_xyz = gather(xyz, fps_idx)                                  # :107
    if knn_idx is None:
        knn_idx = knn(k, xyz, new_xyz)                              # :111
    new_feat = gather(feat, fps_idx)                                # :299
    g = torch.cat([gather(feat, knn_idx), gather(xyz, knn_idx)], dim=-1)     # :301-304  [B,S,k,D+3]
    if normalize == "center":
        mean = torch.mean(g, dim=2, keepdim=True)                   # :306-307
    else:
        mean = torch.cat([new_feat, new_xyz], dim=-1).unsqueeze(-2)  # :308-310 anchor
    std = torch.std((g - mean).reshape(B, -1), dim=-1, keepdim=True)[..., None, None]  # :311-312 unbiased
    g = (g - mean) / (std + 1e-5)                                   # :313
    g = sd[prefix + ".affine_alpha"] * g + sd[prefix + ".affine_beta"]       # :314
    u = torch.cat([g, new_feat[:, :, None, :].expand(-1, -1, k, -1)], dim=-1)  # :315  [B,S,k,2D+3]
    # PreExtraction :178-187
    e = prefix + ".extraction"
    h = F.relu(batch_norm_eval(sd, e + ".transfer.net.1", linear(sd, e + ".transfer.net.0", u)))
    r = F.relu(batch_norm_eval(sd, e + ".operation.0.net1.1", linear(sd, e + ".operation.0.net1.0", h)))
    r = linear(sd, e + ".operation.0.net2.0", r)
    h = F.relu(r + h)
    tokens = h.max(dim=2)[0]                                        # adaptive_max_pool1d over k
    return new_xyz, tokens, fps_idx, knn_idx


def condition_net_points(sd, prefix, pts, patch_size, fps_idx=None, knn_idx=None):
    """ConditionNet point branch (model/scorenet/score.py:20-23,37-41): Conv1d 3->128, LocalGrouper(128, 'center')
    into `patch_size` groups of k = 128 // patch_size * 2 neighbours (`x.shape[1]` there is the CHANNEL count 128,
    :40), Conv1d 128->hidden.  pts [B,N,3] -> pts_condition token-major [B,S,hidden] (reference: (B,hidden,S))."""
    x = linear(sd, prefix + ".pc_conv_in", pts)
    k = x.shape[-1] // patch_size * 2
    _, tok, fps_idx, knn_idx = local_grouper(sd, prefix + ".group", pts, x, patch_size, k, fps_idx, knn_idx,
                                             normalize="center")
    return linear(sd, prefix + ".pc_conv_out", tok), fps_idx, knn_idx


def _bn2d(sd, prefix, x, eps=1e-5):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, eps)


def _basic_block(sd, prefix, x, stride):
    """torchvision BasicBlock: conv3x3(stride) BN ReLU conv3x3 BN (+ 1x1-conv/BN downsample of the input) ReLU."""
    h = F.relu(_bn2d(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1)))
    h = _bn2d(sd, prefix + ".bn2", F.conv2d(h, sd[prefix + ".conv2.weight"], None, 1, 1))
    if prefix + ".downsample.0.weight" in sd:
        x = _bn2d(sd, prefix + ".downsample.1", F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0))
    return F.relu(h + x)


def condition_net_image(sd, prefix, img):
    """ConditionNet image branch (score.py:24-27,33-36): the first six children of torchvision's resnet18
    (conv1 7x7/2, bn1, relu, maxpool 3x3/2, layer1 = 2 BasicBlocks(64), layer2 = 2 BasicBlocks(128, first stride 2)),
    global max pool, Linear 128 -> p_dim.  torchvision is ABSENT from this image: the trunk is restated from its
    published architecture, PARITY UNPINNED; eval-mode BatchNorm.  img [B,3,H,W] -> [B,p_dim]."""
    r = prefix + ".resnet"
    h = F.relu(_bn2d(sd, r + ".1", F.conv2d(img, sd[r + ".0.weight"], None, 2, 3)))
    h = F.max_pool2d(h, 3, 2, 1)
    for layer, stride in ((4, 1), (5, 2)):
        for blk in (0, 1):
            h = _basic_block(sd, "%s.%d.%d" % (r, layer, blk), h, stride if blk == 0 else 1)
    h = F.adaptive_max_pool2d(h, 1).flatten(1)            # .squeeze() upstream (drops the batch axis too when B == 1)
    return linear(sd, prefix + ".ln", h)


def mini_pointnet(sd, prefix, centers):
    """model/Compressor/Network.py:86-101: centres [B,S,3] -> [B,p_dim]."""
    h = F.relu(batch_norm_eval(sd, prefix + ".bn1", linear(sd, prefix + ".conv1", centers)))
    h = F.relu(batch_norm_eval(sd, prefix + ".bn2", linear(sd, prefix + ".conv2", h)))
    return linear(sd, prefix + ".fc", h.max(dim=1)[0])


def act_norm(sd, prefix, x):
    """model/layers.py:103-107 (eval): per-token/channel shift & log_scale of shape (1,T,C)."""
    return (x - sd[prefix + ".shift"]) * torch.exp(-sd[prefix + ".log_scale"])


def compressor_encode(sd, cfg, pts, post_noise, fps_idx=None, knn_idx=None, keep_mask=None, seed_eps=None, label=None):
    """model/Compressor/Network.py:188-249 Compressor.forward (ActNorm True;
    cfg.norm_input -> norm_pts :170-174, cfg.pre_group -> a first LocalGrouper of 256 groups x 32 neighbours :193-194,
    cfg.pos_embedding 'mlp' -> a per-token position condition MLP(centres) :133-134, cfg.class_condition + label ->
    LabelEmbedding added to the position condition and fed to the decoder blocks :137-142,197-198,218,225).

    pts [B,N,3]; post_noise: list of n_layers tensors [B,T,z_dim] token-major — the N(0,1) draws
    of `sample(mu, logvar)` (:26-29) in consumption order.  Returns dict with 'all_eps' [B,T,n*z],
    'set' [B,N,3], plus intermediates for parity tests."""
    B, N, _ = pts.shape
    T = cfg.z_scales
    nk = getattr(cfg, "norm", "layer_norm")                                     # Network.py:114 -> every block's / final layer's get_norm
    if getattr(cfg, "norm_input", False):                                       # :189-190
        pts = (pts - pts.mean(dim=1, keepdim=True)) / pts.std(dim=1, keepdim=True)
    feat = linear(sd, "input", pts)                                             # :192
    if getattr(cfg, "pre_group", False):                                        # :193-194
        pts, feat, _, _ = local_grouper(sd, "pre_grouper", pts, feat, 256, 32)
        N = 256
    centers, x, fps_idx, knn_idx = local_grouper(sd, "group", pts, feat, T, N // T * 2, fps_idx, knn_idx)  # :195
    if getattr(cfg, "pos_embedding", "center") == "mlp":                        # :133-134, :196  [B,T,p_dim]
        pos = linear(sd, "pos_embedding.out", F.gelu(linear(sd, "pos_embedding.fc.0.0", centers)))
    else:
        pos = mini_pointnet(sd, "pos_embedding", centers)                       # :196  [B,p_dim]
    l_emb = None
    if label is not None and getattr(cfg, "class_condition", False):            # :240-241, layers.py:44-52
        e = sd["LabelEmbedding.label_emb.weight"][label]
        l_emb = linear(sd, "LabelEmbedding.mlp.2", F.silu(linear(sd, "LabelEmbedding.mlp.0", e)))
        pos = pos + l_emb                                                       # :197-198
    if getattr(cfg, "ActNorm", True) is not None:
        x = act_norm(sd, "conv_in", x)                                          # :200-201
    enc_out = []
    for i in range(cfg.n_layers):                                               # :203-205
        for j in range(cfg.encoder_layers):
            x = residual_block(sd, "encoder.%d.atts.%d" % (i, j), x, x, pos, cfg.num_heads, norm=nk)   # y = raw x (Q2)
        enc_out.append(final_layer(sd, "encoder.%d.conv_out" % i, x, pos, norm=nk))
    o = initial_set(sd, B, keep_mask=keep_mask, seed_eps=seed_eps)              # :215
    all_eps, mus, logvars = [], [], []
    for j in range(cfg.n_layers):                                               # :217-225
        blk = "decoder.%d" % (cfg.n_layers - 1 - j)
        xj = enc_out[-j - 1]
        p = residual_block(sd, blk + ".att", xj, o if j != 0 else xj, l_emb, cfg.num_heads, act=getattr(cfg, "decoder_act", None), norm=nk)   # :61-74
        post = linear(sd, blk + ".prior.1", F.silu(p))                          # :56,:72
        mu = post[..., :cfg.z_dim]
        logvar = post[..., cfg.z_dim:].clamp(cfg.min_sigma, 10.)                # :76
        eps = mu + torch.exp(logvar / 2.) * post_noise[j]                       # :26-29
        o = decoder_block(sd, blk, o, eps, cfg.num_heads, l_emb, act=getattr(cfg, "decoder_act", None), norm=nk)   # :225
        all_eps.append(eps); mus.append(mu); logvars.append(logvar)
    out = linear(sd, "output", o)                                               # :231
    return {"set": out, "all_eps": torch.cat(all_eps, dim=-1), "mu": mus, "logvar": logvars,
            "fps_idx": fps_idx, "knn_idx": knn_idx, "centers": centers, "enc_out": enc_out, "max": x.max()}


# ----------------------------------------------------------------------------- Chamfer, Trainer.sample


def dist_chamfer(a, b):
    """evaluation/evaluation_metrics.py:23-33 distChamfer: squared-L2 NN both ways via bmm."""
    xx = torch.bmm(a, a.transpose(2, 1))
    yy = torch.bmm(b, b.transpose(2, 1))
    zz = torch.bmm(a, b.transpose(2, 1))
    rx = torch.diagonal(xx, dim1=1, dim2=2).unsqueeze(1).expand_as(xx)
    ry = torch.diagonal(yy, dim1=1, dim2=2).unsqueeze(1).expand_as(yy)
    P = rx.transpose(2, 1) + ry - 2 * zz
    return P.min(1)[0], P.min(2)[0]


def chamfer_cd(a, b):
    """evaluation/evaluation_metrics.py:88: CD = dl.mean(1) + dr.mean(1)."""
    dl, dr = dist_chamfer(a, b)
    return dl.mean(dim=1) + dr.mean(dim=1)


def trainer_sample(score_sd, comp_sd, cfg, x0, noises, record=None, progress=None):
    """trainer/Latent_SDE_Trainer.py:143-165 Trainer.sample (discrete mode, unconditional):
    returns (points [B,N,3], eps [B,T,z])."""
    sde = VPSDE(cfg.sde)
    fn = score_fn_from_model(sde, lambda x, t: score_forward(score_sd, cfg.score, x, t))
    eps = sample_discrete(sde, fn, x0, noises, cfg.sde.sample_N, predictor=cfg.sde.predictor,
                          time_eps=cfg.sde.sample_time_eps, denoise=cfg.sde.denoise,
                          probability_flow=cfg.sde.probability_flow, record=record, progress=progress)
    pts = compressor_decode(comp_sd, cfg.compressor, eps)
    return pts, eps


def draw_noises(seed, B, T, z, N):
    """The reference's draw order on the CPU generator: x0 (:237) then one randn per step (:160)."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn((B, T, z), generator=g)
    noises = [torch.randn((B, T, z), generator=g) for _ in range(N)]
    return x0, noises


def sample_model_ode(sde, score_fn, noise, ode_eps, ode_solver_tol, nfe=None):
    """diffusion/diffusion_continuous.py:88-131 (sample_model_ode): probability-flow ODE dx/dt = f(t) x - g2(t)/2 score
    integrated from t = 1 to ode_eps by `torchdiffeq.odeint(..., method="scipy_solver", options={"solver": "RK45"})`.
    torchdiffeq (requirements.txt:9, version unpinned) is NOT vendored: this restates what its scipy wrapper does with
    those arguments — decreasing times are solved as increasing s = -t with the negated function
    (odeint's time reversal), the state is flattened to a float64 numpy vector, scipy.integrate.solve_ivp(RK45) runs with
    rtol = atol = ode_solver_tol, every function evaluation converts to float32 tensors and back — **parity unpinned**.
    noise [B,T,z] is the initial state (the reference draws it on the GPU generator, :107).  Returns the state at ode_eps."""
    from scipy.integrate import solve_ivp
    shape = tuple(noise.shape)

    def fun(s, y):
        t = torch.tensor(-s, dtype=torch.float32).expand(shape[0])
        x = torch.from_numpy(np.asarray(y)).to(torch.float32).reshape(shape)
        score, _ = score_fn(t, x)
        dx = sde.f(t)[:, None, None] * x - 0.5 * sde.g2(t)[:, None, None] * score       # :102
        if nfe is not None:
            nfe.append(1)
        return (-dx).reshape(-1).double().numpy()

    sol = solve_ivp(fun, t_span=[-1.0, -float(ode_eps)], y0=noise.reshape(-1).double().numpy(), t_eval=[-1.0, -float(ode_eps)],
                    method="RK45", rtol=ode_solver_tol, atol=ode_solver_tol)
    return torch.from_numpy(sol.y[:, -1]).to(torch.float32).reshape(shape)


# ----------------------------------------------------------------------------- validation metrics (evaluation/)


def pairwise_cd(sample_pcs, ref_pcs):
    """evaluation/evaluation_metrics.py:165-199 `_pairwise_CD_`: M[i][j] = CD(sample_i, ref_j) = dl.mean + dr.mean."""
    rows = []
    for i in range(sample_pcs.shape[0]):
        a = sample_pcs[i:i + 1].expand(ref_pcs.shape[0], -1, -1).contiguous()
        dl, dr = dist_chamfer(a, ref_pcs)
        rows.append(dl.mean(dim=1) + dr.mean(dim=1))
    return torch.stack(rows, 0)


def emd_approxmatch_cost(xyz1, xyz2):
    """Approximate-matching EMD of evaluation/pytorch_structural_losses/src/approxmatch.cu: approxmatchkernel (:3-186)
    followed by matchcostkernel (:188-224) — what `match_cost(xyz1, xyz2)` returns per cloud pair (emd_approx_cuda,
    evaluation_metrics.py:40-46 then divides by n).  PARITY UNPINNED: the reference implementation is a CUDA
    extension that cannot be built or run in this image; this is a restatement of the vendored kernel's arithmetic
    (fp32, `expf` in place of `__expf`) and is checked against the exact assignment cost only as a bound.
    xyz1 [B,n,3], xyz2 [B,m,3] -> [B]."""
    out = []
    for b in range(xyz1.shape[0]):
        p1, p2 = xyz1[b].float(), xyz2[b].float()
        n, m = p1.shape[0], p2.shape[0]
        multiL, multiR = (1., float(n // m)) if n >= m else (float(m // n), 1.)      # :6-12 (integer division)
        d2 = ((p2[None, :, :] - p1[:, None, :]) ** 2).sum(-1)                          # [n(k), m(l)]
        remainL = torch.full((n,), multiL)
        remainR = torch.full((m,), multiR)
        match = torch.zeros((n, m))
        for j in range(7, -2, -1):                                                      # :22 j = 7 .. -1
            level = -float(4.0 ** j)
            e = torch.exp(level * d2)
            ratioL = remainL / (1e-9 + (e * remainR[None, :]).sum(1))                   # :27-58
            sumr = (e * ratioL[:, None]).sum(0) * remainR                               # :74-103
            consumption = torch.clamp(remainR / (sumr + 1e-9), max=1.0)
            ratioR = consumption * remainR
            remainR = torch.clamp(remainR - sumr, min=0.0)
            w = e * ratioL[:, None] * ratioR[None, :]                                   # :125-157
            match = match + w
            remainL = torch.clamp(remainL - w.sum(1), min=0.0)
        out.append((match * torch.sqrt(d2)).sum())                                      # matchcostkernel :188-224
    return torch.stack(out)


def lgan_mmd_cov(all_dist):
    """evaluation_metrics.py:234-246; all_dist (N_sample, N_ref)."""
    min_idx = all_dist.min(dim=1)[1]
    return {"mmd": all_dist.min(dim=0)[0].mean(), "cov": torch.tensor(float(min_idx.unique().numel()) / all_dist.shape[1])}


def knn_two_sample(Mxx, Mxy, Myy, k):
    """evaluation_metrics.py:202-231 `knn` (sqrt=False): leave-one-out k-NN accuracy."""
    n0, n1 = Mxx.shape[0], Myy.shape[0]
    label = torch.cat((torch.ones(n0), torch.zeros(n1)))
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.t(), Myy), 1)), 0)
    M = M + torch.diag(float("inf") * torch.ones(n0 + n1))
    idx = M.topk(k, 0, False)[1]
    count = sum(label[idx[i]] for i in range(k))
    pred = (count >= k / 2.).float()
    return {"acc": (label == pred).float().mean(), "tp": (pred * label).sum(), "fp": (pred * (1 - label)).sum(),
            "fn": ((1 - pred) * label).sum(), "tn": ((1 - pred) * (1 - label)).sum()}


def compute_cd_metrics(sample_pcs, ref_pcs):
    """evaluation_metrics.py:299-321 compute_CD_metrics."""
    M_rs = pairwise_cd(ref_pcs, sample_pcs)
    res = {k + "-CD": v for k, v in lgan_mmd_cov(M_rs.t()).items()}
    res["1-NN-CD-acc"] = knn_two_sample(pairwise_cd(ref_pcs, ref_pcs), M_rs, pairwise_cd(sample_pcs, sample_pcs), 1)["acc"]
    return res
