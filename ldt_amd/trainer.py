"""`Trainer` — sampling half of the reference trainer (trainer/Latent_SDE_Trainer.py), MI355X path.

Kept: `Trainer(cfg, model, compressor, device)`, `score_fn(t, x, label, condition) -> (score, params)` (:57-61),
`sample(num_samples, num_points, label, condition) -> (points, eps)` (:143-165), `valsample`-style "Sample rate"
timing (:178-181,206), EMA weight swap (:146,164; tools/utils.py:80-101) and the checkpoint keys (:232-235,
:251-256).  Training (`update*`, optimizers, schedulers) is out of scope for this path.

Multi-GPU: when torch.distributed is initialised, `sample(B)` runs rows [lo,hi) of the batch on this rank and
all-gathers the finished points/latents once (ldt_amd/dist.py); results do not depend on the world size when
every rank seeds the CPU generator identically (the reference's common_init, tools/utils.py:269-276).
"""
import time

import torch

from . import dist as ldist
from .diffusion import DiffusionSubVPSDE, DiffusionVESDE, DiffusionVPSDE


class EMAWeights:
    """The part of tools/utils.py:25-101 (EMA optimizer wrapper) the sampler touches: `state[p]['ema']` and
    `swap_parameters_with_ema`.  With no EMA state (fresh model) the swap is a no-op, as upstream (:93-94)."""

    def __init__(self, params, ema_decay):
        self.ema_decay = ema_decay
        self.apply_ema = ema_decay > 0.
        self.params = list(params)
        self.state = {}

    def load_ema(self, optim_state_dict):
        """Adopt the 'ema' tensors of a reference optimizer state_dict (`score_optim_state_dict`,
        Latent_SDE_Trainer.py:233): its param ids follow parameter order."""
        st = optim_state_dict.get("state", {})
        for i, p in enumerate(self.params):
            if i in st and "ema" in st[i]:
                self.state[p] = {"ema": st[i]["ema"].to(p.device, p.dtype)}

    def swap_parameters_with_ema(self, store_params_in_ema):
        if not self.apply_ema:
            return
        for p in self.params:
            if not p.requires_grad or p not in self.state or "ema" not in self.state[p]:
                continue
            ema = self.state[p]["ema"]
            if store_params_in_ema:
                tmp = p.data.detach()
                p.data = ema.detach()
                self.state[p]["ema"] = tmp
            else:
                p.data = ema.detach()


class Trainer:
    def __init__(self, cfg, model, compressor, device):
        self.cfg = cfg
        if cfg.sde.sde_type == "vpsde":                   # Latent_SDE_Trainer.py:23-28 (any other type leaves no self.SDE upstream)
            self.SDE = DiffusionVPSDE(cfg.sde)
        elif cfg.sde.sde_type == "sub_vpsde":
            self.SDE = DiffusionSubVPSDE(cfg.sde)
        elif cfg.sde.sde_type == "vesde":
            self.SDE = DiffusionVESDE(cfg.sde)
        self.sde_type = cfg.sde.sde_type
        self.num_points = cfg.data.tr_max_sample_points
        self.device = device
        self.num_categorys = cfg.data.num_categorys
        self.model = model.to(device)
        self.compressor = compressor.to(device)
        self.optimizer = EMAWeights(self.model.parameters(), ema_decay=cfg.opt.ema_decay)
        self.sample_time_eps = cfg.sde.sample_time_eps
        self.sample_N = cfg.sde.sample_N
        self.sample_mode = cfg.sde.sample_mode
        self.epoch, self.itr, self.time = 1, 0, 0.

    def score_fn(self, t, x, label=None, condition=None):
        t = t.to(x)
        params = self.model(x, t, label=label, condition=condition)
        from . import ops
        sde = self.SDE                                    # score = -params / sqrt(var(t)), one HIP kernel
        if sde.score_kind == 0:
            return ops.vpsde_score(params, t.float(), sde.beta_start, sde.beta_end, sde.sigma2_0), params
        return ops.sde_score(params, t.float(), sde.score_kind, *sde.score_consts()), params

    @torch.no_grad()
    def sample(self, num_samples, num_points=None, label=None, condition=None, *, x0=None, noise=None, seed=None,
               use_graph=None, trajectory=None):
        self.model.eval()
        self.compressor.eval()
        self.optimizer.swap_parameters_with_ema(store_params_in_ema=True)
        try:
            if self.sample_mode not in ("discrete", "continuous"):
                raise NotImplementedError("sample_mode %r" % (self.sample_mode,))
            cs = self.cfg.score
            if getattr(cs, "graphconv", False):
                # Latent_SDE_Trainer.py:158 samples (z_scale, z_dim + 3) latents under cfg.score.graphconv, which Score.ln_in —
                # Conv1d(z_dim -> hidden), score.py:110 — then rejects: no shipped YAML sets it; same failure, said plainly
                raise RuntimeError("cfg.score.graphconv=True: the sampler would draw latents with z_dim + 3 = %d channels, but "
                                   "Score.ln_in expects z_dim = %d (the reference fails in conv1d on the same mismatch)"
                                   % (cs.z_dim + 3, cs.z_dim))
            shape = (cs.z_scale, cs.z_dim)
            rank, ws = ldist.world()
            lo, hi, per = ldist.shard_bounds(num_samples, rank, ws)
            # every rank draws the FULL-batch x0 from its (identically seeded) CPU generator, then keeps its rows
            if x0 is None:
                x0 = torch.randn((num_samples,) + shape)
            if seed is None and noise is None and self.sample_mode == "discrete":
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())       # Philox key (only when device noise will be drawn)
            if ldist.initialized():
                ldist.check_same_draws(x0, seed, self.device)            # world-size invariance needs identical CPU generators
            x0_loc = _rows(x0, lo, hi, per)
            noise_loc = None if noise is None else _rows(noise.transpose(0, 1), lo, hi, per).transpose(0, 1)
            if ws > 1:                                   # per-sample conditioning follows its samples to their rank
                if torch.is_tensor(label):
                    label = _rows(label, lo, hi, per)
                if isinstance(condition, (tuple, list)):
                    condition = tuple(_rows(c, lo, hi, per) if torch.is_tensor(c) else c for c in condition)
                elif isinstance(condition, dict):        # raw ViPC inputs {'img','pts'}: ConditionNet runs on this rank's rows
                    condition = {k: _rows(v, lo, hi, per) if torch.is_tensor(v) else v for k, v in condition.items()}
            if self.sample_mode == "continuous":             # probability-flow ODE (Latent_SDE_Trainer.py:148-152)
                eps, self.nfe_count, _ = self.SDE.sample_model_ode(
                    score_fn=self.score_fn, num_samples=per, shape=(cs.z_scale, cs.z_dim), label=label, ode_eps=self.sample_time_eps,
                    enable_autocast=False, ode_solver_tol=self.cfg.sde.ode_tol, condition=condition, noise=x0_loc, device=self.device)
            else:
                eps = self.SDE.sample_discrete(score_fn=self.score_fn, N=self.cfg.sde.sample_N,
                                               corrector=self.cfg.sde.corrector, predictor=self.cfg.sde.predictor,
                                               corrector_steps=self.cfg.sde.corrector_steps, shape=shape,
                                               time_eps=self.sample_time_eps, label=label, denoise=self.cfg.sde.denoise,
                                               device=self.device, num_samples=per,
                                               probability_flow=self.cfg.sde.probability_flow, snr=self.cfg.sde.snr,
                                               condition=condition, x0=x0_loc, noise=noise_loc, sample_offset=lo,
                                               seed=seed, use_graph=use_graph, global_batch=num_samples if ws > 1 else None,
                                               trajectory=trajectory)
            npts = self.num_points if num_points is None else num_points
            sample = self.compressor.sample((per, npts), given_eps=eps)
            if ldist.initialized():                      # the single collective of the path
                sample = ldist.all_gather_rows(sample, num_samples)
                eps = ldist.all_gather_rows(eps, num_samples)
        finally:
            self.optimizer.swap_parameters_with_ema(store_params_in_ema=True)
        return sample, eps

    @torch.no_grad()
    def valsample(self, test_loader, val_cate=0, vis=False, *, batch_size=None, ref=None, save_npy=None):
        """The reference's validation sampling loop, same signature and return value (trainer/Latent_SDE_Trainer.py:167-226;
        called as `trainer.valsample(test_loader=test_loader, val_cate=13)` by train_Latent_Diffusion.py:60,85).

        `test_loader` is any iterable of the dataset's dict batches (`te_points`, `tr_points`, `cate_idx`):
          * `cfg.data.num_categorys == 1` (:173-188): one `sample(num_samples=len(batch['tr_points']))` per batch, the
            references are the batches' `te_points`;
          * otherwise (:189-205): the `te_points` whose `cate_idx == val_cate` are the references, and
            ceil(len(ref) / cfg.data.test_batch_size) label-conditioned batches are sampled and cut to len(ref).
            (Upstream appends `self.sample(...)`'s (points, eps) TUPLE there and would fail in `torch.cat`; the points are
            what is meant and what is kept here.)
        Then, as upstream: the "Sample rate" print (:206), the `smp_ep<epoch>.npy` dump into `cfg.log.save_path`
        (:207-210), `compute_all_metrics(smp, ref, batch_size=64)` (:217-220), the summary print, and the returned
        `{"val/gen/<key>": float}` dict.  `vis=True` (mitsuba rendering, :211-216) is out of scope and raises.

        Keyword extensions (not in the reference): `test_loader` may be an int = that many unconditional batches of
        `batch_size` (default `cfg.data.test_batch_size`) scored against `ref` (N, points, 3) when given; `save_npy`
        False suppresses the dump, None (default) dumps whenever `cfg.log.save_path` is set.  The samples and the rate of
        the last call stay available as `self.last_valsample = {"samples", "refs", "rate"}`."""
        import math
        import os
        import numpy as np
        if vis:
            raise NotImplementedError("valsample(vis=True): mitsuba rendering (tools/vis_utils.py) is not on this path")
        self.model.eval()
        self.compressor.eval()
        dev = self.device
        all_ref, all_smp, use_time = [], [], 0.

        def timed_sample(n, label=None):
            nonlocal use_time
            _sync(dev)
            t0 = time.time()
            pts, _ = self.sample(num_samples=n, label=label)
            _sync(dev)
            use_time += time.time() - t0
            return pts

        if isinstance(test_loader, int):                                   # extension: no dataset at hand
            bsize = batch_size or self.cfg.data.test_batch_size
            for _ in range(test_loader):
                all_smp.append(timed_sample(bsize))
            smp = torch.cat(all_smp, 0)
            if ref is not None:
                ref = ref.to(smp)
                smp = smp[:ref.shape[0]]
        elif self.cfg.data.num_categorys == 1:                            # :173-188
            for data in test_loader:
                all_ref.append(data["te_points"].to(dev))
                all_smp.append(timed_sample(data["tr_points"].size(0)))
            smp, ref = torch.cat(all_smp, 0), torch.cat(all_ref, 0).float()
        else:                                                              # :189-205
            for data in test_loader:
                idx = data["cate_idx"] == val_cate
                all_ref.append(data["te_points"][idx])
            ref = torch.cat(all_ref, 0).to(dev).float()
            bsize = self.cfg.data.test_batch_size
            for _ in range(math.ceil(ref.shape[0] / bsize)):
                cates = (torch.ones(bsize) * val_cate).int().to(dev)
                all_smp.append(timed_sample(bsize, label=cates))
            if not all_smp:
                raise ValueError("valsample: no test shape has cate_idx == %r" % (val_cate,))
            smp = torch.cat(all_smp, 0)[:ref.shape[0]]
        rate = smp.shape[0] / max(use_time, 1e-9)
        chief = ldist.world()[0] == 0
        if chief:
            print("Sample rate: %.8f " % rate)
        self.last_valsample = {"samples": smp, "refs": ref, "rate": rate}
        path = getattr(self.cfg.log, "save_path", "") or ""
        if save_npy and not path:
            raise ValueError("valsample(save_npy=True) needs cfg.log.save_path (upstream writes smp_ep<epoch>.npy there)")
        if chief and path and save_npy is not False:
            np.save(os.path.join(path, "smp_ep%d" % self.epoch + ".npy"), smp.detach().cpu().numpy())
        if ref is None:
            return {}
        from .metrics import compute_all_metrics
        gen_res = compute_all_metrics(smp, ref, batch_size=64)
        all_res = {("val/gen/%s" % k): (v if isinstance(v, float) else v.item()) for k, v in gen_res.items()}
        if chief:
            print("Validation Sample (unit) Epoch:%d " % self.epoch, gen_res)
        return all_res

    # ---- checkpoints: the reference's dict layout (:228-266) ------------------------------------------
    def resume(self, epoch=None, strict=False, load_optim=True, finetune=False, pretrain=None, **kwargs):
        """Same arguments as the reference (:241-266): the file is `pretrain` if given, else
        `<cfg.log.save_path>/checkpt_<epoch>.pth` with `epoch` defaulting to the last row of `training.csv`.
        (A path string passed as `epoch` is accepted as shorthand for `pretrain=`.)  Loads both state dicts, adopts
        the optimizer's per-parameter 'ema' tensors (unless finetune / load_optim=False) and re-runs
        `compressor.init()`."""
        import os
        if finetune:
            load_optim, strict = False, False
        if isinstance(epoch, (str, os.PathLike)):
            pretrain, epoch = epoch, None
        if pretrain is None:
            if epoch is None:
                import csv
                with open(os.path.join(self.cfg.log.save_path, "training.csv")) as f:
                    epoch = int(float(list(csv.DictReader(f))[-1]["epoch"]))
            pretrain = os.path.join(self.cfg.log.save_path, "checkpt_{:}.pth".format(epoch))
        ckpt = torch.load(pretrain, map_location="cpu", weights_only=False)   # holds cfg as argparse.Namespace
        self.model.load_state_dict(ckpt["score_state_dict"], strict=strict)
        self.compressor.load_state_dict(ckpt["compressor_state_dict"], strict=strict)
        self.compressor.init()
        if load_optim:
            self.optimizer.load_ema(ckpt["score_optim_state_dict"])
        if finetune:
            self.epoch, self.itr = 1, 0
        else:
            self.epoch, self.itr = ckpt["epoch"] + 1, ckpt["itr"]
        self.time = ckpt["time"]

    def load_pretrain(self, path=None):
        """Stage-1 compressor checkpoint (:268-273): key `state_dict`, strict."""
        path = self.cfg.compressor.pretrain_path if path is None else path
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        self.compressor.load_state_dict(ckpt["state_dict"], strict=True)
        self.compressor.init()


class CompletionTrainer(Trainer):
    """The sampling half of completion_trainer/Latent_SDE_Trainer.py (ShapeNet-ViPC completion, BASELINE configs[4]):
    `sample(num_samples, condition={'img': views, 'pts': partial})` runs the score model's ConditionNet once per call
    (:150-151; needs cfg.score.condition = True), samples image/partial-cloud conditioned latents and returns the decoded
    clouds only (:168); `valsample` is the evaluation loop of :170-215 without the dataset and the renderer."""

    @torch.no_grad()
    def sample(self, num_samples, num_points=None, label=None, condition=None, **kw):
        if isinstance(condition, dict):
            if not hasattr(self.model, "c_net"):
                raise ValueError("a raw condition dict needs cfg.score.condition=True (ConditionNet, score.py:64-65)")
            condition = self.model.c_net({k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in condition.items()})
        return super().sample(num_samples, num_points=num_points, label=label, condition=condition, **kw)[0]

    @torch.no_grad()
    def valsample(self, test_loader, vis=False, full=False, save_npy=None):
        """test_loader yields (views (B,3,H,W), pc (B,N,3), pc_part (B,Np,3)); both clouds are reduced to 2048 points by
        farthest point sampling (:181-184), the partial cloud + views condition the sampler, and L2_ChamferEval_1000 /
        F1Score over everything sampled are reported as upstream (:197-201).  full=False stops once more than 1000 shapes
        are accumulated (:203-205).  The `part_ep / smp_ep / ref_ep<epoch>.npy` dumps of :216-227 are written when
        cfg.log.save_path is set (save_npy=None) or on request (save_npy=True).  vis=True (mitsuba rendering, :208-213) is
        out of scope and raises.  Returns {'cd', 'f1', 'f1score', 'rate', 'samples', 'refs', 'parts'}."""
        import os
        import numpy as np
        from . import ops
        from .metrics import F1Score, L2_ChamferEval_1000
        if vis:
            raise NotImplementedError("valsample(vis=True): mitsuba rendering (tools/vis_utils.py) is not on this path")
        self.model.eval(); self.compressor.eval()
        all_ref, all_smp, all_part, use_time, count = [], [], [], 0., 0
        for views, pc, pc_part in test_loader:
            pc, pc_part = pc.to(self.device).float().contiguous(), pc_part.to(self.device).float().contiguous()
            ref_pts = ops.gather_rows(pc, ops.fps(pc, min(2048, pc.shape[1])))
            part = ops.gather_rows(pc_part, ops.fps(pc_part, min(2048, pc_part.shape[1])))
            torch.cuda.synchronize()
            t0 = time.time()
            smp = self.sample(num_samples=ref_pts.size(0), condition={"img": views.float(), "pts": part})
            torch.cuda.synchronize()
            use_time += time.time() - t0
            all_smp.append(smp); all_ref.append(ref_pts); all_part.append(part)
            count += smp.shape[0]
            if not full and count > 1000:                                             # :203-205
                break
        smp, ref, part = torch.cat(all_smp, 0), torch.cat(all_ref, 0), torch.cat(all_part, 0)
        cd = L2_ChamferEval_1000(smp, ref)
        f1, _, _ = F1Score(smp, ref)
        path = getattr(self.cfg.log, "save_path", "")
        if save_npy or (save_npy is None and path):
            for tag, t in (("part", part), ("smp", smp), ("ref", ref)):
                np.save(os.path.join(path, "%s_ep%d.npy" % (tag, self.epoch)), t.detach().cpu().numpy())
        print("Validation Sample (unit) Epoch:%d " % self.epoch, {"cd": float(cd), "f1score": float(f1.mean())})
        return {"cd": float(cd), "f1": float(f1.mean()), "f1score": float(f1.mean()), "rate": smp.shape[0] / max(use_time, 1e-9),
                "samples": smp, "refs": ref, "parts": part}


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def _rows(t, lo, hi, per):
    """Rows [lo,hi) of a full-batch tensor, zero-padded to `per` rows (last rank when B % world != 0)."""
    part = t[lo:min(hi, t.shape[0])]
    if part.shape[0] < per:
        pad = torch.zeros((per - part.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        part = torch.cat([part, pad], 0)
    return part
