"""The SDE families and the discrete reverse-SDE sampler (reference: diffusion/diffusion_continuous.py).

`DiffusionBase` holds the samplers, written against the schedule (`f, g2, var, e2int_f`) a family supplies: `DiffusionVPSDE` (:626-678;
also `betas, alpha, alphas_cump, N` for the ancestral / ddim / PNDM predictors), `DiffusionSubVPSDE` (:681-729), `DiffusionVESDE`
(:732-766), `DiffusionGeometric` (:595-623); `make_diffusion` is upstream's factory (:18-29).  Every class keeps the reference
constructor (`args` = cfg.sde Namespace) and `sample_discrete(...)` (:133-338).
Schedule scalars are host-side fp32 tables computed with the reference's own op order (SURVEY hard
part 4: var(1e-6) is exactly one ulp in fp32 and must not be recomputed with a different expf) and uploaded.

`sample_discrete` has the reference signature.  Two execution paths, both on HIP kernels:
  * fused loop — when `score_fn` is the bound `Trainer.score_fn` of an `ldt_amd.Score` (unconditional):
    one `ldt_sample_loop` call = AdaLN table for all N steps, then N x (Score forward -> fused predictor
    update with in-kernel Philox noise -> ++step), optionally replayed from one captured hipGraph;
  * generic loop — any other `score_fn(t, x, label=, condition=) -> (score, params)` callable is driven
    step by step from Python, the update still done by `ldt_sampler_step` (params, not score, feed it).
Extra keyword-only arguments (not in the reference): `x0`, `noise` (inject the CPU draws for parity),
`sample_offset` (global index of this rank's first sample, keeps Philox streams shard-invariant), `streams` (sub-batches
sampled concurrently on their own HIP streams; default 1, `LDT_STREAMS=2` measured -2.6 % per step at B=64, T=256).
"""
import ctypes
import os

import numpy as np
import torch

from . import ops
from ._lib import check, lib

PREDICTORS = ("reversediffusion", "ancestral", "eulermaruyama", "ddim")


def make_diffusion(args):
    """diffusion_continuous.py:18-29."""
    if args.sde_type == "geometric_sde":
        return DiffusionGeometric(args)
    if args.sde_type == "vpsde":
        return DiffusionVPSDE(args)
    if args.sde_type == "sub_vpsde":
        return DiffusionSubVPSDE(args)
    if args.sde_type == "vesde":
        return DiffusionVESDE(args)
    raise ValueError("Unrecognized sde type: {}".format(args.sde_type))


class DiffusionBase:
    """diffusion_continuous.py:32-624: the samplers, written against the schedule a subclass supplies (f, g2, var, e2int_f).
    `score_kind` / `score_consts()` name the subclass's var(t) for ldt_sde_score (Trainer.score_fn)."""
    score_kind = None

    def __init__(self, args):
        self.sigma2_0 = args.sigma2_0
        self.sde_type = args.sde_type
        self.time_eps = args.time_eps
        self.sample_time_eps = args.sample_time_eps

    def f(self, t):
        raise NotImplementedError

    def g2(self, t):
        raise NotImplementedError

    def var(self, t):
        raise NotImplementedError

    def e2int_f(self, t):
        raise NotImplementedError

    def std(self, t):
        return torch.sqrt(self.var(t))

    # ---- per-step coefficient table for ldt_sampler_step ------------------------------------------
    def step_table(self, N, predictor, time_eps, probability_flow=False):
        """-> (timesteps [N] fp32 CPU, coef [N,4] fp32 CPU, mode).  All scalars are produced on the CPU with
        the reference's expressions (fp32 tensors), the folded forms in fp64 then rounded once."""
        if predictor not in PREDICTORS:
            raise NotImplementedError("preditor not Implemented")           # diffusion_continuous.py:328
        ts = torch.linspace(1.0, time_eps, N)                               # :238
        if predictor == "ancestral":                                        # :152-162 — exact op order on device
            idx = (ts * (N - 1) / 1.0).long()
            beta = self.betas[idx]
            coef = torch.stack([beta, self.std(ts), torch.sqrt(1. - beta), torch.sqrt(beta)], 1)
            return ts, coef.contiguous(), 0
        std = self.std(ts).double()
        if predictor == "reversediffusion":                                 # :141-150
            dt = (1 - time_eps) / N
            g2 = self.g2(ts).double(); ff = self.f(ts).double()     # fp32 like upstream, then folded in fp64
            k = 0.5 if probability_flow else 1.0
            A = 1 - ff * dt
            Bc = -g2 * k * dt / std                # x_mean = x - (f x - g2 k score) dt, score = -params/std
            Cc = torch.zeros_like(g2) if probability_flow else torch.sqrt(g2) * np.sqrt(dt)
        elif predictor == "eulermaruyama":                                  # :182-191
            dt = -1.0 / N
            g2 = self.g2(ts).double(); ff = self.f(ts).double()     # fp32 like upstream, then folded in fp64
            k = 0.5 if probability_flow else 1.0
            A = 1 + ff * dt
            Bc = g2 * k * dt / std
            Cc = torch.zeros_like(g2) if probability_flow else torch.sqrt(g2) * np.sqrt(-dt)
        else:                                                               # ddim :164-180 (sigma = 0)
            idx = (ts * (N - 1) / 1.0).long()
            at = self.alphas_cump[idx].double()
            at_next = torch.where(idx - 1 < 0, torch.ones_like(at), self.alphas_cump[(idx - 1).clamp_min(0)].double())
            A = at_next.sqrt() / at.sqrt()
            Bc = -at_next.sqrt() * (1 - at).sqrt() / at.sqrt() + (1 - at_next).sqrt()
            Cc = torch.zeros_like(at)
        coef = torch.stack([A, Bc, Cc, torch.zeros_like(A)], 1).float()
        return ts, coef.contiguous(), 1

    # ---- probability-flow ODE sampling (sample_mode: continuous) -----------------------------------------
    @torch.no_grad()
    def sample_model_ode(self, score_fn, num_samples, shape, ode_eps, ode_solver_tol, enable_autocast=False, noise=None,
                         condition=None, label=None, *, device="cuda"):
        """diffusion_continuous.py:88-131: dx/dt = f(t) x - g2(t)/2 * score, from t = 1 to `ode_eps`, solved the way the
        reference's `torchdiffeq.odeint(method="scipy_solver", options={"solver": "RK45"})` does it: scipy's RK45 on the
        host over the flattened float64 state, time reversed to increasing s = -t, rtol = atol = `ode_solver_tol`; every
        function evaluation is one Score forward on the GPU (`score_fn(t, x)` with t a (B,) vector, as upstream).
        Returns (samples, nfe_count, seconds) like the reference.  `noise` defaults to a CPU-generator draw (the reference
        draws on the CUDA generator, :107 — not reproducible across devices either way).
        NB the bf16 Score limits the smoothness the step controller sees to ~1e-3 relative: tolerances much below that
        (the shipped `ode_tol: 1e-5`) are met only by many small steps.  torchdiffeq is not vendored upstream: its wrapper's
        behaviour is restated from its published semantics — parity unpinned."""
        import time
        from scipy.integrate import solve_ivp
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("sample_model_ode: device %s — the HIP path has no CPU fallback" % (device,))
        x0 = torch.randn((num_samples,) + tuple(shape)) if noise is None else noise
        full = tuple(x0.shape)
        nfe = [0]

        def fun(s, y):
            t = torch.full((full[0],), -float(s), dtype=torch.float32, device=dev)
            x = torch.from_numpy(y).to(dev, torch.float32).reshape(full)
            score, _ = score_fn(t, x, label=label, condition=condition)
            dx = self.f(t)[:, None, None] * x - 0.5 * self.g2(t)[:, None, None] * score
            nfe[0] += 1
            return (-dx).reshape(-1).double().cpu().numpy()

        t0 = time.time()
        sol = solve_ivp(fun, t_span=[-1.0, -float(ode_eps)], y0=x0.reshape(-1).double().cpu().numpy(),
                        t_eval=[-1.0, -float(ode_eps)], method="RK45", rtol=ode_solver_tol, atol=ode_solver_tol)
        out = torch.from_numpy(sol.y[:, -1]).to(dev, torch.float32).reshape(full)
        return out, nfe[0], time.time() - t0

    # ---- the sampler --------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_discrete(self, score_fn, num_samples, N, predictor, corrector, corrector_steps, shape, time_eps,
                        probability_flow, denoise, snr, device, condition=None, label=None, print_steps=None,
                        *, x0=None, noise=None, sample_offset=0, seed=None, use_graph=None, record=None, streams=None, global_batch=None,
                        trajectory=None):
        """Reverse-SDE predictor(-corrector) sampling, diffusion_continuous.py:133-338.

        corrector: None, 'ancestral' (AncestralCorrector :212-229; alpha = 1 by the reference's quirk Q11) or 'langevin'
        (LangevinCorrector :193-210).  predictor 'pndm' runs PNDM (:260-316; corrector / noise are not used by it).
        Langevin and PNDM multiply a (B,1) factor into (B,tokens,z) latents: the reference only runs them when B == 1 or
        B == tokens (the factor is batch-uniform, so the result is well defined then) and raises torch's broadcasting
        error otherwise; the same rule is applied here to the GLOBAL batch (`global_batch`, default num_samples).
        Langevin's step size uses batch means of norms — under a sharded batch the two norm sums are all-reduced (the one
        cross-sample quantity on the path); everything else stays per rank.
        print_steps: the trajectory dump of :239-257 (returns the list of tensors).
        trajectory: a list; the fused loop appends ONE tensor [N, B, tokens, z] holding x after every step (parity curves)."""
        if corrector not in (None, "ancestral", "langevin"):
            raise NotImplementedError("corrector not Implemented")           # diffusion_continuous.py:335
        gb = num_samples if global_batch is None else int(global_batch)
        if predictor == "pndm" or corrector == "langevin":
            if not (gb == 1 or gb == shape[0]) or shape[0] == 1 and gb != 1:
                raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1"
                                   % (gb, shape[0]))                         # what torch raises at :208 / :269
        if predictor == "pndm":
            return self._sample_pndm(score_fn, num_samples, shape, time_eps, device, condition, label, x0)
        ts, coef, mode = self.step_table(N, predictor, time_eps, probability_flow)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("sample_discrete: device %s — the HIP path has no CPU fallback" % (device,))
        # initial sample: CPU generator then copy, exactly like the reference (:237)
        x = torch.randn((num_samples,) + tuple(shape)) if x0 is None else x0
        x = x.to(dev, torch.float32).contiguous().clone()
        if seed is None:                                       # Philox key from the (seedable) CPU generator; not drawn (the
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise is None else 0   # reference's stream is kept) when noise is injected
        ncs = corrector_steps if corrector is not None else 0  # noise draws per step: 1 predictor + ncs corrector
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
            assert noise.shape == (N * (1 + ncs),) + tuple(x.shape), "noise must be [N*(1+corrector_steps), B, tokens, z]"
        coef_d = coef.to(dev)
        elem_offset = int(sample_offset) * int(np.prod(shape))
        nstride = x.numel() if noise is not None else 0
        x_mean = torch.empty_like(x)
        model = _fused_model(score_fn)
        if model is not None and record is None and corrector is None and print_steps is None:
            from ._lib import CondArgs
            B, T = x.shape[0], x.shape[1]
            if use_graph is None:
                use_graph = B * T <= 4096                                             # launch-bound regime only
            # Sub-batches on concurrent streams (trajectories are independent): with the persistent GEMM grids capped at
            # half the chip each, one stream's HBM-bound phases (epilogues, attention) run beside the other's MFMA-bound
            # main loops instead of all 256 CUs alternating between the two (measured -2.6 % per SDE step at B=64, T=256;
            # a deliberate phase skew between the streams gained nothing, four quarter batches lose: too few tiles).  Only
            # where a half batch still fills its half of the chip with whole 256x256 tiles.
            if streams is None:                                                       # opt-in: per-kernel accounting (bench roofline,
                streams = int(os.environ.get("LDT_STREAMS", "1"))                     # rocprof) stays one launch = the whole batch
            streams = max(1, min(int(streams), B))
            shared = condition is None and label is None
            if shared:
                _, mod = model.time_table(ts.to(dev))                                 # AdaLN rows for every step, shared by the batch
            else:
                extra, kv, S = model.condition_embedding(label, condition)            # per-sample rows, rebuilt every step
                temb = model.time_embedding(ts.to(dev))
                panel = model.stacked_adaln_bf16() if int(os.environ.get("LDT_ADALN_BF16", "1")) else None
                if panel is not None:                                                 # bf16 weight panel (t_dim % 64 == 0); no fp32 copy kept
                    (w_ada_bf, b_ada), w_ada = panel, None
                else:
                    (w_ada, b_ada), w_ada_bf = model.stacked_adaln(), None
            bounds = [B * i // streams for i in range(streams + 1)]
            jobs, keep = [], []
            fold = None
            fold_mon = None                                                           # running max of mean^2 / variance over the monitored steps
            for i in range(streams):
                lo, hi = bounds[i], bounds[i + 1]
                Bs = hi - lo
                xs, xm = x[lo:hi], x_mean[lo:hi]                                      # contiguous row slices, updated in place
                eps_tmp = torch.empty_like(xs)
                tj = None if trajectory is None else torch.empty((N,) + tuple(xs.shape), dtype=torch.float32, device=dev)
                counter = torch.zeros(1, dtype=torch.int32, device=dev)
                nz = None if noise is None else noise[:, lo:hi]
                wgs = 0 if streams == 1 else max(256 // streams, 1)
                cond_ref = None
                if shared:
                    if fold is None and model.can_fold(Bs, T, wgs):
                        fold = model.fold_table(mod)                                  # (+ the LN-folding S / C rows)
                        # guard: a monitored forward on the first sub-batch at step 0 (0.1 % of a 1000-step call); rows whose
                        # mean dwarfs their spread switch this model to the LayerNorm kernels (Score.fold_probe)
                        model.fold_probe(xs, 0, mod, fold)
                    folding = model.can_fold(Bs, T, wgs)                              # (the probe may just have switched it off)
                    if folding and fold_mon is None:
                        fold_mon = torch.zeros(1, dtype=torch.float32, device=dev)
                    # inside the loop the same monitor runs every LDT_FOLD_MONITOR_EVERY steps (default 50: 47 launches of ~5 us per monitored
                    # step = 0.05 % of a call) and on the last step; the running maximum is read at the NEXT call's first can_fold() (round 6: a trajectory
                    # that passes the bound mid-way is seen, not just its two ends)
                    plan = model.plan(Bs, T, mod, model.n_mod, 0, fold=fold if folding else None, slot=i, gemm_wgs=wgs,
                                      monitor=fold_mon if folding else None, monitor_every=int(os.environ.get("LDT_FOLD_MONITOR_EVERY", "50")))
                else:
                    c_buf = torch.empty((Bs, model.t_dim), dtype=torch.float32, device=dev)
                    modb = torch.empty((Bs, model.n_mod), dtype=torch.float32, device=dev)
                    ex = None if extra is None else extra[lo:hi].contiguous()
                    kvs = None if kv is None else {l: t.view(B, S, -1)[lo:hi].reshape(Bs * S, -1) for l, t in kv.items()}
                    plan = model.plan(Bs, T, modb, 0, model.n_mod, kv_cond=kvs, cond_tokens=S, slot=i, gemm_wgs=wgs)
                    c_bf = None if w_ada_bf is None else torch.empty((Bs, model.t_dim), dtype=torch.bfloat16, device=dev)
                    cond = CondArgs(temb.data_ptr(), ops._p(ex), ops._p(w_ada), b_ada.data_ptr(), c_buf.data_ptr(),
                                    modb.data_ptr(), model.t_dim, model.n_mod, ops._p(w_ada_bf), ops._p(c_bf))
                    cond_ref = ctypes.byref(cond)
                    keep.append((ex, kvs, c_buf, modb, cond, c_bf, w_ada_bf, w_ada, b_ada))
                if nz is not None and streams > 1:
                    nz = nz.contiguous()                                              # [N, Bs, T, z] with step stride Bs*T*z
                jobs.append((plan, xs, xm, eps_tmp, counter, nz, cond_ref, elem_offset + lo * int(np.prod(shape)), tj))

            def run(job, stream):
                plan, xs, xm, eps_tmp, counter, nz, cond_ref, off, tj = job
                with torch.cuda.stream(stream):
                    check(lib().ldt_sample_loop(ctypes.byref(plan), xs.data_ptr(), xm.data_ptr(), eps_tmp.data_ptr(),
                                                coef_d.data_ptr(), mode, ops._p(nz), xs.numel() if nz is not None else 0, off, seed,
                                                counter.data_ptr(), N, cond_ref, ops._p(tj), int(bool(use_graph)), ops.stream_ptr()),
                          "ldt_sample_loop")

            main = torch.cuda.current_stream()
            if streams == 1:
                run(jobs[0], main)
            else:
                import threading
                subs = [torch.cuda.Stream(device=dev) for _ in jobs]
                errs = []

                def worker(job, st):
                    try:
                        torch.cuda.set_device(dev)
                        st.wait_stream(main)                                          # tables / x0 were produced on the caller's stream
                        run(job, st)
                    except BaseException as e:                                        # noqa: BLE001 - re-raised on the caller's thread
                        errs.append(e)

                th = [threading.Thread(target=worker, args=(j, st)) for j, st in zip(jobs, subs)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()                                                         # (ctypes releases the GIL inside the C call)
                if errs:
                    raise errs[0]
                for st in subs:
                    main.wait_stream(st)
            if keep or streams > 1:
                main.synchronize()                                                    # scratch / sub-streams must outlive the loop
            if trajectory is not None:
                trajectory.append(torch.cat([j[-1] for j in jobs], 1))
            if fold_mon is not None:
                model.defer_fold_ratio(fold_mon)                                      # read (a device sync) at the next can_fold(): affects later calls
            return x_mean if denoise else x
        # ---- generic loop: any score_fn, correctors, trajectory dumps; every update is still one HIP kernel ----
        ts_d = ts.to(dev)
        if corrector == "ancestral":                           # folded AncestralCorrector: x_mean = x - 2 snr^2 std params,
            std = self.std(ts).double()                        # x = x_mean + 2 snr std z   (score = -params / std)
            ccoef = torch.stack([torch.ones_like(std), -2.0 * snr * snr * std, 2.0 * snr * std, torch.zeros_like(std)], 1)
            ccoef_d = ccoef.float().contiguous().to(dev)
        elif corrector == "langevin":                          # step size from batch-mean norms, formed on the device per draw
            std_host = self.std(ts)
            lv_coef = torch.empty(4, dtype=torch.float32, device=dev)
            lv_sums = torch.empty(2, dtype=torch.float32, device=dev)
            lv_norms = torch.empty(num_samples, dtype=torch.float32, device=dev)
            per = int(np.prod(shape))
            n_valid = max(0, min(num_samples, gb - int(sample_offset)))
        out_list, every = None, None
        traj_list = [] if trajectory is not None else None
        if print_steps is not None:
            out_list, every = [x.clone()], (N - 1) // (print_steps - 2)
        params_of = _params_fn(score_fn)
        for i in range(N):
            vec_t = torch.ones((num_samples,), device=dev) * ts_d[i]                 # :243-244
            params = params_of(vec_t, x, label=label, condition=condition)
            k = i * (1 + ncs)                                  # index of this step's first draw (noise row / Philox stream id)
            x_new = ops.sampler_step(x, params.contiguous(), coef_d, i, mode, noise=None if noise is None else noise[k],
                                     x_mean_out=x_mean, elem_offset=elem_offset, seed=seed, philox_mul=1 + ncs)
            if record is not None:
                record.append((x, params, x_mean.clone(), x_new))
            x = x_new
            for j in range(ncs):
                params = params_of(vec_t, x, label=label, condition=condition).contiguous()
                if corrector == "ancestral":                                         # AncestralCorrector :212-229
                    x = ops.sampler_step(x, params, ccoef_d, i, 1, noise=None if noise is None else noise[k + 1 + j],
                                         x_mean_out=x_mean, elem_offset=elem_offset, seed=seed, philox_mul=1 + ncs,
                                         philox_add=1 + j)
                    continue
                # LangevinCorrector :193-210
                z = noise[k + 1 + j] if noise is not None else \
                    ops.philox_normal(x.shape, dev, seed, step=i * (1 + ncs) + 1 + j, elem_offset=elem_offset)
                x = langevin_update(x, params, z, x_mean, float(std_host[i]), snr, gb, n_valid, global_batch is not None,
                                    (lv_coef, lv_sums, lv_norms))
            if traj_list is not None:
                traj_list.append(x.clone())
            if out_list is not None and (i + 1) % every == 0:
                out_list.append(x_mean.clone())
        if traj_list is not None:
            trajectory.append(torch.stack(traj_list))
        if out_list is not None:
            out_list.append((x_mean if denoise else x).clone())
            return out_list
        return x_mean if denoise else x


    @torch.no_grad()
    def _sample_pndm(self, score_fn, num_samples, shape, time_eps, device, condition, label, x0):
        """PNDM (diffusion_continuous.py:260-316) around the HIP Score: the pseudo linear multistep sampler with three
        Runge-Kutta warm-up steps (4 Score evaluations each), self.N steps over self.train_N training levels.  Every
        sample sits at the same t, so the (B,1) schedule factors of transfer() (:267-271) are three scalars per call,
        formed here in fp32 exactly as upstream; the element-wise updates are HIP kernels.  Index quirk kept: at the last
        step `timesteps[t_next*2 - 1]` with t_next = 0 reads timesteps[-1] = 1.0 (:307)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("sample_discrete: device %s — the HIP path has no CPU fallback" % (device,))
        N, train_N = self.N, self.train_N
        x = (torch.randn((num_samples,) + tuple(shape)) if x0 is None else x0).to(dev, torch.float32).contiguous().clone()
        betas = torch.from_numpy(np.linspace(self.beta_start / train_N, self.beta_end / train_N, train_N,
                                             dtype=np.float64)).to(torch.float32)                    # :310-313
        alphas_cump = torch.cat((torch.ones(1), (1.0 - betas).cumprod(dim=0)))                     # :314-315 (train_N + 1)
        timesteps = torch.linspace(time_eps, 1.0, N * 2)                                             # :262
        st = ops.stream_ptr

        def level(t):                                            # :264-265
            return int((train_N * (t - time_eps) + 1).long())

        def transfer(xx, t, t_next, et):                         # :263-274
            at, at_next = alphas_cump[level(t)], alphas_cump[level(t_next)]
            d = at_next - at
            p = 1 / (at.sqrt() * (at.sqrt() + at_next.sqrt()))
            q = 1 / (at.sqrt() * (((1 - at_next) * at).sqrt() + ((1 - at) * at_next).sqrt()))
            out = torch.empty_like(xx)
            check(lib().ldt_pndm_transfer(xx.data_ptr(), et.data_ptr(), float(d), float(p), float(q), out.data_ptr(), xx.numel(), st()),
                  "ldt_pndm_transfer")
            return out

        def eps_at(t, xx):
            vec_t = (torch.ones((num_samples,)) * t).to(dev)
            return params_of(vec_t, xx, condition=condition, label=label).contiguous()

        def lincomb(a, c, scale):
            out = torch.empty_like(a[0])
            check(lib().ldt_lincomb4(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), c[0], c[1], c[2], c[3],
                                     scale, out.data_ptr(), out.numel(), st()), "ldt_lincomb4")
            return out

        params_of = _params_fn(score_fn)
        ets = []
        for idx in range(N, 0, -1):                              # :316-317
            t_next = idx - 1
            if len(ets) > 2:                                     # :296-300 linear multistep
                ets.append(eps_at(timesteps[idx * 2 - 1], x))
                ets = ets[-4:]
                noise = lincomb((ets[-1], ets[-2], ets[-3], ets[-4]), (55.0, -59.0, 37.0, -9.0), 1 / 24)
            else:                                                # :276-292 Runge-Kutta warm-up
                t1, t2, t3 = timesteps[idx * 2 - 1], timesteps[int((idx + t_next) / 2 * 2) - 1], timesteps[int(t_next * 2) - 1]
                e1 = eps_at(t1, x)
                ets.append(e1)
                e2 = eps_at(t2, transfer(x, t1, t2, e1))
                e3 = eps_at(t2, transfer(x, t1, t2, e2))
                e4 = eps_at(t3, transfer(x, t1, t3, e3))
                noise = lincomb((e1, e2, e3, e4), (1.0, 2.0, 2.0, 1.0), 1 / 6)
            x = transfer(x, timesteps[idx * 2 - 1], timesteps[t_next * 2 - 1], noise)               # :304-307
        return x


class DiffusionVPSDE(DiffusionBase):
    """diffusion_continuous.py:626-678: dz = -beta(t)/2 z dt + sqrt(beta(t)) dW, linear beta(t)."""
    score_kind = 0

    def __init__(self, args):
        super().__init__(args)
        self.beta_start = args.beta_start
        self.beta_end = args.beta_end
        self.train_N = args.train_N
        if args.sample_mode == "discrete":
            self.N = args.sample_N
            self.betas = torch.from_numpy(np.linspace(self.beta_start / self.N, self.beta_end / self.N, self.N,
                                                      dtype=np.float64)).to(torch.float32)
            self.alpha = 1.0 - self.betas
            self.alphas_cump = self.alpha.cumprod(dim=0)

    def score_consts(self):
        return self.beta_start, self.beta_end, self.sigma2_0

    # ---- schedule (tensor in, tensor out; any device) ------------------------------------------
    def g2(self, t):
        return self.beta_start + (self.beta_end - self.beta_start) * t

    def f(self, t):
        return -0.5 * self.g2(t)

    def var(self, t):
        return 1.0 - (1.0 - self.sigma2_0) * torch.exp(
            -self.beta_start * t - 0.5 * (self.beta_end - self.beta_start) * t * t)

    def e2int_f(self, t):
        return torch.exp(-0.5 * self.beta_start * t - 0.25 * (self.beta_end - self.beta_start) * t * t)

    def discrete(self, idx):
        return self.betas.index_select(0, idx), self.alpha.index_select(0, idx)


class DiffusionSubVPSDE(DiffusionBase):
    """diffusion_continuous.py:681-729: the sub-VP SDE (same drift as the VP-SDE, g2 = beta (1 - exp(-2 int beta))).  It has
    no `betas` table upstream either: the 'ancestral' / 'ddim' predictors raise the reference's AttributeError."""
    score_kind = 1

    def __init__(self, args):
        super().__init__(args)
        self.beta_start = args.beta_start
        self.beta_end = args.beta_end

    def score_consts(self):
        return self.beta_start, self.beta_end, self.sigma2_0

    def beta(self, t):
        return self.beta_start + (self.beta_end - self.beta_start) * t

    def f(self, t):
        return -0.5 * self.beta(t)

    def g2(self, t):
        return self.beta(t) * (1.0 - torch.exp(-2.0 * self.beta_start * t - (self.beta_end - self.beta_start) * t * t))

    def var(self, t):
        int_term = torch.exp(-self.beta_start * t - 0.5 * (self.beta_end - self.beta_start) * t * t)
        return torch.square(1.0 - int_term) + self.sigma2_0 * int_term

    def e2int_f(self, t):
        return torch.exp(-0.5 * self.beta_start * t - 0.25 * (self.beta_end - self.beta_start) * t * t)


class _GeometricVariance(DiffusionBase):
    """var(t) = sigma2_min (sigma2_max / sigma2_min)^t - sigma2_min + sigma2_0, shared by the VE and the geometric SDE."""
    score_kind = 2

    def __init__(self, args):
        super().__init__(args)
        self.sigma2_min = args.sigma2_min
        self.sigma2_max = args.sigma2_max

    def score_consts(self):
        return self.sigma2_min, self.sigma2_max / self.sigma2_min, self.sigma2_0

    def var(self, t):
        return self.sigma2_min * ((self.sigma2_max / self.sigma2_min) ** t) - self.sigma2_min + self.sigma2_0


class DiffusionVESDE(_GeometricVariance):
    """diffusion_continuous.py:732-766: dz = sqrt(beta(t)) dW."""

    def __init__(self, args):
        super().__init__(args)
        assert self.sigma2_min == self.sigma2_0, "VESDE was proposed implicitly assuming sigma2_min = sigma2_0!"

    def f(self, t):
        return torch.zeros_like(t)

    def g2(self, t):
        return self.sigma2_min * np.log(self.sigma2_max / self.sigma2_min) * ((self.sigma2_max / self.sigma2_min) ** t)

    def e2int_f(self, t):
        return torch.ones_like(t)


class DiffusionGeometric(_GeometricVariance):
    """diffusion_continuous.py:595-623: the VP drift with a geometric progression of the variance."""

    def f(self, t):
        return -0.5 * self.g2(t)

    def g2(self, t):
        sigma2_geom = self.sigma2_min * ((self.sigma2_max / self.sigma2_min) ** t)
        log_term = np.log(self.sigma2_max / self.sigma2_min)
        return sigma2_geom * log_term / (1.0 - self.sigma2_0 + self.sigma2_min - sigma2_geom)

    def e2int_f(self, t):
        return torch.sqrt(
            1.0 + self.sigma2_min * (1.0 - (self.sigma2_max / self.sigma2_min) ** t) / (1.0 - self.sigma2_0))


def langevin_update(x, params, z, x_mean, std_t, snr, n_total, n_valid, sharded, scratch):
    """One LangevinCorrector update (diffusion_continuous.py:193-210) of this rank's rows.  The step size uses BATCH means of
    per-sample norms — the path's only cross-sample quantity: the two sums are formed over this rank's `n_valid` real rows
    (the zero-padding rows of the last rank do not count; a rank may hold none), all-reduced when the batch is sharded, and
    divided by the GLOBAL batch `n_total`.  scratch = (coef[4], sums[2], norms[rows]) device tensors.  Returns the new x
    (x_mean is written in place)."""
    lv_coef, lv_sums, lv_norms = scratch
    per = x[0].numel()
    if n_valid > 0:
        ops.batch_norm_sum(params, n_valid, per, lv_norms, lv_sums)
        ops.batch_norm_sum(z, n_valid, per, lv_norms, lv_sums[1:])
    else:
        lv_sums.zero_()
    if sharded:
        from . import dist as ldist
        ldist.all_reduce_sum_(lv_sums)
    ops.langevin_coef(lv_sums, n_total, snr, std_t, lv_coef)
    return ops.sampler_step(x, params, lv_coef, 0, 1, noise=z, x_mean_out=x_mean)


def _is_stock_score_fn(score_fn):
    """True for the bound `ldt_amd.Trainer.score_fn` itself (a subclass that overrides it is driven as an opaque callable)."""
    from .trainer import Trainer
    return getattr(score_fn, "__func__", None) is Trainer.score_fn


def _params_fn(score_fn):
    """The samplers only consume `params` (the eps prediction) of `score_fn`'s (score, params) pair.  For the bound,
    un-overridden `Trainer.score_fn` the model is called directly, so the score (-params / std, one more pass over the
    latents per evaluation) is not formed just to be dropped; any other callable is evaluated as given."""
    owner = getattr(score_fn, "__self__", None)
    model = getattr(owner, "model", None)
    if model is not None and _is_stock_score_fn(score_fn):
        return lambda t, x, label=None, condition=None: model(x, t.to(x), label=label, condition=condition)
    return lambda t, x, label=None, condition=None: score_fn(t, x, label=label, condition=condition)[1]


def _fused_model(score_fn):
    """The `ldt_amd.Score` behind a bound `Trainer.score_fn`, else None."""
    from .score import Score
    owner = getattr(score_fn, "__self__", None)
    model = getattr(owner, "model", None)
    if isinstance(model, Score) and not model.host_blocks and _is_stock_score_fn(score_fn):
        return model                                            # (the U-Net / non-LayerNorm variants are driven by the generic loop)
    return None
