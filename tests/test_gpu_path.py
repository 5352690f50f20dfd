"""GPU (-m gpu): the drop-in classes (Score, DiffusionVPSDE, Compressor, Trainer) through the C-ABI vs
(a) golden vectors captured from the reference and (b) the CPU oracle on seeded inputs.

Stated bf16 tolerances (north-star: per-step MSE + final Chamfer), relative MSE = ||a-b||^2/||b||^2:
  * teacher-forced Score output `params`                                      <= 1e-4   (measured ~1e-6)
  * free-running latents, every recorded step and the final x_mean            <= 1e-4   (measured 2.6e-6)
  * decoder on N(0,1)-scale latents (its trained operating range)             <= 1e-4
  * END-TO-END decoded cloud and Chamfer distance with weights TRAINED by the reference's own Trainer.update
    (tests/golden/trained_tiny.npz; latents at the data scale): points <= 1e-3, CD / mean squared radius <= 1e-3, FIXED
  * the same with RANDOM weights (a robustness case, not the Chamfer check): self-calibrated.  With random (untrained)
    weights the reverse SDE inflates the latents to rms ~600 (prod 1/sqrt(1-beta_i) = e^5), where the decoder's
    softmaxes saturate and the map is ill-conditioned: the fp32 CPU oracle itself moves by ~7e-3 rel-MSE when
    x0 is perturbed by ONE bf16 rounding (2^-9 relative).  The bar is therefore
        err_gpu <= max(2e-3, 4 x err_oracle(x0 * (1 + 2^-9 * N(0,1)))),  same bar for CD / mean squared radius.
The fp32 pieces (AdaLN tables, sampler update) are held to fp32 round-off in test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_mse

pytestmark = pytest.mark.gpu

TOL_PARAMS, TOL_LATENT, TOL_DECODE = 1e-4, 1e-4, 1e-4


@pytest.fixture(scope="module")
def env(tiny_cfg):
    import ldt_amd
    from oracle import ldt_oracle as O
    assert torch.cuda.is_available()
    _, ssd = load_golden("score_tiny")
    tg, csd = load_golden("trainer_sample_tiny")
    score = ldt_amd.Score(tiny_cfg.score)
    score.load_state_dict(ssd["w"], strict=True)            # reference names/shapes load verbatim
    comp = ldt_amd.Compressor(tiny_cfg.compressor)
    comp.load_state_dict(csd["c"], strict=True)
    tr = ldt_amd.Trainer(tiny_cfg, score, comp, "cuda:0")
    return dict(ldt=ldt_amd, O=O, score=score, comp=comp, tr=tr, ssd=ssd["w"], csd=csd["c"], tg=tg, cfg=tiny_cfg)


def test_score_forward_golden(env):
    a, _ = load_golden("score_tiny")
    out = env["score"](a["x"].cuda(), a["t"].cuda())
    assert rel_mse(out.cpu(), a["out"]) < TOL_PARAMS


def test_score_forward_conditioned_golden(env):
    """ViPC-style conditioning (score.py:135,148-149): img condition added to the time embedding, point-cloud
    condition cross-attended on even blocks (K/V from the RAW condition, projected once and cached)."""
    a, _ = load_golden("score_tiny")
    pts = a["pts_cond"].transpose(1, 2).contiguous().cuda()          # back to the reference's channels-first (B,hidden,S)
    out = env["score"](a["x"].cuda(), a["t"].cuda(), condition=(pts, a["img_cond"].cuda()))
    assert rel_mse(out.cpu(), a["out_cond"]) < TOL_PARAMS
    assert rel_mse(out.cpu(), a["out"]) > 1e-3                       # and it is not the unconditional answer
    again = env["score"](a["x"].cuda(), a["t"].cuda(), condition=(pts, a["img_cond"].cuda()))
    assert torch.equal(out, again)                                    # cached K/V projection path
    only_img = env["score"](a["x"].cuda(), a["t"].cuda(), condition=(None, a["img_cond"].cuda()))
    ref = env["O"].score_forward(env["ssd"], env["cfg"].score, a["x"], a["t"], condition=(None, a["img_cond"]))
    assert rel_mse(only_img.cpu(), ref) < TOL_PARAMS
    with pytest.raises(ValueError):                                   # raw dicts need cfg.score.condition=True (c_net)
        env["score"](a["x"].cuda(), a["t"].cuda(), condition={"pts": pts})


def test_conditioned_sampling_loop_vs_oracle(env, monkeypatch):
    """Fused conditional loop (per-sample AdaLN rows recomputed in C++ every step + cross-attention) == the
    Python-driven loop == the oracle with the same condition, on injected noise — with the per-step rows from the fp32 SGEMM
    (LDT_ADALN_BF16=0: equals the Python-driven loop, whose rows are fp32, to 1e-6) and from the bf16 weight panel (the default:
    half the bytes of that HBM-bound GEMM; bf16 operand rounding like every token GEMM), both within the latent bar of the oracle."""
    O, tg, tr, cfg = env["O"], env["tg"], env["tr"], env["cfg"]
    monkeypatch.setenv("LDT_ADALN_BF16", "0")
    a, _ = load_golden("score_tiny")
    pts_tm = a["pts_cond"][:2]; img = a["img_cond"][:2]                     # token-major [B,S,hidden] / [B,t_dim]
    cond_dev = (pts_tm.transpose(1, 2).contiguous().cuda(), img.cuda())
    kw = dict(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", corrector=None, corrector_steps=1,
              shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False,
              denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"], condition=cond_dev)
    fused = tr.SDE.sample_discrete(**kw, use_graph=0)
    graph = tr.SDE.sample_discrete(**kw, use_graph=1)
    assert torch.equal(fused, graph)
    generic = tr.SDE.sample_discrete(**kw, record=[])
    assert rel_mse(generic.cpu(), fused.cpu()) < 1e-6
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, t: O.score_forward(env["ssd"], cfg.score, x, t, condition=(pts_tm, img)))
    ref = O.sample_discrete(sde, fn, tg["x0"], list(tg["noises"]), cfg.sde.sample_N)
    assert rel_mse(fused.cpu(), ref) < TOL_LATENT
    assert rel_mse(fused.cpu(), tg["eps"]) > 1e-3                            # differs from the unconditional trajectory
    monkeypatch.delenv("LDT_ADALN_BF16")                                     # the default: bf16 weight panel for the per-step rows
    fused_bf = tr.SDE.sample_discrete(**kw, use_graph=0)
    assert torch.equal(fused_bf, tr.SDE.sample_discrete(**kw, use_graph=1))
    e_bf, e_32 = rel_mse(fused_bf.cpu(), ref), rel_mse(fused.cpu(), ref)
    print("conditional loop vs oracle: fp32 AdaLN rows %.2e, bf16 weight panel %.2e" % (e_32, e_bf))
    assert e_bf < TOL_LATENT and not torch.equal(fused_bf, fused)


def test_label_conditioning_vs_oracle(tiny_cfg):
    """Class-conditional Score (num_categorys > 1): c = t_emb + LabelEmbedding(label) (score.py:125-135)."""
    import copy
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = copy.deepcopy(tiny_cfg)
    cfg.score.num_categorys = 5
    torch.manual_seed(9)
    score = ldt_amd.Score(cfg.score)
    sd = {k: v.detach().clone() for k, v in score.state_dict().items()}
    score = score.cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, cfg.score.z_scale, cfg.score.z_dim, generator=g); t = torch.tensor([0.8, 0.2, 0.01])
    label = torch.tensor([4, 0, 2])
    out = score(x.cuda(), t.cuda(), label=label.cuda())
    emb = torch.nn.functional.embedding(label, sd["LabelEmbedding.label_emb.weight"])
    l_emb = O.linear(sd, "LabelEmbedding.mlp.2", torch.nn.functional.silu(O.linear(sd, "LabelEmbedding.mlp.0", emb)))
    ref = O.score_forward(sd, cfg.score, x, t, label_emb=l_emb)
    assert rel_mse(out.cpu(), ref) < TOL_PARAMS


def test_time_table_matches_oracle(env):
    """c and every AdaLN row in fp32: the batch-shared table == per-sample evaluation."""
    O, score, cfg = env["O"], env["score"], env["cfg"]
    t = torch.tensor([1.0, 0.37, 1e-6])
    c, mod = score.time_table(t.cuda())
    c_ref = O.time_embedding(env["ssd"], "TimeEmbedding", t, cfg.score.t_dim // 4)
    assert rel_mse(c.cpu(), c_ref) < 1e-10
    D = cfg.score.hidden_size
    for l in range(cfg.score.num_blocks):
        ref = O.linear(env["ssd"], "Transformer.%d.adaLN.1" % l, torch.nn.functional.silu(c_ref))
        assert rel_mse(mod[:, l * 6 * D:(l + 1) * 6 * D].cpu(), ref) < 1e-10
    ref = O.linear(env["ssd"], "ln_out.adaLN.1", torch.nn.functional.silu(c_ref))
    assert rel_mse(mod[:, -2 * D:].cpu(), ref) < 1e-10


def test_teacher_forced_steps_golden(env):
    """Feed the reference's own x_i / t_i (captured trajectory) and compare params — per-step parity."""
    tg = env["tg"]
    worst = 0.0
    for j in range(tg["step_x"].shape[0]):
        p = env["score"](tg["step_x"][j].cuda(), tg["step_t"][j].cuda())
        worst = max(worst, rel_mse(p.cpu(), tg["step_params"][j]))
    assert worst < TOL_PARAMS, worst


@pytest.fixture(scope="module")
def conditioning(env):
    """Sensitivity of the REFERENCE map (fp32 oracle) to one bf16 rounding of x0: the floor for end-to-end
    decoded-cloud parity with these (random-weight) fixtures."""
    O, tg, cfg = env["O"], env["tg"], env["cfg"]
    g = torch.Generator().manual_seed(0)
    x0p = tg["x0"] * (1 + 2 ** -9 * torch.randn(tg["x0"].shape, generator=g))
    pts, eps = O.trainer_sample(env["ssd"], env["csd"], cfg, x0p, list(tg["noises"]))
    r2 = (tg["points"] ** 2).sum(-1).mean(1)
    return dict(eps=rel_mse(eps, tg["eps"]), pts=rel_mse(pts, tg["points"]),
                cd=float((O.chamfer_cd(pts, tg["points"]) / r2).max()))


@pytest.mark.parametrize("use_graph", [0, 1])
def test_trainer_sample_golden(env, conditioning, use_graph):
    """Trainer.sample end-to-end with the reference's recorded draws injected: latents, points, Chamfer."""
    O, tg, tr = env["O"], env["tg"], env["tr"]
    pts, eps = tr.sample(2, x0=tg["x0"], noise=tg["noises"], use_graph=use_graph)
    assert pts.shape == tg["points"].shape and eps.shape == tg["eps"].shape
    assert rel_mse(eps.cpu(), tg["eps"]) < TOL_LATENT
    # the decoder on the GPU's own latents vs the oracle decoder on the SAME latents, and vs the golden cloud
    assert rel_mse(pts.cpu(), tg["points"]) < max(2e-3, 4 * conditioning["pts"]), conditioning
    cd = O.chamfer_cd(pts.cpu(), tg["points"])
    radius2 = (tg["points"] ** 2).sum(-1).mean(1)
    assert float((cd / radius2).max()) < max(2e-3, 4 * conditioning["cd"]), conditioning


@pytest.mark.parametrize("use_graph", [0, 1])
def test_trainer_sample_trained_weights_fixed_bars(tiny_cfg, use_graph):
    """THE end-to-end points / Chamfer check (north-star: "final Chamfer"): weights trained by the reference's own
    Trainer.update (oracle/gen_trained_tiny_golden.py), so the sampled latents stay at the data scale (rms 0.5) and the
    decode is well conditioned.  FIXED bars, no floor multiplier: latents <= 1e-4, decoded points <= 1e-3 relative MSE,
    Chamfer / mean squared radius <= 1e-3, against what the REFERENCE's Trainer.sample produced on the same draws."""
    import ldt_amd
    from oracle import ldt_oracle as O
    a, sds = load_golden("trained_tiny")
    score = ldt_amd.Score(tiny_cfg.score); score.load_state_dict(sds["w"], strict=True)
    comp = ldt_amd.Compressor(tiny_cfg.compressor); comp.load_state_dict(sds["c"], strict=True)
    tr = ldt_amd.Trainer(tiny_cfg, score, comp, "cuda:0")
    worst = max(rel_mse(score(a["step_x"][j].cuda(), a["step_t"][j].cuda()).cpu(), a["step_params"][j]) for j in range(a["step_x"].shape[0]))
    assert worst < TOL_PARAMS, worst                                     # teacher-forced, the reference's own trajectory
    B = a["x0"].shape[0]
    traj = []
    pts, eps = tr.sample(B, x0=a["x0"], noise=a["noises"], use_graph=use_graph, trajectory=traj)
    e_lat, e_pts = rel_mse(eps.cpu(), a["eps"]), rel_mse(pts.cpu(), a["points"])
    cd = O.chamfer_cd(pts.cpu(), a["points"]) / (a["points"] ** 2).sum(-1).mean(1)
    print("trained-tiny end to end: latents %.2e points %.2e chamfer/r2 %.2e" % (e_lat, e_pts, float(cd.max())))
    assert e_lat < 1e-4 and e_pts < 1e-3 and float(cd.max()) < 1e-3, (e_lat, e_pts, float(cd.max()))
    # Why the latents sit at 7e-5 when every other latent comparison of the suite is at 2-4e-6: this fixture's schedule (N = 50:
    # betas up to 0.4, and a last denoising step that multiplies the Score output by beta / std(1e-6) = 6) carries a Score-output error of
    # relative MSE d to ~2 d in the final latents when it is independent from step to step, ~5-8 d when it is systematic — and ANY bf16-weight
    # path has a systematic d ~ 1e-5 here.  Measured on the fp32 oracle itself: the same trajectory with nothing but the GEMM weights rounded
    # to bf16 once (AdaLN / time MLP left in fp32, activations in fp32).  The GPU path must stay within 4x that at the end and at every step.
    sens = _oracle_bf16_weight_sensitivity(O, tiny_cfg, sds, a)
    xs = traj[0].cpu()
    curve = [rel_mse(xs[i], sens["rec"][i]) for i in range(xs.shape[0])]
    ratio = max(c / max(s_, 1e-7) for c, s_ in zip(curve, sens["curve"]))
    print("  oracle under bf16-rounded GEMM weights: final latents %.2e (GPU / that = %.2f), per-step max %.2e; GPU per-step max %.2e, worst per-step ratio %.2f"
          % (sens["final"], e_lat / sens["final"], max(sens["curve"]), max(curve), ratio))
    assert e_lat <= 4 * sens["final"], (e_lat, sens["final"])
    assert all(c <= 4 * max(s_, 1e-6) for c, s_ in zip(curve, sens["curve"])), (curve, sens["curve"])


_SENS_CACHE = {}


def _oracle_bf16_weight_sensitivity(O, cfg, sds, a):
    """The fp32 oracle's own trajectory on the trained-tiny fixture with the Score's GEMM weights (ln_in, fc_q / fc_kv / fc_o, mlp fc / out,
    ln_out.ln: what the product keeps as bf16 MFMA panels) rounded to bf16 once: {"final": rel-MSE of the final latents, "curve": per step,
    "rec": the unperturbed per-step states}.  CPU, a few seconds; cached per session."""
    if "v" in _SENS_CACHE:
        return _SENS_CACHE["v"]
    sd = sds["w"]
    is_w = lambda k: k.endswith("weight") and "adaLN" not in k and any(t in k for t in ("fc_q", "fc_kv", "fc_o", "mlp.fc", "mlp.out", "ln_in", "ln_out.ln"))
    sdq = {k: (v.to(torch.bfloat16).float() if is_w(k) else v) for k, v in sd.items()}
    assert sum(is_w(k) for k in sd) == 2 + 5 * cfg.score.num_blocks          # ln_in, ln_out.ln + (fc_q, fc_kv, fc_o, mlp.fc, mlp.out) per block
    nl = [a["noises"][i] for i in range(a["noises"].shape[0])]
    with torch.no_grad():
        rec, recq = [], []
        _, eps = O.trainer_sample(sd, sds["c"], cfg, a["x0"], nl, record=rec)
        _, epsq = O.trainer_sample(sdq, sds["c"], cfg, a["x0"], nl, record=recq)
    assert rel_mse(eps, a["eps"]) < 1e-10                                # the oracle reproduces the reference's capture
    _SENS_CACHE["v"] = {"final": rel_mse(epsq, eps), "curve": [rel_mse(recq[i][3], rec[i][3]) for i in range(len(rec))], "rec": [r[3] for r in rec]}
    return _SENS_CACHE["v"]


def test_free_running_per_step_curve(env):
    """Generic (Python-driven) loop with record: per-step relative MSE of x against the reference trajectory."""
    tg, tr, cfg = env["tg"], env["tr"], env["cfg"]
    rec = []
    tr.SDE.sample_discrete(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", corrector=None,
                           corrector_steps=1, shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps,
                           probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"],
                           record=rec)
    errs = [rel_mse(rec[i][0].cpu(), tg["step_x"][j]) for j, i in enumerate(tg["step_ids"].tolist())]
    assert max(errs) < TOL_LATENT, errs
    assert rel_mse(rec[-1][0].cpu(), tg["last_x"]) < TOL_LATENT


def test_fused_loop_equals_generic_loop_and_graph(env):
    tg, tr, cfg = env["tg"], env["tr"], env["cfg"]
    kw = dict(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", corrector=None,
              corrector_steps=1, shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps,
              probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"])
    a = tr.SDE.sample_discrete(**kw, use_graph=0)
    b = tr.SDE.sample_discrete(**kw, use_graph=1)
    assert torch.equal(a, b)                                  # graph replay == eager launches, bit for bit
    c = tr.SDE.sample_discrete(**kw, record=[])               # python-driven loop, per-sample AdaLN path
    assert rel_mse(c.cpu(), a.cpu()) < 1e-6


@pytest.mark.parametrize("pred", ["reversediffusion", "eulermaruyama", "ddim"])
def test_other_predictors_golden(env, pred):
    a, _ = load_golden("other_predictors")
    tg, tr, cfg = env["tg"], env["tr"], env["cfg"]
    out = tr.SDE.sample_discrete(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor=pred, corrector=None,
                                 corrector_steps=1, shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps,
                                 probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"])
    assert rel_mse(out.cpu(), a[pred]) < TOL_LATENT


@pytest.mark.parametrize("name", ["sub_vpsde", "vesde", "geometric_sde"])
def test_other_sde_families_golden(env, name):
    """The sub-VP, VE and geometric SDEs (diffusion_continuous.py:595-623, :681-766) through the generic predictors vs the
    reference's own sample_discrete (tests/golden/sde_types.npz).  sub_vpsde / vesde go through Trainer (its sde_type dispatch,
    Latent_SDE_Trainer.py:23-28, stock score_fn -> fused loop); geometric_sde, which upstream's Trainer cannot build either, through
    make_diffusion and a caller-side score_fn (python-driven loop, ldt_sde_score)."""
    import copy
    ldt, tg, tiny = env["ldt"], env["tg"], env["cfg"]
    a, _ = load_golden("sde_types")
    cfg = copy.deepcopy(tiny)
    cfg.sde.sde_type = name
    for k in ("sigma2_min", "sigma2_max", "sigma2_0"):
        if "%s/%s" % (name, k) in a:
            setattr(cfg.sde, k, float(a["%s/%s" % (name, k)]))
    if name == "geometric_sde":
        sde = ldt.make_diffusion(cfg.sde)
        score = env["score"]

        def score_fn(t, x, label=None, condition=None):
            params = score(x, t, label=label, condition=condition)
            return ldt.ops.sde_score(params, t.float(), sde.score_kind, *sde.score_consts()), params
    else:
        tr = ldt.Trainer(cfg, env["score"], env["comp"], "cuda:0")
        sde, score_fn = tr.SDE, tr.score_fn
    # the score half of score_fn against the reference's formula on the host
    t = torch.tensor([1.0, 0.37], device="cuda:0")
    sc, params = score_fn(t, tg["x0"].cuda())
    want = -params.cpu() / torch.sqrt(sde.var(t.cpu()))[:, None, None]
    assert rel_mse(sc.cpu(), want) < 1e-12
    for pred, pf in (("reversediffusion", False), ("eulermaruyama", False), ("reversediffusion", True)):
        out = sde.sample_discrete(score_fn=score_fn, num_samples=2, N=cfg.sde.sample_N, predictor=pred, corrector=None, corrector_steps=1,
                                  shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=pf,
                                  denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"])
        assert rel_mse(out.cpu(), a["%s/%s%s" % (name, pred, "_pf" if pf else "")]) < TOL_LATENT, (name, pred, pf)
    with pytest.raises(AttributeError):                                      # no betas table outside the VP-SDE, as upstream
        sde.sample_discrete(score_fn=score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", corrector=None, corrector_steps=1,
                            shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False,
                            denoise=True, snr=0.01, device="cuda:0", x0=tg["x0"], noise=tg["noises"])


def test_ancestral_corrector_and_print_steps_golden(env):
    """predictor + AncestralCorrector (2 corrector steps) and the print_steps trajectory dump vs the reference."""
    a, _ = load_golden("sampler_extras")
    tr, cfg = env["tr"], env["cfg"]
    kw = dict(score_fn=tr.score_fn, num_samples=2, N=cfg.sde.sample_N, predictor="ancestral", shape=(cfg.score.z_scale, cfg.score.z_dim),
              time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True, snr=float(a["snr"]), device="cuda:0")
    out = tr.SDE.sample_discrete(corrector="ancestral", corrector_steps=2, x0=a["corr_x0"], noise=a["corr_noise"], **kw)
    assert rel_mse(out.cpu(), a["corr_out"]) < TOL_LATENT
    traj = tr.SDE.sample_discrete(corrector=None, corrector_steps=1, print_steps=5, x0=a["print_x0"], noise=a["print_noise"], **kw)
    assert isinstance(traj, list) and len(traj) == a["print_out"].shape[0]
    assert rel_mse(torch.stack(traj).cpu(), a["print_out"]) < TOL_LATENT
    # device-noise mode runs and is reproducible
    o1 = tr.SDE.sample_discrete(corrector="ancestral", corrector_steps=1, x0=a["corr_x0"], seed=5, **kw)
    o2 = tr.SDE.sample_discrete(corrector="ancestral", corrector_steps=1, x0=a["corr_x0"], seed=5, **kw)
    assert torch.equal(o1, o2) and torch.isfinite(o1).all()
    with pytest.raises(NotImplementedError):
        tr.SDE.sample_discrete(x0=a["corr_x0"], corrector="bogus", corrector_steps=1, **kw)
    # B = 2 is neither 1 nor tokens (8): Langevin / PNDM raise the reference's broadcasting error (:208, :269)
    with pytest.raises(RuntimeError, match="must match the size of tensor"):
        tr.SDE.sample_discrete(x0=a["corr_x0"], corrector="langevin", corrector_steps=1, **kw)
    with pytest.raises(RuntimeError, match="must match the size of tensor"):
        tr.SDE.sample_discrete(**{**kw, "predictor": "pndm"}, corrector=None, corrector_steps=1, x0=a["corr_x0"])


def test_langevin_corrector_and_pndm_golden(env):
    """LangevinCorrector (diffusion_continuous.py:193-210) and PNDM (:260-316) vs outputs captured from the reference at
    B == tokens == 8 and B == 1 (the only batch sizes its (B,1) factors broadcast for), recorded draws injected."""
    import copy
    import ldt_amd
    a, _ = load_golden("sampler_langevin_pndm")
    cfg = copy.deepcopy(env["cfg"])
    cfg.sde.sample_N, cfg.sde.train_N = int(a["N"]), int(a["train_N"])
    tr = ldt_amd.Trainer(cfg, env["score"], env["comp"], "cuda:0")
    T, N = cfg.score.z_scale, cfg.sde.sample_N
    kw = dict(score_fn=tr.score_fn, N=N, shape=(T, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False,
              denoise=True, snr=float(a["snr"]), device="cuda:0")
    out = tr.SDE.sample_discrete(num_samples=T, predictor="eulermaruyama", corrector="langevin", corrector_steps=2,
                                 x0=a["lv8_x0"], noise=a["lv8_noise"], **kw)
    assert rel_mse(out.cpu(), a["lv8_out"]) < TOL_LATENT
    out = tr.SDE.sample_discrete(num_samples=1, predictor="reversediffusion", corrector="langevin", corrector_steps=1,
                                 x0=a["lv1_x0"], noise=a["lv1_noise"], **kw)
    assert rel_mse(out.cpu(), a["lv1_out"]) < TOL_LATENT
    for tag, B in (("pndm8", T), ("pndm1", 1)):
        out = tr.SDE.sample_discrete(num_samples=B, predictor="pndm", corrector=None, corrector_steps=1, x0=a[tag + "_x0"], **kw)
        assert rel_mse(out.cpu(), a[tag + "_out"]) < TOL_LATENT
    # device-noise Langevin: seeded, reproducible, finite
    o1 = tr.SDE.sample_discrete(num_samples=T, predictor="eulermaruyama", corrector="langevin", corrector_steps=1, x0=a["lv8_x0"], seed=11, **kw)
    o2 = tr.SDE.sample_discrete(num_samples=T, predictor="eulermaruyama", corrector="langevin", corrector_steps=1, x0=a["lv8_x0"], seed=11, **kw)
    assert torch.equal(o1, o2) and bool(torch.isfinite(o1).all())
    # Trainer.sample dispatches on cfg.sde.predictor / corrector
    cfg.sde.predictor, cfg.sde.corrector = "pndm", None
    _, eps = tr.sample(T, x0=a["pndm8_x0"])
    assert rel_mse(eps.cpu(), a["pndm8_out"]) < TOL_LATENT


def test_decoder_golden(env):
    a, _ = load_golden("decoder_tiny")
    pts = env["comp"].sample((2, 64), given_eps=a["given_eps"].cuda())
    assert rel_mse(pts.cpu(), a["points"]) < TOL_DECODE
    cd = env["O"].chamfer_cd(pts.cpu(), a["points"]) / (a["points"] ** 2).sum(-1).mean(1)
    assert float(cd.max()) < TOL_DECODE
    pts2 = env["comp"].decode(a["given_eps"].cuda(), 64)
    assert torch.equal(pts, pts2)


def test_decoder_keep_mask_golden(env):
    """Compressor.sample((B, n), ...) with n < max_outputs vs the reference: explicit keep_mask, and the reference's own
    randperm stream from a seeded CPU generator (`reference_rng`)."""
    a, _ = load_golden("decoder_keepmask")
    comp, n = env["comp"], int(a["num_points"])
    B = a["given_eps"].shape[0]
    pts = comp.sample((B, n), given_eps=a["given_eps"].cuda(), keep_mask=a["keep_mask"])
    assert pts.shape == (B, n, 3) and rel_mse(pts.cpu(), a["points"]) < TOL_DECODE
    torch.manual_seed(int(a["seed"]))
    pts2 = comp.sample((B, n), given_eps=a["given_eps"].cuda())      # draws B randperms exactly like sample_mask
    assert torch.equal(pts2, pts)
    other = comp.sample((B, n), given_eps=a["given_eps"].cuda(), keep_mask=a["keep_mask"].roll(1, 1))
    assert rel_mse(other.cpu(), a["points"]) > 1e-3                  # (the subset matters)


def test_philox_sampling_is_seeded_and_shard_invariant(env):
    """Device-noise mode: same seed -> same shapes; a batch of 4 == two shards of 2 with sample_offset."""
    tr, cfg = env["tr"], env["cfg"]
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(4, cfg.score.z_scale, cfg.score.z_dim, generator=g)
    kw = dict(score_fn=tr.score_fn, N=cfg.sde.sample_N, predictor="ancestral", corrector=None, corrector_steps=1,
              shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False,
              denoise=True, snr=0.01, device="cuda:0", seed=99)
    full = tr.SDE.sample_discrete(num_samples=4, x0=x0, **kw)
    again = tr.SDE.sample_discrete(num_samples=4, x0=x0, **kw)
    assert torch.equal(full, again)
    lo = tr.SDE.sample_discrete(num_samples=2, x0=x0[:2], sample_offset=0, **kw)
    hi = tr.SDE.sample_discrete(num_samples=2, x0=x0[2:], sample_offset=2, **kw)
    assert rel_mse(torch.cat([lo, hi]).cpu(), full.cpu()) < 1e-6
    assert torch.isfinite(full).all()


def test_midsize_score_vs_oracle():
    """Seeded random weights at a size the oracle finishes in seconds: B=4, T=64, hidden 256 (Dh 64), 4 blocks."""
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = ldt_amd.airplane_config(latent_tokens=64, **{"score.hidden_size": 256, "score.num_heads": 4,
                                                       "score.num_blocks": 4, "score.t_dim": 128})
    torch.manual_seed(3)
    score = ldt_amd.Score(cfg.score)
    sd = {k: v.detach().clone() for k, v in score.state_dict().items()}
    score = score.cuda()
    x = torch.randn(4, 64, 120); t = torch.tensor([1.0, 0.6, 0.2, 1e-6])
    out = score(x.cuda(), t.cuda())
    ref = O.score_forward(sd, cfg.score, x, t)
    assert rel_mse(out.cpu(), ref) < TOL_PARAMS


def test_lnfold_sampler_vs_oracle_and_unfolded(monkeypatch):
    """The fused sampling loop with LN folding forced on (hidden 256, 3 blocks, M = 8*32 = 256 rows) against the CPU
    oracle's free-running trajectory, and against the same loop with the LayerNorm kernels (LDT_LN_FOLD=0)."""
    import ldt_amd
    from oracle import ldt_oracle as O
    N = 25                                                    # (N <= 20 makes beta_max = 20/N >= 1: 1/sqrt(1-beta) blows up, in the reference too)
    cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N, **{
        "score.hidden_size": 256, "score.num_heads": 4, "score.num_blocks": 3, "score.t_dim": 128,
        "compressor.max_outputs": 256, "compressor.outsize": 256, "data.tr_max_sample_points": 256})
    torch.manual_seed(11)
    score = ldt_amd.Score(cfg.score)
    sd = {k: v.detach().clone() for k, v in score.state_dict().items()}
    comp = ldt_amd.Compressor(cfg.compressor)
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    B, T, z = 8, cfg.score.z_scale, cfg.score.z_dim
    x0, noises = O.draw_noises(77, B, T, z, N)
    kw = dict(score_fn=tr.score_fn, num_samples=B, N=N, predictor="ancestral", corrector=None, corrector_steps=1, shape=(T, z),
              time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=x0,
              noise=torch.stack(noises), streams=1)                  # (a sub-batch of 4 x 32 rows could not fold)
    monkeypatch.setenv("LDT_LN_FOLD", "2")
    assert tr.model.can_fold(B, T)
    folded = tr.SDE.sample_discrete(**kw, use_graph=0)
    folded_g = tr.SDE.sample_discrete(**kw, use_graph=1)
    assert torch.equal(folded, folded_g)
    monkeypatch.setenv("LDT_LN_FOLD", "0")
    assert not tr.model.can_fold(B, T)
    plain = tr.SDE.sample_discrete(**kw, use_graph=0)
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, t: O.score_forward(sd, cfg.score, x, t))
    ref = O.sample_discrete(sde, fn, x0, noises, N, predictor=cfg.sde.predictor, time_eps=cfg.sde.sample_time_eps,
                            denoise=cfg.sde.denoise, probability_flow=cfg.sde.probability_flow)
    e_f, e_p = rel_mse(folded.cpu(), ref), rel_mse(plain.cpu(), ref)
    assert e_f < TOL_LATENT and e_p < TOL_LATENT, (e_f, e_p)
    assert not torch.equal(folded, plain)                     # the two paths really are different code
    assert rel_mse(folded.cpu(), plain.cpu()) < TOL_LATENT


@pytest.mark.parametrize("use_graph", [0, 1])
def test_fold_monitor_inside_the_loop_sees_a_mid_trajectory_offset(monkeypatch, use_graph):
    """The LN-fold guard watches the WHOLE trajectory (round 6): `ldt_sample_loop` runs the monitored forward every
    LDT_FOLD_MONITOR_EVERY steps and on the last one, `sample_discrete` reads the running maximum of mean^2 / variance once after the
    loop.  A common offset over the hidden channels appears half-way through the steps (from step N/2 on the AdaLN table holds a gate
    of 1e4 for block 0's attention branch, whose output bias is 1 on every channel: every row of the residual stream gains the same
    constant): step 0 and a clean trajectory stay far below the bound; this one trips it, the model switches to the LayerNorm kernels,
    and the NEXT call runs exactly the LDT_LN_FOLD=0 path.  Both the plain loop and the two-graph replay (monitored / plain step graphs)."""
    import warnings
    import ldt_amd
    from oracle import ldt_oracle as O
    N = 25
    cfg = ldt_amd.airplane_config(latent_tokens=32, sample_N=N, **{
        "score.hidden_size": 256, "score.num_heads": 4, "score.num_blocks": 3, "score.t_dim": 128,
        "compressor.max_outputs": 256, "compressor.outsize": 256, "data.tr_max_sample_points": 256})
    torch.manual_seed(11)
    score = ldt_amd.Score(cfg.score)
    comp = ldt_amd.Compressor(cfg.compressor)
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    B, T, z = 8, cfg.score.z_scale, cfg.score.z_dim
    x0, noises = O.draw_noises(77, B, T, z, N)
    noise = torch.stack(noises)
    kw = dict(score_fn=tr.score_fn, num_samples=B, N=N, predictor="ancestral", corrector=None, corrector_steps=1, shape=(T, z),
              time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True, snr=0.01, device="cuda:0", x0=x0, streams=1,
              use_graph=use_graph)
    monkeypatch.setenv("LDT_FOLD_MONITOR_EVERY", "5")
    monkeypatch.setenv("LDT_LN_FOLD", "2")                    # (this small shape folds only when forced; the guard still records and switches)
    assert tr.model.can_fold(B, T)
    D = cfg.score.hidden_size
    with torch.no_grad():
        score.Transformer[0].fc_o.bias.fill_(1.0)             # (in place through the parameter: version bump -> repack)
        score.Transformer[0].fc_o.weight.zero_()              # block 0's attention branch = the constant 1 on every channel, times its gate
    clean = tr.SDE.sample_discrete(**kw, noise=noise)         # (the seeded gates are small and of either sign: no common offset)
    assert tr.model._fold_pending is not None                 # the loop left its running maximum on the device: no sync at the end of a call
    assert 0.0 < tr.model.collect_fold_ratio() < tr.model.FOLD_MAX_MEAN_RATIO and not tr.model._fold_disabled
    # from step N/2 on, block 0's attention gate is 1e4 on every channel; its branch is the constant 1 (fc_o.weight = 0, bias = 1): x gains a
    # common offset of 1e4 there (x += gate * 1) against a spread of O(10) (random-weight latents grow along the trajectory), i.e. mean^2 /
    # variance ~ 10^5 on every later LayerNorm input; steps 0 .. N/2 - 1 are untouched
    time_table = tr.model.time_table

    def late_gate(t, *a, **k):
        c, mod = time_table(t, *a, **k)
        if mod.shape[0] == N:
            mod[N // 2:, 2 * D:3 * D] = 1.0e4                 # block 0: shift_msa | scale_msa | gate_msa | ...
        return c, mod
    monkeypatch.setattr(tr.model, "time_table", late_gate)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.SDE.sample_discrete(**kw, noise=noise)
        tr.model.collect_fold_ratio()                         # (what the next call's can_fold() does first)
    monkeypatch.setattr(tr.model, "time_table", time_table)
    assert tr.model._fold_disabled and tr.model.fold_ratio_seen > tr.model.FOLD_MAX_MEAN_RATIO
    assert any("LayerNorm kernels from now on" in str(x.message) for x in w)
    monkeypatch.delenv("LDT_LN_FOLD")
    assert not tr.model.can_fold(B, T)                        # the next call takes the fallback path ...
    after = tr.SDE.sample_discrete(**kw, noise=noise)
    monkeypatch.setenv("LDT_LN_FOLD", "0")
    tr.model._fold_disabled = False
    plain = tr.SDE.sample_discrete(**kw, noise=noise)
    assert torch.equal(after, plain)                          # ... which is the LayerNorm-kernel loop, bit for bit
    assert rel_mse(clean.cpu(), plain.cpu()) < TOL_LATENT


def test_substreams_equal_single_stream(env, monkeypatch):
    """Sub-batches sampled on two concurrent HIP streams (two host threads, each with its own plan, workspace and step
    counter) give the same latents as one stream — injected noise and device Philox noise (keyed by global element index),
    unconditional and conditioned."""
    tg, tr, cfg = env["tg"], env["tr"], env["cfg"]
    g = torch.Generator().manual_seed(9)
    B = 6
    x0 = torch.randn(B, cfg.score.z_scale, cfg.score.z_dim, generator=g)
    kw = dict(score_fn=tr.score_fn, num_samples=B, N=cfg.sde.sample_N, predictor="ancestral", corrector=None, corrector_steps=1,
              shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, probability_flow=False, denoise=True,
              snr=0.01, device="cuda:0", x0=x0)
    noise = torch.randn(cfg.sde.sample_N, B, cfg.score.z_scale, cfg.score.z_dim, generator=g)
    a, _ = load_golden("score_tiny")
    pts = torch.randn(B, cfg.score.hidden_size, 5, generator=g).cuda(); img = torch.randn(B, cfg.score.t_dim, generator=g).cuda()
    for extra in (dict(noise=noise), dict(seed=123), dict(seed=7, condition=(pts, img))):
        one = tr.SDE.sample_discrete(**kw, **extra, streams=1)
        two = tr.SDE.sample_discrete(**kw, **extra, streams=2)
        three = tr.SDE.sample_discrete(**kw, **extra, streams=3)
        assert torch.isfinite(one).all()
        assert rel_mse(two.cpu(), one.cpu()) < 1e-6 and rel_mse(three.cpu(), one.cpu()) < 1e-6
    monkeypatch.setenv("LDT_STREAMS", "2")
    env_two = tr.SDE.sample_discrete(**kw, seed=123)
    assert rel_mse(env_two.cpu(), tr.SDE.sample_discrete(**kw, seed=123, streams=1).cpu()) < 1e-6


def test_ode_sampling_mode_vs_oracle(env):
    """sample_mode 'continuous' (sample_model_ode, diffusion_continuous.py:88-131): scipy RK45 on the host driving the HIP
    Score, against the oracle's restatement with the CPU Score on the same initial noise.  Both are adaptive solvers fed
    with slightly different function values (bf16 vs fp32), so the accepted steps may differ: the bar is the solver's own
    tolerance scale, not the per-step bf16 bar."""
    tr, cfg, O = env["tr"], env["cfg"], env["O"]
    g = torch.Generator().manual_seed(21)
    B, T, z = 2, cfg.score.z_scale, cfg.score.z_dim
    x1 = torch.randn(B, T, z, generator=g)
    tol, eps = 1e-3, 1e-2
    out, nfe, secs = tr.SDE.sample_model_ode(tr.score_fn, B, (T, z), eps, tol, noise=x1, device="cuda:0")
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, t: O.score_forward(env["ssd"], cfg.score, x, t))
    cnt = []
    ref = O.sample_model_ode(sde, fn, x1, eps, tol, nfe=cnt)
    assert out.shape == ref.shape and torch.isfinite(out).all() and nfe >= 7 and secs > 0
    assert rel_mse(out.cpu(), ref) < 1e-3, (rel_mse(out.cpu(), ref), nfe, len(cnt))
    # Trainer.sample in continuous mode returns decoded points of the ODE latents
    old_mode = tr.sample_mode
    try:
        tr.sample_mode = "continuous"
        tr.sample_time_eps, keep = eps, tr.sample_time_eps
        cfg.sde.ode_tol, keep_tol = tol, cfg.sde.ode_tol
        pts, lat = tr.sample(B, x0=x1)
        assert rel_mse(lat.cpu(), out.cpu()) < 1e-6 and pts.shape[0] == B and torch.isfinite(pts).all()
    finally:
        tr.sample_mode, tr.sample_time_eps, cfg.sde.ode_tol = old_mode, keep, keep_tol


def test_ema_swap_repacks_weights(env):
    """EMA swap (tools/utils.py:80-101) re-points parameters; the packed bf16 panels must follow."""
    ldt, cfg = env["ldt"], env["cfg"]
    score = ldt.Score(cfg.score)
    score.load_state_dict(env["ssd"])
    score = score.cuda()
    a, _ = load_golden("score_tiny")
    x, t = a["x"].cuda(), a["t"].cuda()
    base = score(x, t)
    ema = ldt.EMAWeights(score.parameters(), 0.999)
    for p in score.parameters():
        ema.state[p] = {"ema": torch.zeros_like(p.data)}
    ema.swap_parameters_with_ema(True)
    zeroed = score(x, t)
    assert float(zeroed.abs().max()) == 0.0                     # all-zero EMA weights => zero output
    ema.swap_parameters_with_ema(True)
    assert torch.equal(score(x, t), base)


def test_no_cpu_fallback(env):
    a, _ = load_golden("score_tiny")
    with pytest.raises(RuntimeError):
        env["score"](a["x"], a["t"])


def test_resume_reference_checkpoint_then_sample_golden():
    """(f)3: a checkpoint written by the reference's `Trainer.save` (tests/golden/checkpoint_tiny.pth) is resumed
    by `ldt_amd.Trainer`, `sample` swaps the optimizer's EMA weights in (repacking the bf16 panels) and reproduces
    the latents the reference produced after its own `resume` + `sample`; with EMA off it reproduces the raw-weight run."""
    import os
    import ldt_amd
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "checkpoint_tiny.pth")
    a, _ = load_golden("checkpoint_tiny_expect")
    cfg = torch.load(path, map_location="cpu", weights_only=False)["cfg"]
    torch.manual_seed(9)
    tr = ldt_amd.Trainer(cfg, ldt_amd.Score(cfg.score), ldt_amd.Compressor(cfg.compressor), "cuda:0")
    tr.resume(pretrain=path, strict=True)
    pts, eps = tr.sample(2, x0=a["x0"], noise=a["noises"])
    assert rel_mse(eps.cpu(), a["eps"]) < TOL_LATENT
    assert pts.shape == a["points"].shape and torch.isfinite(pts).all()
    tr.optimizer.apply_ema = False
    _, eps_raw = tr.sample(2, x0=a["x0"], noise=a["noises"])
    assert rel_mse(eps_raw.cpu(), a["eps_raw_weights"]) < TOL_LATENT
    assert rel_mse(eps_raw.cpu(), a["eps"]) > 30 * TOL_LATENT         # the swap matters (golden: 9.9e-3)


def test_score_unet_variant_golden(tiny_cfg):
    """(f)4: `unet: True` Score (up / mid / down blocks with skip concats, conv shortcut, adaLN1/adaLN2) vs the golden
    captured from the reference; the sampler drives it through the generic loop."""
    import copy
    import ldt_amd
    a, sds = load_golden("score_unet_tiny")
    cfg = copy.deepcopy(tiny_cfg)
    cfg.score.unet, cfg.score.num_blocks = True, int(a["num_blocks"])
    score = ldt_amd.Score(cfg.score)
    score.load_state_dict(sds["w"], strict=True)
    score = score.cuda()
    out = score(a["x"].cuda(), a["t"].cuda())
    assert rel_mse(out.cpu(), a["out"]) < TOL_PARAMS
    out_img = score(a["x"].cuda(), a["t"].cuda(), condition=(None, a["img_cond"].cuda()))
    assert rel_mse(out_img.cpu(), a["out_img"]) < TOL_PARAMS
    tr = ldt_amd.Trainer(cfg, score, ldt_amd.Compressor(cfg.compressor), "cuda:0")
    from oracle import ldt_oracle as O
    N = cfg.sde.sample_N
    x0, noises = O.draw_noises(77, 2, cfg.score.z_scale, cfg.score.z_dim, N)
    pts, eps = tr.sample(2, x0=x0, noise=torch.stack(noises))
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, tt: O.score_forward(sds["w"], cfg.score, x, tt))
    ref = O.sample_discrete(sde, fn, x0, noises, N)
    assert rel_mse(eps.cpu(), ref) < TOL_LATENT and torch.isfinite(pts).all()
