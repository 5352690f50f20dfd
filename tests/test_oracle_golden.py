"""CPU: pins the oracle (oracle/ldt_oracle.py) against golden vectors captured from the
imported upstream reference (oracle/gen_golden.py).  fp32 vs fp32 on the same CPU => the bar is
round-off (rel-MSE <= 1e-10; tables bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_mse
from oracle import ldt_oracle as O

TOL = 1e-10


def test_time_embedding():
    a, sds = load_golden("time_embedding")
    sd = {"TimeEmbedding." + k: v for k, v in sds["w"].items()}
    assert torch.equal(O.sinusoid(a["t"], 256), a["sinusoid"])          # Q5, bit-exact
    out = O.time_embedding(sd, "TimeEmbedding", a["t"], 256)
    assert rel_mse(out, a["out"]) < TOL


def _blk_sd(sds):
    return {"b." + k: v for k, v in sds["w"].items()}


def test_resblock_self_q1_q2():
    a, sds = load_golden("resblock_self")
    out = O.residual_block(_blk_sd(sds), "b", a["x"], None, a["c"], int(a["heads"]))
    assert rel_mse(out, a["out"]) < TOL


def test_resblock_encoder_raw_kv():
    a, sds = load_golden("resblock_encoder")
    out = O.residual_block(_blk_sd(sds), "b", a["x"], a["x"], a["c"], int(a["heads"]))
    assert rel_mse(out, a["out"]) < TOL
    # Q2: K/V from the raw stream differs from K/V from the modulated stream
    other = O.residual_block(_blk_sd(sds), "b", a["x"], None, a["c"], int(a["heads"]))
    assert rel_mse(other, a["out"]) > 1e-6


def test_resblock_cross():
    a, sds = load_golden("resblock_cross")
    out = O.residual_block(_blk_sd(sds), "b", a["x"], a["y"], a["c"], int(a["heads"]))
    assert rel_mse(out, a["out"]) < TOL


def test_resblock_decoder_affine_ln():
    a, sds = load_golden("resblock_decoder")
    sd = _blk_sd(sds)
    assert rel_mse(O.residual_block(sd, "b", a["x"], a["y"], None, int(a["heads"])), a["out"]) < TOL
    assert rel_mse(O.residual_block(sd, "b", a["x"], None, None, int(a["heads"])), a["out_self"]) < TOL


def test_q1_head_merge_is_raw_reinterpret():
    """A 'clean' head merge (permute heads back) must NOT match the reference."""
    a, sds = load_golden("resblock_decoder")
    sd = _blk_sd(sds)
    H = int(a["heads"])
    x = O.layer_norm(a["x"], sd["b.norm1.norm.weight"], sd["b.norm1.norm.bias"])
    q = O.linear(sd, "b.fc_q", x); kv = O.linear(sd, "b.fc_kv", a["y"])
    B, N, C = q.shape
    k, v = kv[..., :C], kv[..., C:]
    sp = lambda z: z.reshape(B, -1, H, C // H).permute(0, 2, 1, 3)
    o = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * (C // H) ** -0.5, -1) @ sp(v)
    clean = O.linear(sd, "b.fc_o", o.permute(0, 2, 1, 3).reshape(B, N, C))
    raw = O.attention(sd, "b", x, a["y"], H)
    assert rel_mse(clean, raw) > 1e-3


def test_final_layer():
    a, sds = load_golden("final_layer")
    out = O.final_layer({"f." + k: v for k, v in sds["w"].items()}, "f", a["x"], a["c"])
    assert rel_mse(out, a["out"]) < TOL


def test_score_tiny(tiny_cfg):
    a, sds = load_golden("score_tiny")
    out = O.score_forward(sds["w"], tiny_cfg.score, a["x"], a["t"])
    assert rel_mse(out, a["out"]) < TOL
    outc = O.score_forward(sds["w"], tiny_cfg.score, a["x"], a["t"], condition=(a["pts_cond"], a["img_cond"]))
    assert rel_mse(outc, a["out_cond"]) < TOL


def test_vpsde_tables_bit_exact(tiny_cfg):
    for N in (100, 1000):
        a, _ = load_golden("vpsde_tables_N%d" % N)
        tiny_cfg.sde.sample_N = N
        sde = O.VPSDE(tiny_cfg.sde)
        ts = torch.linspace(1.0, tiny_cfg.sde.sample_time_eps, N)
        assert torch.equal(ts, a["timesteps"])
        assert torch.equal(sde.betas, a["betas"]) and torch.equal(sde.alphas_cump, a["alphas_cump"])
        assert torch.equal((ts * (N - 1) / 1.0).long(), a["idx"])
        assert torch.equal(a["idx"], torch.arange(N - 1, -1, -1))        # Q7: idx == N-1-i
        for name in ("var", "std", "g2", "f", "e2int_f"):
            assert torch.equal(getattr(sde, name)(ts), a[name]), name
        # hard part 4: var(1e-6) is exactly one fp32 ulp
        assert float(a["var"][-1]) == float(np.float32(1.1920929e-07))
    tiny_cfg.sde.sample_N = 50


def test_trainer_sample_and_trajectory(tiny_cfg):
    a, sds = load_golden("trainer_sample_tiny")
    s, _ = load_golden("score_tiny")
    score_sd = load_golden("score_tiny")[1]["w"]
    N = int(a["N"])
    assert N == tiny_cfg.sde.sample_N
    # the recorded draws equal a fresh CPU generator seeded 1234 in the documented order
    x0, noises = O.draw_noises(1234, *a["x0"].shape, N)
    assert torch.equal(x0, a["x0"]) and torch.equal(torch.stack(noises), a["noises"])
    rec = []
    pts, eps = O.trainer_sample(score_sd, sds["c"], tiny_cfg, a["x0"], list(a["noises"]), record=rec)
    # teacher-forced per-step check
    sde = O.VPSDE(tiny_cfg.sde)
    for j, i in enumerate(a["step_ids"].tolist()):
        p = O.score_forward(score_sd, tiny_cfg.score, a["step_x"][j], a["step_t"][j])
        assert rel_mse(p, a["step_params"][j]) < TOL, i
    # free-running trajectory
    for j, i in enumerate(a["step_ids"].tolist()):
        assert rel_mse(rec[i][0], a["step_x"][j]) < 1e-8, i
    assert rel_mse(eps, a["eps"]) < 1e-8
    assert rel_mse(pts, a["points"]) < 1e-8


def test_trained_tiny_end_to_end(tiny_cfg):
    """The well-conditioned fixture: weights trained by the reference's own Trainer.update for 600 CPU iterations
    (oracle/gen_trained_tiny_golden.py), sampled by the reference's Trainer.sample — latents at the data scale."""
    a, sds = load_golden("trained_tiny")
    N = int(a["N"])
    assert N == tiny_cfg.sde.sample_N and float(a["latent_rms"]) < 1.0 and float(a["train_loss_last"]) < 0.5 * float(a["train_loss_first"])
    x0, noises = O.draw_noises(1234, *a["x0"].shape, N)
    assert torch.equal(x0, a["x0"]) and torch.equal(torch.stack(noises), a["noises"])
    for j in range(a["step_x"].shape[0]):
        p = O.score_forward(sds["w"], tiny_cfg.score, a["step_x"][j], a["step_t"][j])
        assert rel_mse(p, a["step_params"][j]) < TOL, j
    pts, eps = O.trainer_sample(sds["w"], sds["c"], tiny_cfg, a["x0"], list(a["noises"]))
    assert rel_mse(eps, a["eps"]) < 1e-9 and rel_mse(pts, a["points"]) < 1e-9
    # and the map IS well conditioned here: one bf16 rounding of x0 moves the decoded cloud by < 1e-4 (random weights: 7e-3)
    g = torch.Generator().manual_seed(0)
    x0p = a["x0"] * (1 + 2 ** -9 * torch.randn(a["x0"].shape, generator=g))
    pts_p, _ = O.trainer_sample(sds["w"], sds["c"], tiny_cfg, x0p, list(a["noises"]))
    assert rel_mse(pts_p, a["points"]) < 1e-4


def test_other_predictors(tiny_cfg):
    a, _ = load_golden("other_predictors")
    t, _ = load_golden("trainer_sample_tiny")
    score_sd = load_golden("score_tiny")[1]["w"]
    sde = O.VPSDE(tiny_cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, tt: O.score_forward(score_sd, tiny_cfg.score, x, tt))
    for pred in ("reversediffusion", "eulermaruyama", "ddim"):
        out = O.sample_discrete(sde, fn, t["x0"], list(t["noises"]), tiny_cfg.sde.sample_N, predictor=pred)
        assert rel_mse(out, a[pred]) < 1e-8, pred


def test_other_sde_families(tiny_cfg):
    """sub-VP, VE and geometric SDE through the generic predictors vs the reference's own sample_discrete (sde_types.npz)."""
    import copy
    a, _ = load_golden("sde_types")
    t, _ = load_golden("trainer_sample_tiny")
    score_sd = load_golden("score_tiny")[1]["w"]
    for name in ("sub_vpsde", "vesde", "geometric_sde"):
        c = copy.deepcopy(tiny_cfg.sde)
        c.sde_type = name
        for k in ("sigma2_min", "sigma2_max", "sigma2_0"):
            if "%s/%s" % (name, k) in a:
                setattr(c, k, float(a["%s/%s" % (name, k)]))
        sde = O.make_sde(c)
        fn = O.score_fn_from_model(sde, lambda x, tt: O.score_forward(score_sd, tiny_cfg.score, x, tt))
        for pred, pf in (("reversediffusion", False), ("eulermaruyama", False), ("reversediffusion", True)):
            out = O.sample_discrete(sde, fn, t["x0"], list(t["noises"]), tiny_cfg.sde.sample_N, predictor=pred, probability_flow=pf)
            assert rel_mse(out, a["%s/%s%s" % (name, pred, "_pf" if pf else "")]) < 1e-8, (name, pred, pf)


def test_corrector_and_print_steps(tiny_cfg):
    a, _ = load_golden("sampler_extras")
    score_sd = load_golden("score_tiny")[1]["w"]
    sde = O.VPSDE(tiny_cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, tt: O.score_forward(score_sd, tiny_cfg.score, x, tt))
    N = tiny_cfg.sde.sample_N
    assert a["corr_noise"].shape[0] == 3 * N                      # 1 predictor + 2 corrector draws per step
    out = O.sample_discrete(sde, fn, a["corr_x0"], list(a["corr_noise"]), N, corrector="ancestral", corrector_steps=2,
                            snr=float(a["snr"]))
    assert rel_mse(out, a["corr_out"]) < 1e-8
    traj = O.sample_discrete(sde, fn, a["print_x0"], list(a["print_noise"]), N, print_steps=5)
    assert len(traj) == a["print_out"].shape[0]
    assert rel_mse(torch.stack(traj), a["print_out"]) < 1e-8


def test_langevin_and_pndm(tiny_cfg):
    """LangevinCorrector (:193-210) and PNDM (:260-316) vs the reference at B == tokens == 8 and B == 1, plus its shape rule."""
    import copy
    a, _ = load_golden("sampler_langevin_pndm")
    score_sd = load_golden("score_tiny")[1]["w"]
    cfg = copy.deepcopy(tiny_cfg)
    cfg.sde.sample_N, cfg.sde.train_N = int(a["N"]), int(a["train_N"])
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, tt: O.score_forward(score_sd, cfg.score, x, tt))
    N, snr = cfg.sde.sample_N, float(a["snr"])
    out = O.sample_discrete(sde, fn, a["lv8_x0"], list(a["lv8_noise"]), N, predictor="eulermaruyama", corrector="langevin",
                            corrector_steps=2, snr=snr)
    assert rel_mse(out, a["lv8_out"]) < 1e-8
    out = O.sample_discrete(sde, fn, a["lv1_x0"], list(a["lv1_noise"]), N, predictor="reversediffusion", corrector="langevin",
                            corrector_steps=1, snr=snr)
    assert rel_mse(out, a["lv1_out"]) < 1e-8
    for tag in ("pndm8", "pndm1"):
        assert rel_mse(O.sample_pndm(cfg.sde, fn, a[tag + "_x0"], cfg.sde.sample_time_eps), a[tag + "_out"]) < 1e-8
    with pytest.raises(RuntimeError, match="must match the size of tensor"):
        O.sample_pndm(cfg.sde, fn, a["pndm8_x0"][:3], cfg.sde.sample_time_eps)
    with pytest.raises(RuntimeError, match="must match the size of tensor"):
        O.sample_discrete(sde, fn, a["lv8_x0"][:3], list(a["lv8_noise"][:, :3]), N, corrector="langevin", snr=snr)


def test_decoder(tiny_cfg):
    a, _ = load_golden("decoder_tiny")
    sd = load_golden("trainer_sample_tiny")[1]["c"]
    out = O.compressor_decode(sd, tiny_cfg.compressor, a["given_eps"])
    assert rel_mse(out, a["points"]) < TOL


def test_decoder_keep_mask(tiny_cfg):
    """num_points < max_outputs: InitialSet keeps the prior rows whose randperm rank is below num_points, in index order
    (model/Compressor/ops.py:6-14, layers.py:31-34)."""
    a, _ = load_golden("decoder_keepmask")
    sd = load_golden("trainer_sample_tiny")[1]["c"]
    assert int(a["keep_mask"].sum(1).min()) == int(a["num_points"]) == int(a["keep_mask"].sum(1).max())
    out = O.compressor_decode(sd, tiny_cfg.compressor, a["given_eps"], keep_mask=a["keep_mask"])
    assert out.shape == a["points"].shape and rel_mse(out, a["points"]) < TOL
    torch.manual_seed(int(a["seed"]))                               # the recorded draws are what a seeded CPU generator yields
    perms = torch.stack([torch.randperm(tiny_cfg.compressor.max_outputs) for _ in range(a["perms"].shape[0])])
    assert torch.equal(perms, a["perms"])


def test_compressor_options(tiny_cfg):
    """Options no shipped YAML uses, vs the reference: norm_input + pre_group (Network.py:170-174,188-195) and the
    mixture-of-Gaussians InitialSet (max_outputs None, Compressor/layers.py:17-24,38-42)."""
    import copy
    a, sds = load_golden("compressor_options")
    cc = copy.deepcopy(tiny_cfg.compressor)
    cc.n_layers, cc.encoder_layers = 2, 1
    ca = copy.deepcopy(cc); ca.norm_input, ca.pre_group = True, True
    r = O.compressor_encode(sds["a"], ca, a["a_pts"], list(a["a_post_noise"]))
    assert rel_mse(r["all_eps"], a["a_all_eps"]) < 1e-9 and rel_mse(r["set"], a["a_set"]) < 1e-9
    assert abs(float(r["max"]) - float(a["a_max"])) < 1e-4 * abs(float(a["a_max"]))
    dec = O.compressor_decode(sds["b"], cc, a["b_given_eps"], seed_eps=a["b_seed_eps"])
    assert dec.shape == a["b_points"].shape and rel_mse(dec, a["b_points"]) < TOL
    r = O.compressor_encode(sds["b"], cc, a["b_pts"], list(a["b_post_noise"]), seed_eps=a["b_fwd_seed_eps"])
    assert rel_mse(r["all_eps"], a["b_all_eps"]) < 1e-9 and rel_mse(r["set"], a["b_set"]) < 1e-9
    # pos_embedding: mlp (per-token position condition, Network.py:133-134) and class_condition (:137-142,197-198,218,225)
    cm = copy.deepcopy(cc); cm.pos_embedding = "mlp"
    r = O.compressor_encode(sds["c"], cm, a["c_pts"], list(a["c_post_noise"]))
    assert rel_mse(r["all_eps"], a["c_all_eps"]) < 1e-9 and rel_mse(r["set"], a["c_set"]) < 1e-9
    cl = copy.deepcopy(cc); cl.class_condition, cl.num_categorys = True, 5
    r = O.compressor_encode(sds["d"], cl, a["c_pts"], list(a["d_post_noise"]), label=a["d_label"])
    assert rel_mse(r["all_eps"], a["d_all_eps"]) < 1e-9 and rel_mse(r["set"], a["d_set"]) < 1e-9
    assert rel_mse(O.compressor_decode(sds["d"], cl, a["b_given_eps"]), a["d_points"]) < TOL    # decode never sees a label


def test_compressor_variants(tiny_cfg):
    """`decoder_act` (four activations, incl. an unknown name = ReLU as upstream), `ActNorm: ~` and the dead `AdaLN: False` flag vs outputs
    captured from the reference (oracle/gen_compressor_variants_golden.py; one shared weight set)."""
    import copy
    a, sds = load_golden("compressor_variants")
    sd = sds["w"]
    cc = copy.deepcopy(tiny_cfg.compressor)
    cc.n_layers, cc.encoder_layers = 2, 1
    for tag, act in (("g", "gelu"), ("l", "leakyrelu0.2"), ("h", "hardswish"), ("r", "anything-else-is-relu")):
        ca = copy.deepcopy(cc); ca.decoder_act = act
        dec = O.compressor_decode(sd, ca, a["given_eps"])
        assert rel_mse(dec, a[tag + "_points"]) < TOL, tag
        r = O.compressor_encode(sd, ca, a["pts"], list(a[tag + "_post_noise"]))
        assert rel_mse(r["all_eps"], a[tag + "_all_eps"]) < 1e-9 and rel_mse(r["set"], a[tag + "_set"]) < 1e-9, tag
    assert rel_mse(a["g_points"], a["l_points"]) > 1e-3                      # the activation matters
    cn = copy.deepcopy(cc); cn.ActNorm, cn.AdaLN = None, False
    r = O.compressor_encode({k: v for k, v in sd.items() if not k.startswith("conv_in.")}, cn, a["pts"], list(a["n_post_noise"]))
    assert rel_mse(r["all_eps"], a["n_all_eps"]) < 1e-9 and rel_mse(r["set"], a["n_set"]) < 1e-9


def test_norm_variants(tiny_cfg):
    """`norm: group_norm` and `norm: ~` (tools/utils.py:168-181) for the Score (plain, with point + image condition) and the Compressor (decode,
    encode) vs outputs captured from the reference (oracle/gen_norm_variants_golden.py); `batch_norm` raised upstream when the fixture was made."""
    import copy
    a, sds = load_golden("norm_variants")
    assert float(a["batch_norm_error"]) == 1.0
    for tag, kind in (("gn", "group_norm"), ("id", None)):
        cs = copy.deepcopy(tiny_cfg.score); cs.norm = kind
        out = O.score_forward(sds[tag + "s"], cs, a["x"], a["t"])
        assert rel_mse(out, a[tag + "_out"]) < TOL, tag
        out = O.score_forward(sds[tag + "s"], cs, a["x"], a["t"], condition=(a["pts_cond"], a["img_cond"]))
        assert rel_mse(out, a[tag + "_out_cond"]) < TOL, tag
        cc = copy.deepcopy(tiny_cfg.compressor); cc.norm = kind
        cc.n_layers, cc.encoder_layers = 2, 1
        assert rel_mse(O.compressor_decode(sds[tag + "c"], cc, a["given_eps"]), a[tag + "_points"]) < TOL, tag
        r = O.compressor_encode(sds[tag + "c"], cc, a["pts"], list(a[tag + "_post_noise"]))
        assert rel_mse(r["all_eps"], a[tag + "_all_eps"]) < 1e-9 and rel_mse(r["set"], a[tag + "_set"]) < 1e-9, tag
    assert rel_mse(a["gn_out"], a["id_out"]) > 1e-3                            # the norm matters


def test_encoder(tiny_cfg):
    a, _ = load_golden("compressor_fwd_tiny")
    sd = load_golden("trainer_sample_tiny")[1]["c"]
    cc = tiny_cfg.compressor
    T = cc.z_scales
    # discrete stages: FPS indices, kNN index *sets*
    fi = O.fps(a["pts"], T)
    assert torch.equal(fi, a["fps_idx"].long())
    assert rel_mse(O.square_distance(O.gather(a["pts"], fi), a["pts"]), a["sqdist"]) < TOL
    ki = O.knn(a["pts"].shape[1] // T * 2, a["pts"], O.gather(a["pts"], fi))
    assert torch.equal(ki.sort(-1)[0], a["knn_idx"].sort(-1)[0])
    r = O.compressor_encode(sd, cc, a["pts"], list(a["post_noise"]))
    assert rel_mse(r["centers"], a["centers"].transpose(1, 2) if a["centers"].shape[1] == 3 else a["centers"]) < TOL
    assert rel_mse(torch.stack(r["mu"]), a["mu"]) < 1e-9
    assert rel_mse(torch.stack(r["logvar"]), a["logvar"]) < 1e-9
    assert rel_mse(r["all_eps"], a["all_eps"]) < 1e-9
    assert rel_mse(r["set"], a["set"]) < 1e-9
    assert abs(float(r["max"]) - float(a["max"])) < 1e-4 * abs(float(a["max"]))


def test_compressor_encode_near_origin_points(tiny_cfg):
    """The encode golden whose clouds hold points inside the |p|^2 <= 1e-3 ball (FPS with upstream pointnet2_ops' skip rule,
    the oracle's default): centres, latents and reconstruction vs the reference's Compressor.forward."""
    a, _ = load_golden("compressor_fwd_origin")
    sd = load_golden("trainer_sample_tiny")[1]["c"]
    cc = tiny_cfg.compressor
    assert O.FPS_SKIP_NEAR_ORIGIN is True
    assert torch.equal(O.fps(a["pts"], cc.z_scales), a["fps_idx"].long())
    assert torch.equal(O.fps(a["pts"], cc.z_scales, skip_near_origin=False), a["fps_idx_twin"].long())
    r = O.compressor_encode(sd, cc, a["pts"], list(a["post_noise"]))
    assert rel_mse(r["all_eps"], a["all_eps"]) < 1e-9 and rel_mse(r["set"], a["set"]) < 1e-9


def test_fps_tie_break_and_start():
    """Exact ties (duplicate points): index 0 is always first (sampling.cu:105-107); a tie goes to
    the smaller (k % 512, k // 512) — the 512-thread strided scan + pairwise tree (:141-158)."""
    p = torch.tensor([[[0., 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 2, 0]]])
    idx = O.fps(p, 4)[0].tolist()
    assert idx == [0, 4, 1, 3]
    # n > 512: k=513 is scanned by thread 1, k=2 by thread 2 -> 513 wins the tie
    q = torch.zeros(1, 600, 3)
    q[0, 2] = torch.tensor([3., 0, 0]); q[0, 513] = torch.tensor([3., 0, 0])
    assert O.fps(q, 2)[0].tolist() == [0, 513]


def test_chamfer():
    a, _ = load_golden("chamfer")
    dl, dr = O.dist_chamfer(a["a"], a["b"])
    assert rel_mse(dl, a["dl"]) < TOL and rel_mse(dr, a["dr"]) < TOL
    assert rel_mse(O.chamfer_cd(a["a"], a["b"]), a["cd"]) < TOL


def test_checkpoint_ema_weights_reproduce_reference_sample():
    """a3 / (f)3: the oracle fed the checkpoint's EMA tensors (optimizer state, keyed by parameter position) gives the
    sample the reference produced after `resume` + `sample`; fed the raw weights it gives the other recorded one."""
    import os
    from conftest import GOLDEN
    ck = torch.load(os.path.join(GOLDEN, "checkpoint_tiny.pth"), map_location="cpu", weights_only=False)
    a, _ = load_golden("checkpoint_tiny_expect")
    cfg = ck["cfg"]
    assert int(a["epoch_after_resume"]) == ck["epoch"] + 1 and int(a["itr"]) == ck["itr"]
    raw = ck["score_state_dict"]
    st = ck["score_optim_state_dict"]["state"]
    names = [k for k in raw]                                       # state_dict order == parameter order (no buffers)
    assert len(st) == len(names) == int(a["n_params"])
    ema = {n: st[i]["ema"] for i, n in enumerate(names)}
    for n in names:
        assert ema[n].shape == raw[n].shape
    _, eps = O.trainer_sample(ema, ck["compressor_state_dict"], cfg, a["x0"], list(a["noises"]))
    _, eps_raw = O.trainer_sample(raw, ck["compressor_state_dict"], cfg, a["x0"], list(a["noises"]))
    assert rel_mse(eps, a["eps"]) < 1e-8
    assert rel_mse(eps_raw, a["eps_raw_weights"]) < 1e-8
    assert rel_mse(a["eps"], a["eps_raw_weights"]) > 5e-3          # the two are clearly distinguishable


def test_condition_net_point_branch():
    """(f)1: ViPC ConditionNet point branch (score.py:37-41; LocalGrouper normalize='center') vs the reference."""
    a, sds = load_golden("condition_net_pts")
    sd = {"c_net." + k: v for k, v in sds["w"].items()}
    out, fi, ki = O.condition_net_points(sd, "c_net", a["pts"], int(a["patch_size"]))
    assert int(a["k"]) == 128 // int(a["patch_size"]) * 2          # k comes from the CHANNEL count (score.py:40)
    assert torch.equal(fi, a["fps_idx"])
    assert torch.equal(ki.sort(-1)[0], a["knn_idx"].sort(-1)[0])
    assert rel_mse(out, a["pts_condition"].transpose(1, 2)) < TOL
    # 'anchor' normalisation would give a different answer: the mode is exercised
    x = O.linear(sd, "c_net.pc_conv_in", a["pts"])
    _, tok, _, _ = O.local_grouper(sd, "c_net.group", a["pts"], x, int(a["patch_size"]), int(a["k"]), normalize="anchor")
    assert rel_mse(O.linear(sd, "c_net.pc_conv_out", tok), a["pts_condition"].transpose(1, 2)) > 1e-3


def test_score_unet_variant(tiny_cfg):
    """(f)4: `unet: True` Score (score.py:67-83,138-146) incl. the dim_in != dim_out down blocks (layers.py:216-218)."""
    import copy
    a, sds = load_golden("score_unet_tiny")
    cfg = copy.deepcopy(tiny_cfg.score)
    cfg.unet, cfg.num_blocks = True, int(a["num_blocks"])
    assert any(k.startswith("Transformer_Down.0.adaLN2") for k in sds["w"])
    assert rel_mse(O.score_forward(sds["w"], cfg, a["x"], a["t"]), a["out"]) < TOL
    assert rel_mse(O.score_forward(sds["w"], cfg, a["x"], a["t"], condition=(None, a["img_cond"])), a["out_img"]) < TOL


def test_validation_metrics_cd():
    """(f)4: pairwise CD, MMD/COV and the 1-NN test vs the reference's evaluation module on CPU."""
    a, _ = load_golden("metrics_cd")
    M_rs = O.pairwise_cd(a["ref"], a["smp"])
    assert rel_mse(M_rs, a["M_rs"]) < 1e-10
    mc = O.lgan_mmd_cov(a["M_rs"].t())
    assert torch.allclose(mc["mmd"], a["lgan_mmd"]) and float(mc["cov"]) == float(a["lgan_cov"])
    k1 = O.knn_two_sample(a["M_rr"], a["M_rs"], a["M_ss"], 1)
    for nm in ("acc", "tp", "fp", "fn", "tn"):
        assert float(k1[nm]) == float(a["knn1_" + nm]), nm
    res = O.compute_cd_metrics(a["smp"], a["ref"])
    assert torch.allclose(res["mmd-CD"], a["mmd_cd"]) and float(res["cov-CD"]) == float(a["cov_cd"])
    assert float(res["1-NN-CD-acc"]) == float(a["one_nn_cd_acc"])
    assert 0.0 < float(a["cov_cd"]) < 1.0 and 0.0 < float(a["one_nn_cd_acc"]) < 1.0       # a non-trivial fixture


def test_emd_approxmatch_properties():
    """EMD restatement (approxmatch.cu; parity unpinned — CUDA-only upstream): sanity anchors.  The matching transports
    (almost) all mass, so cost/n is bounded below by ~the exact assignment's mean distance (the reference's CPU fallback
    `emd_approx`, golden `emd_exact`) and stays within a modest factor of it; identical clouds cost ~0."""
    a, _ = load_golden("metrics_cd")
    x, y = a["smp"][:6], a["ref"][:6]
    approx = O.emd_approxmatch_cost(x, y) / x.shape[1]
    exact = a["emd_exact"]
    assert (approx > 0.97 * exact).all() and (approx < 1.6 * exact).all(), (approx, exact)
    same = O.emd_approxmatch_cost(x[:2], x[:2]) / x.shape[1]
    assert (same < 0.05 * exact[:2]).all(), same
    perm = torch.randperm(x.shape[1], generator=torch.Generator().manual_seed(0))
    assert torch.allclose(O.emd_approxmatch_cost(x[:2][:, perm], y[:2]), O.emd_approxmatch_cost(x[:2], y[:2]), rtol=1e-4)


def test_ode_sampler_restatement_analytic():
    """oracle.sample_model_ode (probability-flow ODE through scipy RK45 in reversed time, diffusion_continuous.py:88-131)
    on a case with a closed form: data = a point mass at 0 gives score(t, x) = -x / var(t), and the flow then carries
    x(1) to x(1) * std(eps) / std(1); data ~ N(0, I) gives score = -x and a constant state."""
    import json, os
    from conftest import GOLDEN, to_ns
    from oracle import ldt_oracle as O
    with open(os.path.join(GOLDEN, "tiny_cfg.json")) as f:
        cfg = to_ns(json.load(f))
    sde = O.VPSDE(cfg.sde)
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(3, 4, 5, generator=g)
    eps = 1e-2
    nfe = []
    out = O.sample_model_ode(sde, lambda t, x: (-x / sde.var(t)[:, None, None], None), x1, eps, 1e-7, nfe=nfe)
    t1, te = torch.tensor(1.0), torch.tensor(eps)
    ref = x1 * sde.std(te) / sde.std(t1)
    assert rel_mse(out, ref) < 1e-8 and len(nfe) > 6
    const = O.sample_model_ode(sde, lambda t, x: (-x, None), x1, eps, 1e-7)
    assert rel_mse(const, x1) < 1e-10
