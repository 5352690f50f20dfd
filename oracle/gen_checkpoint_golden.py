"""TEST INFRASTRUCTURE ONLY — writes tests/golden/checkpoint_tiny.pth (+ checkpoint_tiny_expect.npz).

Runs only in the build container (imports /root/reference through oracle/ref_import.py).  The fixture is a
checkpoint file produced by the reference's own `Trainer.save` (trainer/Latent_SDE_Trainer.py:228-239) after two
real `EMA.step`s (tools/utils.py:33-70) on synthetic gradients, so it carries the genuine on-disk layout:
`score_state_dict`, `compressor_state_dict`, `score_optim_state_dict` (Adam `exp_avg`, `exp_avg_sq`, `step` and
the `ema` tensors keyed by parameter position), `score_scheduler`, `cfg` (argparse.Namespace), `epoch`, `itr`,
`time`.  The expectations are what a FRESH reference trainer produces after `resume(epoch)` (:241-266) followed by
`Trainer.sample(2)` (EMA weights swapped in, :146) on recorded noise.

    python oracle/gen_checkpoint_golden.py
"""
import os
import shutil
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import OUT, Recorder, save  # noqa: E402

OVERRIDES = {
    # smallest sizes the MFMA kernels accept, to keep the committed file small
    "score.z_scale": 8, "score.hidden_size": 64, "score.num_heads": 1, "score.num_blocks": 2, "score.t_dim": 64,
    "compressor.z_scales": 8, "compressor.hidden_dim": 64, "compressor.num_heads": 2, "compressor.p_dim": 32,
    "compressor.n_layers": 2, "compressor.z_dim": 60,
    "compressor.max_outputs": 64, "compressor.outsize": 64, "sde.sample_N": 30,
    # large steps + fast EMA so that raw and EMA weights give clearly different samples
    "opt.lr": 0.02, "opt.ema_decay": 0.5}


def ckpt_cfg():
    cfg = R.load_airplane_cfg(**OVERRIDES)
    cfg.data.tr_max_sample_points = 64
    return cfg


def torch2_optimizer_compat(opt):
    """The reference's EMA wrapper (tools/utils.py:25-31) subclasses Optimizer without calling its __init__; torch >= 2.0
    expects the hook registries that __init__ creates.  Give the instance empty ones (no behaviour change)."""
    from collections import OrderedDict
    for name in ("_optimizer_state_dict_pre_hooks", "_optimizer_state_dict_post_hooks",
                 "_optimizer_load_state_dict_pre_hooks", "_optimizer_load_state_dict_post_hooks",
                 "_optimizer_step_pre_hooks", "_optimizer_step_post_hooks"):
        if not hasattr(opt, name):
            setattr(opt, name, OrderedDict())
    return opt


def main():
    R.setup()
    from model.scorenet.score import Score
    from model.Compressor.Network import Compressor
    from trainer.Latent_SDE_Trainer import Trainer

    cfg = ckpt_cfg()
    assert cfg.opt.ema_decay > 0
    torch.manual_seed(77)
    score, comp = Score(cfg.score), Compressor(cfg.compressor)
    comp.eval(); comp.init()
    with R.quiet():
        tr = Trainer(cfg, score, comp, "cpu")
    torch2_optimizer_compat(tr.optimizer)
    # two genuine optimizer steps (Adam + EMA update) on synthetic gradients
    g = torch.Generator().manual_seed(3)
    for _ in range(2):
        for p in score.parameters():
            p.grad = 0.5 * torch.randn(p.shape, generator=g)
        tr.optimizer.step()
    n_ema = sum(1 for p in score.parameters() if "ema" in tr.optimizer.state[p])
    assert n_ema == len(list(score.parameters()))
    tr.epoch, tr.itr, tr.time = 7, 123, 4.5
    tr.save()
    src = os.path.join(cfg.log.save_path, "checkpt_7.pth")
    dst = os.path.join(OUT, "checkpoint_tiny.pth")
    shutil.copyfile(src, dst)
    print("wrote checkpoint_tiny.pth %.1f KB" % (os.path.getsize(dst) / 1024))

    # a fresh reference trainer (different init) resumes from the file and samples
    torch.manual_seed(5)
    score2, comp2 = Score(cfg.score), Compressor(cfg.compressor)
    comp2.eval()
    with R.quiet():
        tr2 = Trainer(cfg, score2, comp2, "cpu")
        torch2_optimizer_compat(tr2.optimizer)
        o_load = torch.load                     # torch >= 2.6 defaults to weights_only=True; the file holds a Namespace
        torch.load = lambda *a, **k: o_load(*a, **{**k, "weights_only": False})
        try:
            tr2.resume(epoch=7, strict=True)
        finally:
            torch.load = o_load
    assert tr2.epoch == 8 and tr2.itr == 123
    names = [n for n, _ in score2.named_parameters()]
    params = list(score2.parameters())
    probe = {}
    for i in (0, len(params) // 2, len(params) - 1):
        probe["ema::" + names[i]] = tr2.optimizer.state[params[i]]["ema"].clone()
        probe["raw::" + names[i]] = params[i].detach().clone()
    torch.set_grad_enabled(False)
    B, N = 2, cfg.sde.sample_N
    torch.manual_seed(99)
    with Recorder() as rec:
        pts, eps = tr2.sample(B)
    x0 = rec.draws[0][1]
    noises = torch.stack([d for _, d in rec.draws[1:N + 1]], 0)
    # same draws WITHOUT the EMA swap (raw weights): shows the test can tell the two apart
    tr2.optimizer.apply_ema = False
    it = iter([x0] + list(noises))
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: next(it)
    torch.randn_like = lambda *a, **k: next(it)
    try:
        _, eps_raw = tr2.sample(B)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    save("checkpoint_tiny_expect", x0=x0, noises=noises, points=pts, eps=eps, eps_raw_weights=eps_raw,
         epoch_after_resume=tr2.epoch, itr=tr2.itr, n_params=len(params),
         **probe)


if __name__ == "__main__":
    main()
