"""TEST INFRASTRUCTURE ONLY — writes tests/golden/trained_tiny.npz: a WELL-CONDITIONED end-to-end sampling fixture.

Runs only in the build container (imports /root/reference through oracle/ref_import.py).

Why: with untrained Score weights the reverse SDE inflates the latents to rms ~ 10^2..10^3 (prod 1/sqrt(1-beta_i) = e^5),
where the decoder's softmaxes saturate and even the fp32 CPU path moves by 1e-2 under one bf16 rounding of its input — an
end-to-end points / Chamfer comparison can then only be held to a self-calibrated bar.  Here the reference's
OWN training step — `Trainer.update` (trainer/Latent_SDE_Trainer.py:94-141: compressor encode under no_grad, VP-SDE noising,
eps-prediction loss, grad clipping, Adam + EMA) — is run for a few hundred CPU iterations on the tiny config, so that the
Score has learned to denoise and sampled latents stay at the data scale (rms ~ 1).  Then the reference's `Trainer.sample`
(:143-165, EMA weights swapped in) is captured on recorded noise:

    w::*      Score state_dict holding the EMA weights the sample used          c::*  Compressor state_dict
    x0, noises, step_ids / step_x / step_params (teacher-forcing), eps (final latents), points (decoded clouds)

The GPU suite holds latents, points and the normalised Chamfer distance of this run to FIXED bars (tests/test_gpu_path.py).

    python oracle/gen_trained_tiny_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_checkpoint_golden import torch2_optimizer_compat  # noqa: E402
from oracle.gen_golden import Recorder, save, sd_np, tiny_cfg  # noqa: E402

ITERS, BATCH, POOL = 600, 32, 256


def main():
    R.setup()
    from model.scorenet.score import Score
    from model.Compressor.Network import Compressor
    from trainer.Latent_SDE_Trainer import Trainer

    cfg = tiny_cfg(N=50)                       # == tests/golden/tiny_cfg.json (checked below)
    cfg.opt.lr, cfg.opt.warmup_iters, cfg.opt.ema_decay = 2e-3, 20, 0.98
    torch.manual_seed(21)
    score, comp = Score(cfg.score), Compressor(cfg.compressor)
    comp.eval(); comp.init()
    with R.quiet():
        tr = Trainer(cfg, score, comp, "cpu")
    torch2_optimizer_compat(tr.optimizer)

    # synthetic "dataset": unit-normalised random clouds (datasets/ShapeNet_55.py:50-54 normalisation)
    g = torch.Generator().manual_seed(5)
    P = cfg.data.tr_max_sample_points
    pool = torch.randn(POOL, P, 3, generator=g)
    pool = pool - pool.mean(1, keepdim=True)
    pool = pool / pool.norm(dim=-1).amax(1)[:, None, None]
    import numpy as np
    np.random.seed(11)                          # update_score draws its time indices from numpy (:109)
    losses = []
    for it in range(ITERS):
        idx = torch.randint(0, POOL, (BATCH,), generator=g)
        loss = tr.update({"tr_points": pool[idx]})
        losses.append(float(loss))
        if it % 100 == 0 or it == ITERS - 1:
            print("iter %4d  loss %.4f" % (it, sum(losses[-20:]) / len(losses[-20:])))
    assert sum(losses[-50:]) / 50 < 0.6 * sum(losses[:5]) / 5, "the tiny Score did not learn"

    # make the EMA weights the model's parameters (what sample() swaps in), freeze the swap, and sample
    tr.optimizer.swap_parameters_with_ema(store_params_in_ema=True)
    tr.optimizer.apply_ema = False
    score.eval()
    torch.set_grad_enabled(False)
    B, N = 4, cfg.sde.sample_N
    steps = [0, 1, N // 2, N - 2, N - 1]
    rec_steps = {}
    orig_fwd = score.forward
    count = {"i": 0}

    def spy(x, t, label=None, condition=None):
        out = orig_fwd(x, t, label=label, condition=condition)
        if count["i"] in steps:
            rec_steps[count["i"]] = (x.clone(), t.clone(), out.clone())
        count["i"] += 1
        return out

    score.forward = spy
    torch.manual_seed(1234)
    with Recorder() as rec:
        pts, eps = tr.sample(B)
    score.forward = orig_fwd
    x0 = rec.draws[0][1]
    noises = torch.stack([d for _, d in rec.draws[1:N + 1]], 0)
    rms = float(eps.pow(2).mean().sqrt())
    data_rms = float(comp(pool[:BATCH])["all_eps"].pow(2).mean().sqrt())
    print("sampled latents rms %.3f (data latents rms %.3f); points range [%.3f, %.3f]" % (rms, data_rms, float(pts.min()), float(pts.max())))
    assert rms < 3.0 * data_rms + 1.0, "latents left the data scale: the fixture would be ill-conditioned again"
    save("trained_tiny", x0=x0, noises=noises, points=pts, eps=eps, N=N, latent_rms=rms, data_rms=data_rms,
         step_ids=torch.tensor(steps),
         step_x=torch.stack([rec_steps[i][0] for i in steps]),
         step_t=torch.stack([rec_steps[i][1] for i in steps]),
         step_params=torch.stack([rec_steps[i][2] for i in steps]),
         train_loss_first=sum(losses[:5]) / 5, train_loss_last=sum(losses[-50:]) / 50,
         **sd_np(score.state_dict(), "w::"), **sd_np(comp.state_dict(), "c::"))


if __name__ == "__main__":
    main()
