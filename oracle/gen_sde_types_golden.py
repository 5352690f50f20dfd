"""TEST INFRASTRUCTURE ONLY — writes tests/golden/sde_types.npz: the reference's other SDE families on the tiny Score.

Runs only in the build container (imports /root/reference through oracle/ref_import.py).

For each sde_type `make_diffusion` builds besides 'vpsde' (diffusion/diffusion_continuous.py:18-29: sub_vpsde, vesde,
geometric_sde) the reference's own `sample_discrete` is run on CPU with the generic predictors (reversediffusion, eulermaruyama;
'ancestral' / 'ddim' need the VP betas table and raise AttributeError upstream) on the x0 / noise draws of
tests/golden/trainer_sample_tiny.npz and the Score weights of tests/golden/score_tiny.npz, with the reference's score
definition (trainer/Latent_SDE_Trainer.py:57-61: -params / sqrt(var(t))).  The schedule functions f, g2, var, e2int_f are
captured at probe times as well.  The sde constants each family needs that the shipped YAML lacks (sigma2_min / sigma2_max;
sigma2_0 = sigma2_min for the VE-SDE, :741) are stored in the fixture.

    python oracle/gen_sde_types_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import save, tiny_cfg  # noqa: E402

FAMILIES = {
    "sub_vpsde": dict(),
    "vesde": dict(sigma2_min=0.01, sigma2_max=4.0, sigma2_0=0.01),
    "geometric_sde": dict(sigma2_min=3e-5, sigma2_max=0.999, sigma2_0=0.0),
}


def main():
    R.setup()
    from model.scorenet.score import Score
    from diffusion.diffusion_continuous import make_diffusion
    torch.set_grad_enabled(False)
    g = np.load(os.path.join(ROOT, "tests", "golden", "score_tiny.npz"))
    t = np.load(os.path.join(ROOT, "tests", "golden", "trainer_sample_tiny.npz"))
    x0, noises = torch.from_numpy(t["x0"]), torch.from_numpy(t["noises"])
    out = {}
    probe = torch.tensor([1.0, 0.73519, 0.5, 0.1, 1e-2, 1e-3], dtype=torch.float32)
    for name, extra in FAMILIES.items():
        cfg = tiny_cfg(N=50)
        cfg.sde.sde_type = name
        for k, v in extra.items():
            setattr(cfg.sde, k, v)
        score = Score(cfg.score).eval()
        score.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")})
        with R.quiet():
            sde = make_diffusion(cfg.sde)

        def score_fn(tt, x, label=None, condition=None):            # Latent_SDE_Trainer.py:57-61
            params = score(x, tt, label=label, condition=condition)
            return -params / torch.sqrt(sde.var(tt)[:, None, None]), params

        for fn in ("f", "g2", "var", "e2int_f"):
            out["%s/%s" % (name, fn)] = getattr(sde, fn)(probe)
        for k, v in extra.items():
            out["%s/%s" % (name, k)] = v
        for pred, pf in (("reversediffusion", False), ("eulermaruyama", False), ("reversediffusion", True)):
            it = iter([x0] + list(noises))
            o_randn, o_like = torch.randn, torch.randn_like
            torch.randn = lambda *a, **k: next(it)
            torch.randn_like = lambda *a, **k: next(it)
            try:
                res = sde.sample_discrete(score_fn=score_fn, N=cfg.sde.sample_N, corrector=None, predictor=pred, corrector_steps=1,
                                          shape=(cfg.score.z_scale, cfg.score.z_dim), time_eps=cfg.sde.sample_time_eps, label=None,
                                          denoise=True, device="cpu", num_samples=x0.shape[0], probability_flow=pf, snr=0.01,
                                          condition=None)
            finally:
                torch.randn, torch.randn_like = o_randn, o_like
            assert torch.isfinite(res).all(), (name, pred)
            out["%s/%s%s" % (name, pred, "_pf" if pf else "")] = res
            print(name, pred, pf, "rms %.4g" % res.pow(2).mean().sqrt().item())
    save("sde_types", probe_t=probe, **out)


if __name__ == "__main__":
    main()
