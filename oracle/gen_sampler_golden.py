"""TEST INFRASTRUCTURE ONLY — golden vectors for LangevinCorrector and PNDM, captured from the imported reference.

    python oracle/gen_sampler_golden.py      # writes tests/golden/sampler_langevin_pndm.npz

Both samplers multiply a (B,1) factor into (B,tokens,z) latents (diffusion/diffusion_continuous.py:208-209, :267-271),
so the reference only runs them when B == 1 or B == tokens: captured at B == tokens == 8 (and B == 1 for Langevin) on the
weights of tests/golden/score_tiny.npz.  Draws are recorded in consumption order like oracle/gen_golden.py does.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import OUT, Recorder, save, tiny_cfg  # noqa: E402


def main():
    R.setup()
    from model.scorenet.score import Score
    from model.Compressor.Network import Compressor
    from trainer.Latent_SDE_Trainer import Trainer
    torch.set_grad_enabled(False)
    cfg = tiny_cfg(N=12)
    cfg.sde.train_N = 40                                           # PNDM walks sample_N steps over train_N levels
    z = np.load(os.path.join(OUT, "score_tiny.npz"))
    torch.manual_seed(0)
    score = Score(cfg.score).eval()
    score.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}, strict=True)
    comp = Compressor(cfg.compressor).eval()
    with R.quiet():
        tr = Trainer(cfg, score, comp, "cpu")
    T, zd, N = cfg.score.z_scale, cfg.score.z_dim, cfg.sde.sample_N
    out = {"N": N, "train_N": cfg.sde.train_N, "snr": 0.16}
    base = dict(score_fn=tr.score_fn, N=N, shape=(T, zd), time_eps=cfg.sde.sample_time_eps, label=None, denoise=True,
                device="cpu", probability_flow=False, snr=out["snr"], condition=None)
    for tag, B, kw in (("lv8", T, dict(predictor="eulermaruyama", corrector="langevin", corrector_steps=2)),
                       ("lv1", 1, dict(predictor="reversediffusion", corrector="langevin", corrector_steps=1)),
                       ("pndm8", T, dict(predictor="pndm", corrector=None, corrector_steps=1)),
                       ("pndm1", 1, dict(predictor="pndm", corrector=None, corrector_steps=1))):
        torch.manual_seed(99)
        with Recorder() as rec:
            res = tr.SDE.sample_discrete(num_samples=B, **base, **kw)
        draws = [d for _, d in rec.draws]
        out[tag + "_x0"] = draws[0]
        if len(draws) > 1:
            out[tag + "_noise"] = torch.stack(draws[1:], 0)
        out[tag + "_out"] = res
    # the shape rule: B neither 1 nor tokens -> torch's broadcasting error
    for pred, corr in (("ancestral", "langevin"), ("pndm", None)):
        try:
            tr.SDE.sample_discrete(num_samples=3, **base, predictor=pred, corrector=corr, corrector_steps=1)
            raise SystemExit("expected a broadcasting error")
        except RuntimeError as e:
            assert "must match the size of tensor" in str(e), e
    save("sampler_langevin_pndm", **out)


if __name__ == "__main__":
    main()
