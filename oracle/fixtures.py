"""Test infrastructure (like everything under oracle/): deterministic weight fixtures for the parity tests and bench.py's
`parity` block.  Nothing in ldt_amd/ imports this file.

`condition_score_head` — a WELL-CONDITIONED production-width Score without training.

With seeded random-init weights the reverse SDE inflates the latents to rms 400-600 (prod 1/sqrt(1-beta_i) = e^5: the random
Score's output does not depend on the SCALE of x — every path to the head goes through a LayerNorm — so nothing contracts), and
the decoder, fed such latents, is ill-conditioned even in fp32.  Scaling `ln_out.ln` so that eps_hat has rms 1
(the round-4 review's suggestion) was measured and makes it worse: final latents rms 493 instead of 378 (T = 32, N = 100, B = 2,
CPU oracle) — the eps_hat term is a drift the latents' own growth outruns either way.  What a trained denoiser has and a random
one lacks is an output component ALONG x (for data at the origin the exact answer is eps_hat = x / sigma_t).  The fixture adds
that component through the one linear path the architecture offers, ln_in -> residual stream -> final LayerNorm -> ln_out.ln:

    ln_out.ln.weight  +=  pinv(ln_in.weight)            (120 x 1024; factor 1.0, every other tensor as seeded)

so eps_hat ~ x / rms-ish(x) + the seeded network's own output (the 24 blocks still contribute their full residual branches:
projected through pinv(W_in) AND through the seeded head).  Measured on the CPU oracle with the airplane schedule: latents stay
at rms 0.9 -> 0.09 through the loop and end at rms 2.2 (N = 100, T = 32 or 256) / 0.12 (N = 1000, T = 32): the decoder's
operating range, and a stricter per-step test than the inflated run (a given absolute error in eps_hat is a far larger
relative error of latents of rms 0.1 than of latents of rms 400).
Reference: model/scorenet/score.py:110-151 (ln_in, ln_out), model/layers.py:240-248 (FinalLayer).
"""
import torch


def condition_score_head(sd):
    """-> a copy of the Score state_dict `sd` with ln_out.ln.weight += pinv(ln_in.weight) (fp32, computed once on the CPU;
    the GPU model and the oracle must both be given THIS dict so that they hold the same numbers)."""
    out = {k: v.detach().float().cpu().clone() for k, v in sd.items()}
    w_in = out["ln_in.weight"][:, :, 0].double()                   # Conv1d (hidden, z, 1)
    pinv = torch.linalg.pinv(w_in).float()                         # (z, hidden)
    out["ln_out.ln.weight"] = (out["ln_out.ln.weight"][:, :, 0] + pinv)[:, :, None].contiguous()
    return out
