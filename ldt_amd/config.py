"""Configuration objects for the hot path.

The reference turns `experiments/<trainer>/<dataset>/config.yaml` into nested
`argparse.Namespace`s (tools/io.py:13-21, train_Latent_Diffusion.py:115-120); the
classes here accept those Namespaces unchanged.  `airplane_config()` restates the
shipped airplane hyper-parameters that the path reads
(experiments/Latent_Diffusion_Trainer/airplane/config.yaml:44-113) so benches/tests
do not need the reference tree.
"""
import argparse
import copy

import yaml


def dict2namespace(config):
    """Nested dict -> nested Namespace (same contract as tools/io.py:13-21)."""
    ns = argparse.Namespace()
    for key, value in config.items():
        setattr(ns, key, dict2namespace(value) if isinstance(value, dict) else value)
    return ns


def namespace2dict(ns):
    return {k: (namespace2dict(v) if isinstance(v, argparse.Namespace) else v) for k, v in vars(ns).items()}


_AIRPLANE = {
    "data": {"num_categorys": 1, "tr_max_sample_points": 2048, "te_max_sample_points": 2048,
             "batch_size": 64, "test_batch_size": 64},
    "opt": {"ema_decay": 0.9999},
    "common": {"num_points": 2048, "seed": 0},
    "score": {"num_steps": 1000, "z_dim": 120, "z_scale": 32, "hidden_size": 1024, "num_heads": 16,
              "num_blocks": 24, "num_categorys": 1, "c_dim": 0.0, "t_dim": 1024, "dropout": 0.0,
              "norm": "layer_norm", "learn_sigma": False, "act": "swish", "unet": False, "AdaLN": True,
              "condition": False},
    "compressor": {"outsize": 2048, "max_outputs": 2048, "input_dim": 3, "z_dim": 20, "z_scales": 32,
                   "p_dim": 256, "n_layers": 6, "hidden_dim": 128, "num_heads": 4, "activation": "swish",
                   "encoder_dropout_p": 0.0, "decoder_dropout_p": 0.0, "norm": "layer_norm", "neighbors": 128,
                   "encoder_layers": 2, "mlp_ratio": 4.0, "min_sigma": -30, "cluster_norm": "anchor",
                   "norm_input": False, "pre_group": False, "decoder_act": None, "ActNorm": True, "AdaLN": True,
                   "pos_embedding": "center", "class_condition": False},
    "sde": {"beta_start": 0.1, "beta_end": 20, "sde_type": "vpsde", "sigma2_0": 0, "time_eps": 0.01,
            "ode_tol": 1e-5, "sample_time_eps": 1e-6, "sample_mode": "discrete", "predictor": "ancestral",
            "corrector": None, "train_N": 1000, "sample_N": 1000, "snr": 0.01, "corrector_steps": 1,
            "denoise": True, "probability_flow": False, "alpha": 1.0},
}


def airplane_config(latent_tokens=None, sample_N=None, **overrides):
    """Shipped airplane config; `latent_tokens` sets score.z_scale = compressor.z_scales (BASELINE uses 256,
    the shipped YAML 32).  `overrides` are 'section.key'=value."""
    d = copy.deepcopy(_AIRPLANE)
    if latent_tokens is not None:
        d["score"]["z_scale"] = latent_tokens
        d["compressor"]["z_scales"] = latent_tokens
    if sample_N is not None:
        d["sde"]["sample_N"] = sample_N
    for dotted, v in overrides.items():
        sect, key = dotted.split(".")
        d[sect][key] = v
    cfg = dict2namespace(d)
    cfg.score.graphconv = False          # read at trainer/Latent_SDE_Trainer.py:158; no shipped YAML defines it (Q6)
    return cfg


def load_config(path):
    """YAML file -> Namespace, with the Q6 default."""
    with open(path) as f:
        cfg = dict2namespace(yaml.safe_load(f))
    if hasattr(cfg, "score") and not hasattr(cfg.score, "graphconv"):
        cfg.score.graphconv = False
    return cfg
