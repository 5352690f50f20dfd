"""ldt_amd — MI355X-native sampling hot path of LDT (Latent Diffusion Transformer for point clouds).

Public surface mirrors the reference's classes on this path:
    Score (+ ConditionNet), Compressor, DiffusionVPSDE, Trainer (+ CompletionTrainer), dict2namespace
All arithmetic runs in hand-written HIP kernels (libldt_hip.so, C-ABI in include/ldt_hip.h).
"""
from .compressor import Compressor
from .condition import ConditionNet
from .config import airplane_config, dict2namespace, load_config
from .diffusion import (DiffusionBase, DiffusionGeometric, DiffusionSubVPSDE, DiffusionVESDE, DiffusionVPSDE,
                        make_diffusion)
from . import metrics
from .score import Score
from .trainer import CompletionTrainer, EMAWeights, Trainer

__all__ = ["Score", "Compressor", "ConditionNet", "DiffusionVPSDE", "DiffusionSubVPSDE", "DiffusionVESDE", "DiffusionGeometric", "DiffusionBase", "make_diffusion", "Trainer", "CompletionTrainer", "EMAWeights", "dict2namespace",
           "airplane_config", "load_config", "metrics"]
