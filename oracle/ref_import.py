"""TEST INFRASTRUCTURE ONLY — imports the upstream Python reference on CPU.

Runs ONLY in the build container where /root/reference exists (it never travels
to the GPU box).  Used by the `oracle/gen_*golden.py` scripts to capture golden vectors and by
`tests/test_host_logic.py::test_default_init_draws_same_weights_as_reference` (auto-skipped when /root/reference is absent).

The reference hard-codes `'cuda'` devices and imports packages that are not in
this image; this module installs the minimum shims to run it on CPU:

  * stub modules: torchvision(.models/.transforms), torchdiffeq.odeint, mitsuba,
    pointnet2_ops.pointnet2_utils.furthest_point_sample (-> our numpy FPS that
    follows the vendored twin model/functional/src/sampling/sampling.cu:86-167)
  * `'cuda'` -> `'cpu'` in Tensor.to/.cuda and in factory functions' device=
  * cfg.score.graphconv=False (trainer/Latent_SDE_Trainer.py:158 reads it; no
    shipped YAML defines it), cfg.log.save_path -> tmp dir.
"""
import argparse
import contextlib
import os
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

REF_ROOT = os.environ.get("LDT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model", "scorenet"))


def fps_numpy(xyz: np.ndarray, m: int, skip_near_origin: bool = True) -> np.ndarray:
    """FPS as sampling.cu:86-167 (start idx 0, dist init 1e38, running min, argmax with the 512-thread tie order: smaller
    (k % 512, k // 512) wins) + what the upstream pointnet2_ops kernel the reference really calls does in addition
    (restated from its published source, not vendored): points with |p|^2 <= 1e-3 are skipped — they neither
    update their distance nor can be selected.  Delegates to the oracle's restatement (one implementation)."""
    from oracle.ldt_oracle import fps as oracle_fps
    return oracle_fps(torch.from_numpy(np.asarray(xyz, dtype=np.float32)), int(m), skip_near_origin=skip_near_origin).numpy()


def _fps_numpy_twin_only(xyz: np.ndarray, m: int) -> np.ndarray:
    """(kept for reference: the vendored twin alone, no near-origin skip — what rounds 1-2 captured the goldens with)"""
    b, n, _ = xyz.shape
    out = np.zeros((b, m), dtype=np.int64)
    k = np.arange(n)
    tie_rank = (k % 512) * (n // 512 + 1) + k // 512
    for bi in range(b):
        p = xyz[bi].astype(np.float32)
        dist = np.full((n,), np.float32(1e38), dtype=np.float32)
        old = 0
        for j in range(1, m):
            dx = p[:, 0] - p[old, 0]
            dy = p[:, 1] - p[old, 1]
            dz = p[:, 2] - p[old, 2]
            d = (dx * dx + dy * dy) + dz * dz
            dist = np.minimum(d, dist)
            best = dist.max()
            cand = np.nonzero(dist == best)[0]
            old = int(cand[np.argmin(tie_rank[cand])])
            out[bi, j] = old
    return out


def _install_stubs():
    if "pointnet2_ops" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.modules["torchvision.transforms"] = tv.transforms
    td = types.ModuleType("torchdiffeq")
    td.odeint = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("torchdiffeq stub"))
    sys.modules["torchdiffeq"] = td
    sys.modules["mitsuba"] = types.ModuleType("mitsuba")
    p2 = types.ModuleType("pointnet2_ops")
    p2u = types.ModuleType("pointnet2_ops.pointnet2_utils")

    def furthest_point_sample(xyz, npoint):
        idx = fps_numpy(xyz.detach().cpu().numpy(), int(npoint))
        return torch.from_numpy(idx).to(torch.int32)

    p2u.furthest_point_sample = furthest_point_sample
    p2.pointnet2_utils = p2u
    sys.modules["pointnet2_ops"] = p2
    sys.modules["pointnet2_ops.pointnet2_utils"] = p2u


def _cpu(dev):
    if isinstance(dev, str) and dev.startswith("cuda"):
        return "cpu"
    if isinstance(dev, torch.device) and dev.type == "cuda":
        return torch.device("cpu")
    return dev


_PATCHED = False


def _patch_cuda_to_cpu():
    global _PATCHED
    if _PATCHED:
        return
    _PATCHED = True
    orig_to = torch.Tensor.to

    def to(self, *args, **kwargs):
        args = tuple(_cpu(a) for a in args)
        if "device" in kwargs:
            kwargs["device"] = _cpu(kwargs["device"])
        return orig_to(self, *args, **kwargs)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig_mod_to = torch.nn.Module.to

    def mod_to(self, *args, **kwargs):
        args = tuple(_cpu(a) for a in args)
        if "device" in kwargs:
            kwargs["device"] = _cpu(kwargs["device"])
        return orig_mod_to(self, *args, **kwargs)

    torch.nn.Module.to = mod_to
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for name in ["tensor", "ones", "zeros", "randn", "rand", "linspace", "arange", "full",
                 "ones_like", "zeros_like", "randn_like", "rand_like", "empty"]:
        orig = getattr(torch, name)

        def wrap(orig):
            def f(*args, **kwargs):
                if "device" in kwargs:
                    kwargs["device"] = _cpu(kwargs["device"])
                return orig(*args, **kwargs)
            return f

        setattr(torch, name, wrap(orig))


def setup():
    """Install shims and put the reference on sys.path."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    _install_stubs()
    _patch_cuda_to_cpu()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def dict2ns(d):
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2ns(v) if isinstance(v, dict) else v)
    return ns


def load_airplane_cfg(**overrides):
    """Shipped airplane YAML (experiments/Latent_Diffusion_Trainer/airplane/config.yaml)
    as nested Namespace + the two fix-ups the reference needs to run."""
    path = os.path.join(REF_ROOT, "experiments", "Latent_Diffusion_Trainer", "airplane", "config.yaml")
    with open(path) as f:
        raw = yaml.safe_load(f)
    for dotted, val in overrides.items():
        sect, key = dotted.split(".")
        raw[sect][key] = val
    cfg = dict2ns(raw)
    cfg.score.graphconv = False
    cfg.log.save_path = tempfile.mkdtemp(prefix="ldt_ref_log_")
    return cfg


@contextlib.contextmanager
def quiet():
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        yield
