import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (arrays dict as torch tensors, {prefix: state_dict}) from tests/golden/<name>.npz"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs, sds = {}, {}
    for k in z.files:
        v = torch.from_numpy(np.asarray(z[k]))
        if "::" in k:
            p, n = k.split("::", 1)
            sds.setdefault(p, {})[n] = v
        else:
            arrs[k] = v
    return arrs, sds


def to_ns(d):
    return SimpleNamespace(**{k: (to_ns(v) if isinstance(v, dict) else v) for k, v in d.items()})


@pytest.fixture(scope="session")
def tiny_cfg():
    with open(os.path.join(GOLDEN, "tiny_cfg.json")) as f:
        return to_ns(json.load(f))


def rel_mse(a, b):
    a = a.double(); b = b.double()
    return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-300))


def host_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota) — the GPU boxes expose every host core in
    the affinity mask but cap the container at a quota; oversubscribed torch threads run the CPU oracle 10x slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + p // 2) // p))
        except (OSError, ValueError):
            pass
    return n
