"""GPU (-m gpu): the REAL collective path as far as one GPU allows — `python -m torch.distributed.run --nproc-per-node 1
bench.py ...` in a fresh child process: init_process_group("nccl") (RCCL), the x0/key agreement check and the final
all_gather_into_tensor on device tensors, the barriers and the max-over-ranks reduction all execute once.

This file sorts first among the GPU tests on purpose: the child is started before THIS process has touched the GPU
(no HIP call, no torch.cuda.is_available() at import or here), and the launcher itself never initialises the GPU before it
spawns the worker."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_world1_runs_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sde-steps", "25", "--steps", "1",
           "--warmup", "0", "--batch-per-gpu", "8", "--tokens", "32", "--no-cpu-baseline", "--no-extras", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["value"] > 0
    assert line["config"]["collective"].startswith("nccl"), line["config"]
    assert line["config"]["global_batch"] == 8


def test_bench_self_launch_path_world1():
    """`python bench.py --gpus 1 --force-launch`: the parent starts torch.distributed.run as a child before touching the GPU (the path
    `--gpus N > 1` takes without a launcher) and relays its JSON line; also runs --config c5 (BASELINE configs[4]) once."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-launch", "--config", "c5", "--sde-steps", "25", "--steps", "1",
           "--warmup", "0", "--batch-per-gpu", "8", "--no-cpu-baseline", "--no-extras", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["name"] == "c5"
    assert line["config"]["collective"] == "nccl, world 1", line["config"]
    assert line["config"]["global_batch"] == 8 and line["config"]["latent_tokens"] == 32
