"""CPU, world_size 2 over gloo: the N>1 path of Trainer.sample — contiguous batch slices by global sample index,
identical full-batch x0 draw on every rank, sample_offset for the noise streams, zero padding of the last rank,
and the single all-gather at the end (ldt_amd/dist.py, ldt_amd/trainer.py).  The HIP kernels are replaced by a
deterministic CPU stand-in here (this test is about the sharding logic; kernel parity is the -m gpu suite)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_sample_discrete(calls):
    def f(score_fn, num_samples, N, predictor, corrector, corrector_steps, shape, time_eps, probability_flow, denoise,
          snr, device, condition=None, label=None, print_steps=None, *, x0=None, noise=None, sample_offset=0, seed=None,
          use_graph=None, record=None, global_batch=None, trajectory=None):
        calls.append(dict(num_samples=num_samples, sample_offset=sample_offset, seed=seed, x0_rows=x0.shape[0],
                          global_batch=global_batch))
        idx = torch.arange(sample_offset, sample_offset + num_samples, dtype=torch.float32)[:, None, None]
        out = x0 * 2.0 + idx + float(seed % 7)             # depends on the GLOBAL sample index and the shared seed
        # per-sample conditioning must have followed its samples to this rank (trainer.py: label / tuple / dict branches)
        if label is not None:
            assert label.shape[0] == num_samples
            out = out + 10.0 * label.float()[:, None, None]
        if isinstance(condition, (tuple, list)):
            pts_c, img_c = condition
            assert pts_c.shape[0] == num_samples and img_c.shape[0] == num_samples
            out = out + pts_c.float().mean((1, 2))[:, None, None] + 3.0 * img_c.float().mean(1)[:, None, None]
        elif isinstance(condition, dict):
            assert condition["pts"].shape[0] == num_samples and condition["img"].shape[0] == num_samples and condition["flag"] == "keep"
            out = out + condition["pts"].float().mean((1, 2))[:, None, None] - condition["img"].float().mean((1, 2, 3))[:, None, None]
        return out
    return f


def _fake_decode(shape, given_eps=None):
    return given_eps[:, :3, :4].reshape(shape[0], 1, 12).repeat(1, shape[1], 1)[..., :3] * 0.5


def _build(tiny):
    import ldt_amd
    score = ldt_amd.Score(tiny.score)
    comp = ldt_amd.Compressor(tiny.compressor)
    tr = ldt_amd.Trainer(tiny, score, comp, "cpu")
    calls = []
    tr.SDE.sample_discrete = _fake_sample_discrete(calls)
    tr.compressor.sample = _fake_decode
    return tr, calls


def _tiny():
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import to_ns
    with open(os.path.join(ROOT, "tests", "golden", "tiny_cfg.json")) as f:
        return to_ns(json.load(f))


def _conditioning(kind, B):
    g = torch.Generator().manual_seed(5)
    if kind == "label":
        return dict(label=torch.randint(0, 9, (B,), generator=g))
    if kind == "tuple":
        return dict(condition=(torch.randn(B, 16, 5, generator=g), torch.randn(B, 24, generator=g)))
    if kind == "dict":
        return dict(condition={"pts": torch.randn(B, 32, 3, generator=g), "img": torch.randn(B, 3, 8, 8, generator=g), "flag": "keep"})
    return {}


def _worker(rank, world, port, B, out_dir, kind, bad_seed):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(123)                               # common_init: same seed on every rank
        tr, calls = _build(_tiny())
        torch.manual_seed(77 + (rank if bad_seed else 0))
        err = None
        try:
            pts, eps = tr.sample(B, **_conditioning(kind, B))
        except RuntimeError as e:
            if not bad_seed:
                raise
            pts = eps = None
            err = str(e)
        torch.save(dict(pts=pts, eps=eps, calls=calls, err=err), os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,kind", [(4, "none"), (5, "none"), (5, "label"), (4, "tuple"), (5, "tuple"), (5, "dict")])
def test_world2_matches_single_process(tmp_path, B, kind):
    sys.path.insert(0, ROOT)
    # single-process expectation
    torch.manual_seed(123)
    tr, calls1 = _build(_tiny())
    torch.manual_seed(77)
    pts1, eps1 = tr.sample(B, **_conditioning(kind, B))
    assert calls1[0]["num_samples"] == B and calls1[0]["sample_offset"] == 0 and calls1[0]["global_batch"] is None
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path), kind, False), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    per = (B + 1) // 2
    assert r0["calls"][0] == dict(num_samples=per, sample_offset=0, seed=r0["calls"][0]["seed"], x0_rows=per, global_batch=B)
    assert r1["calls"][0]["sample_offset"] == per and r1["calls"][0]["num_samples"] == per     # padded to equal shapes
    assert r0["calls"][0]["seed"] == r1["calls"][0]["seed"] == calls1[0]["seed"]                  # shared Philox key
    for r in (r0, r1):                                       # every rank holds the full, identical result
        assert r["pts"].shape == pts1.shape and r["eps"].shape == eps1.shape
        assert torch.equal(r["eps"], eps1) and torch.equal(r["pts"], pts1)


def test_world2_detects_mismatched_generators(tmp_path):
    """Ranks that seeded their CPU generators differently would silently break world-size invariance (every rank draws
    the full-batch x0 and keeps its rows): the 16-byte check in ldt_amd/dist.py turns that into an error on every rank."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path), "none", True), nprocs=2, join=True)
    for r in range(2):
        rec = torch.load(os.path.join(tmp_path, "rank%d.pt" % r), weights_only=False)
        assert rec["err"] is not None and "different x0" in rec["err"] and not rec["calls"]
