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


# ---- sharded LangevinCorrector step: the batch-mean norms are the path's one cross-sample quantity ----------------
def _install_cpu_ops(setter=None):
    """CPU stand-ins for the four C-ABI wrappers `langevin_update` calls (same arithmetic as csrc/samplers.hip)."""
    from ldt_amd import ops

    def batch_norm_sum(x, n_valid, per, norms, sum_out):
        sum_out[0] = x[:n_valid].reshape(n_valid, -1).float().norm(dim=1).sum()

    def langevin_coef(sums, n_total, snr, std_t, coef):
        grad_norm = (sums[0] / n_total) / std_t
        noise_norm = sums[1] / n_total
        step = (snr * noise_norm / grad_norm) ** 2 * 2.0
        coef[0], coef[1], coef[2], coef[3] = 1.0, -step / std_t, torch.sqrt(step * 2.0), 0.0

    def sampler_step(x, params, coef, step, mode, noise=None, x_mean_out=None, **kw):
        assert mode == 1
        xm = coef[0] * x + coef[1] * params
        if x_mean_out is not None:
            x_mean_out.copy_(xm)
        return xm + coef[2] * noise

    setter = setter or (lambda name, fn: setattr(ops, name, fn))
    for name, fn in (("batch_norm_sum", batch_norm_sum), ("langevin_coef", langevin_coef), ("sampler_step", sampler_step)):
        setter(name, fn)


def _langevin_inputs(B):
    g = torch.Generator().manual_seed(B)
    return (torch.randn(B, 8, 12, generator=g), torch.randn(B, 8, 12, generator=g) * 3.0, torch.randn(B, 8, 12, generator=g))


def _langevin_rows(x, params, z, lo, per, n_total, sharded):
    from ldt_amd.diffusion import langevin_update
    from ldt_amd.trainer import _rows
    xs, ps, zs = (_rows(t, lo, lo + per, per) for t in (x, params, z))
    n_valid = max(0, min(per, n_total - lo))
    xm = torch.empty_like(xs)
    out = langevin_update(xs, ps, zs, xm, 0.7, 0.16, n_total, n_valid, sharded, (torch.zeros(4), torch.zeros(2), torch.zeros(per)))
    return out[:n_valid], xm[:n_valid]


def _langevin_worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _install_cpu_ops()
        from ldt_amd import dist as ldist
        lo, hi, per = ldist.shard_bounds(B, rank, world)
        out, xm = _langevin_rows(*_langevin_inputs(B), lo, per, B, True)
        torch.save(dict(out=out, xm=xm), os.path.join(out_dir, "lv%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 5, 1])          # even split; last rank padded (3 + 2 real rows); rank 1 holds NO real row
def test_world2_langevin_step_matches_single_process(tmp_path, monkeypatch, B):
    sys.path.insert(0, ROOT)
    from ldt_amd import ops
    _install_cpu_ops(lambda name, fn: monkeypatch.setattr(ops, name, fn))      # (undone after the test)
    x, params, z = _langevin_inputs(B)
    want, want_m = _langevin_rows(x, params, z, 0, B, B, False)
    port = _free_port()
    mp.spawn(_langevin_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    parts = [torch.load(os.path.join(tmp_path, "lv%d.pt" % r), weights_only=False) for r in range(2)]
    got, got_m = torch.cat([p["out"] for p in parts]), torch.cat([p["xm"] for p in parts])
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6) and torch.allclose(got_m, want_m, rtol=1e-6, atol=1e-6)
