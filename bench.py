#!/usr/bin/env python3
"""Bench of the LDT sampling hot path on MI355X (contract: see the task's bench.py section).

One "step" = one `Trainer.sample(B)` call = B shapes x N_sde reverse-SDE steps of the 24-block Score
Transformer + Compressor decode to 2048 points (reference: trainer/Latent_SDE_Trainer.py:143-165, the
"Sample rate" the reference prints at :178-181,206).  Workload = BASELINE.json configs[1]:
ShapeNet-airplane shapes, batch 64 per GPU, 256 latent tokens x 120, 1000 ancestral steps, bf16 MFMA.
Weights are seeded random-init (no checkpoints reachable), noise is device Philox: data = synthetic.

    python bench.py                                  # 1 GPU, K=2 timed sample() calls after W=1 warm-up
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line: metric shapes/sec (whole job), + `roofline` (dominant kernel, HIP-event timed
inside this process) + `cpu_baseline` (the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec, same table


_T0 = time.time()


def log(msg):
    """progress to stderr (the JSON line on stdout stays alone)"""
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def host_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota)."""
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=256, help="latent tokens (BASELINE: 256; shipped YAML: 32)")
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--sde-steps", type=int, default=1000)
    ap.add_argument("--cpu-steps", type=int, default=12, help="SDE steps of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def score_flops_per_sample_step(cfg):
    """MACs per sample-step x 2 (token-linear 12 D^2/token/block, attention 2 T^2 D/block,
    ln_in + ln_out 2 T z D; the batch-shared AdaLN/time MLP are excluded)."""
    D, T, L, z = cfg.score.hidden_size, cfg.score.z_scale, cfg.score.num_blocks, cfg.score.z_dim
    macs = L * (12 * D * D * T + 2 * T * T * D) + 2 * T * z * D
    return 2.0 * macs


def measured_traffic(symbol):
    """HBM bytes per launch of `symbol` from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate passes; gfx950: FETCH_SIZE doubled, units KiB — MI355X_MICROARCH.md §HBM).
    None when no measurement of this kernel at this workload has been committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(symbol, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def roofline_pass(trainer, cfg, B, reps=3):
    """HIP-event timing of every launch of one Score forward (same stream), averaged over `reps` passes."""
    from ldt_amd import _lib, ops
    model = trainer.model
    dev = next(model.parameters()).device
    T, z, D, F = cfg.score.z_scale, cfg.score.z_dim, cfg.score.hidden_size, 4 * cfg.score.hidden_size
    t = torch.linspace(1.0, 1e-6, 8).to(dev)
    _, mod = model.time_table(t)
    folded = model.can_fold(B, T)                       # the same decision sample_discrete takes for this (B, T)
    plan = model.plan(B, T, mod, model.n_mod, 0, fold=model.fold_table(mod) if folded else None)
    x = torch.randn(B, T, z, device=dev)
    out = torch.empty_like(x)
    ncls = len(_lib.PROF_CLASSES)
    ms = (ctypes.c_float * ncls)()
    cnt = (ctypes.c_int32 * ncls)()
    tot_ms, tot_cnt = [0.0] * ncls, [0] * ncls
    for r in range(reps + 1):
        _lib.check(_lib.lib().ldt_score_forward_profile(ctypes.byref(plan), x.data_ptr(), out.data_ptr(), None, ms, cnt,
                                                        ops.stream_ptr()), "ldt_score_forward_profile")
        if r == 0:
            continue                                    # first pass warms caches / clocks
        for c in range(ncls):
            tot_ms[c] += ms[c]; tot_cnt[c] += cnt[c]
    M = B * T
    flops = {"gemm_qkv": 2.0 * M * D * 3 * D, "gemm_gelu": 2.0 * M * D * F,
             "gemm_resid": (2.0 * M * D * D + 2.0 * M * F * D) / 2.0,        # average of fc_o and mlp.out launches
             "attention": 4.0 * M * T * D}
    kernels = {}
    for c, name in enumerate(_lib.PROF_CLASSES):
        if tot_cnt[c] == 0:
            continue
        avg = tot_ms[c] / tot_cnt[c]
        k = {"launches_per_forward": tot_cnt[c] // reps, "avg_ms": round(avg, 5), "ms_per_forward": round(tot_ms[c] / reps, 4)}
        if name in flops:
            k["tflops"] = round(flops[name] / (avg * 1e-3) / 1e12, 2)
        kernels[name] = k
    dom = max(("gemm_qkv", "gemm_gelu", "gemm_resid"), key=lambda n: kernels[n]["ms_per_forward"])
    # symbols as rocprofv3 prints them: <epilogue id (1 BF16, 2 GELU_BF16, 4 RESID_F32), LN folding (0 none, 1 producer:
    # also emits x(1+scale) + row statistics, 2 consumer: applies the LayerNorm in its epilogue)>
    sym = {"gemm_qkv": "gemm_bf16_nt_256_kernel<1, %d>" % (2 if folded else 0), "gemm_gelu": "gemm_bf16_nt_256_kernel<2, %d>" % (2 if folded else 0),
           "gemm_resid": "gemm_bf16_nt_256_kernel<4, %d>" % (1 if folded else 0)}[dom]
    ach = kernels[dom]["tflops"]
    roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": measured_traffic(sym), "kernel": sym,
            "flops_per_launch": flops[dom], "avg_launch_ms": kernels[dom]["avg_ms"], "ln_folding": folded}
    a = kernels["attention"]
    abytes = 4.0 * M * D * 2                                               # read Q,K,V + write O in bf16
    gbs = abytes / (a["avg_ms"] * 1e-3) / 1e9
    attn = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": measured_traffic("attn_fwd_kernel<64, false>"), "kernel": "attn_fwd_kernel<64, false>",
            "bytes_per_launch": abytes, "avg_launch_ms": a["avg_ms"]}
    return roof, attn, kernels


def cpu_baseline(trainer, cfg, n_steps):
    """The oracle (CPU restatement of the reference, kind "port") on this box's host cores: B=4 shapes,
    same T and schedule, first `n_steps` of the 1000-step loop + one decode; per-step cost is
    step-invariant, so shapes/s(1000 steps) = B / (1000 * t_step + t_decode)."""
    from oracle import ldt_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    B = 4
    sd_s = {k: v.detach().float().cpu() for k, v in trainer.model.state_dict().items()}
    sd_c = {k: v.detach().float().cpu() for k, v in trainer.compressor.state_dict().items()}
    x0, noises = O.draw_noises(1234, B, cfg.score.z_scale, cfg.score.z_dim, n_steps)
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda x, t: O.score_forward(sd_s, cfg.score, x, t))
    log("cpu baseline: %d threads, B=%d, %d steps" % (cores, B, n_steps))
    with torch.no_grad():
        t0 = time.time()
        eps = O.sample_discrete(sde, fn, x0, noises, cfg.sde.sample_N, max_steps=n_steps)
        t1 = time.time()
        O.compressor_decode(sd_c, cfg.compressor, eps)
        t2 = time.time()
    t_step = (t1 - t0) / n_steps
    per_call = cfg.sde.sample_N * t_step + (t2 - t1)
    return {"value": B / per_call, "unit": "shapes/sec", "cores": cores, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32 restatement of the reference, %d threads): B=%d shapes, T=%d tokens, first %d "
                      "of %d ancestral steps (%.3f s/step) + 1 decode (%.2f s); extrapolated linearly to %d steps"
                      % (cores, B, cfg.score.z_scale, n_steps, cfg.sde.sample_N, t_step, t2 - t1, cfg.sde.sample_N)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", init_method="env://")       # RCCL over xGMI; MASTER_* / RANK from the launcher
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    import ldt_amd
    cfg = ldt_amd.airplane_config(latent_tokens=args.tokens, sample_N=args.sde_steps)
    torch.manual_seed(0)                                 # same weights + same CPU generator stream on every rank
    score = ldt_amd.Score(cfg.score)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    trainer = ldt_amd.Trainer(cfg, score, comp, device)
    B = args.batch_per_gpu * world

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    log("model built on %s (world %d), B=%d T=%d N=%d" % (device, world, B, args.tokens, args.sde_steps))
    for i in range(args.warmup):
        trainer.sample(B)
        torch.cuda.synchronize()
        log("warmup %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pts, eps = trainer.sample(B)
    barrier()
    dt = time.perf_counter() - t0
    log("timed region: %.2f s for %d sample() calls" % (dt, args.steps))
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert pts.shape == (B, cfg.data.tr_max_sample_points, 3) and bool(torch.isfinite(eps).all())

    if rank == 0:
        value = B * args.steps / dt
        flops_call = score_flops_per_sample_step(cfg) * args.sde_steps * B
        line = {
            "metric": "shapes/sec (2048-pt, %d-step SDE sample)" % args.sde_steps, "value": round(value, 4),
            "unit": "shapes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ShapeNet-airplane sampling, batch %d/GPU, %d latent tokens x %d, "
                                   "%d ancestral SDE steps (24-block d=1024 Score) + decode to %d points"
                                   % (args.batch_per_gpu, args.tokens, cfg.score.z_dim, args.sde_steps, cfg.data.tr_max_sample_points),
                       "global_batch": B, "batch_per_gpu": args.batch_per_gpu, "latent_tokens": args.tokens,
                       "sde_steps": args.sde_steps, "points": cfg.data.tr_max_sample_points,
                       "parallelism": "dp%d batch slices, one all-gather" % world},
            "achieved_tflops_whole_job": round(flops_call * args.steps / dt / 1e12, 1),
        }
        if not args.no_roofline:
            roof, attn, kernels = roofline_pass(trainer, cfg, args.batch_per_gpu)
            log("roofline pass done: %s" % json.dumps(kernels))
            line["roofline"] = roof
            line["roofline_attention"] = attn
            line["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(trainer, cfg, args.cpu_steps)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
