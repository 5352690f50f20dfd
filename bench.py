#!/usr/bin/env python3
"""Bench of the LDT sampling hot path on MI355X (contract: see the task's bench.py section).

One "step" = one `Trainer.sample(B)` call = B shapes x N_sde reverse-SDE steps of the 24-block Score
Transformer + Compressor decode to 2048 points (reference: trainer/Latent_SDE_Trainer.py:143-165, the
"Sample rate" the reference prints at :178-181,206).  Workload = BASELINE.json configs[1]:
ShapeNet-airplane shapes, batch 64 per GPU, 256 latent tokens x 120, 1000 ancestral steps, bf16 MFMA.
Weights are seeded random-init (no checkpoints reachable), noise is device Philox: data = synthetic.

    python bench.py                                  # 1 GPU, K=2 timed sample() calls after W=1 warm-up
    python bench.py --gpus N --steps K --warmup W    # N > 1 without a launcher: starts its own ranks (a child `torch.distributed.run`)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --config c5 [--gpus N]           # BASELINE configs[4]: ViPC-conditioned sampling, 32 shapes per GPU, 32 tokens

Rank 0 prints ONE JSON line: metric shapes/sec (whole job) and, at N=1,
  roofline      dominant kernel of the Score forward, HIP-event timed inside this process, with ITS OWN PMC traffic / MFMA utilisation
                (+ every kernel class in `roofline_kernels`, each against the roofline that bounds it; `roofline_attention` = HBM GB/s);
                also carries every secondary figure of the line as an `x_...` scalar (the driver's record keeps only the scalar members
                of the contract keys)
  cpu_baseline  config C1 EXACTLY (B=4, T=256, N=100, decode included) on the CPU oracle, x 1/10
  parity        top level: the FIXED-BAR end-to-end fixture at production width (C1's shape, well-conditioned weights): per-step / final
                latents, decoded points, Chamfer, pass; `c1_latents`: the C1 run the cpu_baseline times, same injected noise (latents to
                1e-4); `ill_conditioned_random_weights`: that run's decoded-cloud figures, informational
  extra         the headline workload on the LayerNorm-kernel path (the LN-fold guard's fallback), BASELINE configs[3] (Compressor
                B=1024), configs[4]'s per-GPU share (ViPC, B=32, T=32), shipped T=32
  rccl_ranks_seen  ranks that met in a one-int RCCL all-gather before the model was built (0: no launcher / single process)
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import socket
import subprocess
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c5"), default="c2",
                    help="c2 (default): BASELINE configs[1]/[2], unconditional, 64 shapes per GPU, 256 tokens; "
                         "c5: BASELINE configs[4], ViPC-conditioned, 32 shapes per GPU, 32 tokens")
    ap.add_argument("--tokens", type=int, default=None, help="latent tokens (c2: 256 = BASELINE, shipped YAML: 32; c5: 32)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="shapes per GPU (c2: 64; c5: 32)")
    ap.add_argument("--force-launch", action="store_true",
                    help="start the ranks through a child `torch.distributed.run` even at --gpus 1 (the path --gpus N > 1 takes "
                         "when no launcher started this process)")
    ap.add_argument("--sde-steps", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the C1 oracle run (cpu_baseline + parity)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4] / T=32 blocks")
    ap.add_argument("--budget-s", type=float, default=540.0, help="extras are skipped once the run is older than this")
    args = ap.parse_args(argv)
    if args.tokens is None:
        args.tokens = 256 if args.config == "c2" else 32
    if args.batch_per_gpu is None:
        args.batch_per_gpu = 64 if args.config == "c2" else 32
    return args


RENDEZVOUS_RETRY_WINDOW_S = 120.0      # self_launch repeats a launch only when the child failed this early on the store's bind


def needs_self_launch(args, environ):
    """True when this process must start the ranks itself: more than one GPU asked for (or --force-launch) and no launcher
    (torch.distributed.run sets RANK) started us."""
    return "RANK" not in environ and (args.gpus > 1 or args.force_launch)


def launcher_argv(argv, gpus, port):
    """The command `python bench.py --gpus N ...` runs as a CHILD when no launcher started it: one rank per GPU of this node
    under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    rest = [a for a in argv if a != "--force-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def self_launch(args, argv, attempts=3):
    """Start the ranks as a child process and relay its output.  Runs BEFORE anything in this process touches the GPU (never exec
    from a process that initialised HIP); the child's single JSON line passes through on stdout, its return code is ours.
    The rendezvous port is found by bind(0) / close, which another job on the box can take before the child binds it: the child's
    stderr is relayed line by line and, when it failed on an address already in use, the launch is repeated on a fresh port."""
    import threading
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rc = 1
    for attempt in range(attempts):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = launcher_argv(argv, args.gpus, port)
        log("no launcher in the environment: starting %d rank(s): %s" % (args.gpus, " ".join(cmd)))
        t_start = time.time()
        child = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE)     # bytes: an undecodable byte must not end the relay
        in_use = []

        def relay():
            for raw in child.stderr:
                ln = raw.decode("utf-8", errors="replace")
                sys.stderr.write(ln)
                # only the rendezvous store's own bind failure counts (c10d TCPStore / the elastic agent), not any later socket error
                if ("EADDRINUSE" in ln or "ddress already in use" in ln) and any(k in ln for k in ("TCPStore", "c10d", "rendezvous", "store", "%d" % port)):
                    in_use.append(ln)
            sys.stderr.flush()

        th = threading.Thread(target=relay, daemon=True)
        th.start()
        rc = child.wait()
        th.join()                                                         # the pipe is at EOF once the child is gone
        early = time.time() - t_start < RENDEZVOUS_RETRY_WINDOW_S         # a bind failure ends the child before any rank has run
        if rc == 0 or not in_use or not early:
            return rc
        log("rendezvous port %d was taken before the child bound it (attempt %d of %d)" % (port, attempt + 1, attempts))
    return rc


def score_flops_per_sample_step(cfg):
    """MACs per sample-step x 2 (token-linear 12 D^2/token/block, attention 2 T^2 D/block,
    ln_in + ln_out 2 T z D; the batch-shared AdaLN/time MLP are excluded)."""
    D, T, L, z = cfg.score.hidden_size, cfg.score.z_scale, cfg.score.num_blocks, cfg.score.z_dim
    macs = L * (12 * D * D * T + 2 * T * T * D) + 2 * T * z * D
    return 2.0 * macs


def csrc_sha():
    """sha256 (16 hex) over the kernel sources: ties a PMC traffic measurement to the code it was collected on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ldt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ldt_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(symbol):
    """(HBM bytes per launch of `symbol`, provenance) from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE in separate passes; gfx950: FETCH_SIZE doubled, units KiB — MI355X_MICROARCH.md §HBM;
    reduced by tools/reduce_pmc.py).  The value is None when no measurement of this kernel exists OR the kernel
    sources changed since it was collected (the provenance says which)."""
    prov = {"file": "profiles/traffic.json", "current_csrc_sha": csrc_sha()}
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get(symbol)
    except (OSError, ValueError):
        e = None
    if not e:
        prov["status"] = "no measurement of this kernel committed"
        return None, prov
    prov.update(source=e.get("source"), collected_on_csrc_sha=e.get("csrc_sha"), note=e.get("note"))
    if e.get("csrc_sha") != prov["current_csrc_sha"]:
        prov["status"] = "stale: kernel sources changed since the PMC passes (value withheld: %d)" % e["hbm_bytes_per_launch"]
        return None, prov
    prov["status"] = "current"
    return e["hbm_bytes_per_launch"], prov


def measured_mfma_util(symbol):
    """MFMA utilisation of `symbol` from the committed PMC pass (profiles/mfma_util.json, tools/reduce_pmc_mfma.py):
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).  None when absent or collected on other kernel sources."""
    try:
        with open(os.path.join(ROOT, "profiles", "mfma_util.json")) as f:
            e = json.load(f).get(symbol)
    except (OSError, ValueError):
        e = None
    if not e or e.get("csrc_sha") != csrc_sha():
        return None, ("no measurement committed" if not e else "stale: kernel sources changed since the PMC pass (value withheld: %.3f)" % e["mfma_util"])
    return e["mfma_util"], e.get("source")


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
    # algorithmic work per launch: flops, and HBM bytes for the kernels whose roofline is HBM.
    # fc_o at K = hidden is 177 flop/B (ridge ~310): bytes = Ob in (bf16) + x read + x write (fp32) + x(1+scale) out (bf16) + W
    flops = {"gemm_qkv": 2.0 * M * D * 3 * D, "gemm_gelu": 2.0 * M * D * F, "gemm_o": 2.0 * M * D * D, "gemm_dn": 2.0 * M * F * D,
             "attention": 4.0 * M * T * D}
    xs_out = 2.0 * M * D if folded else 0.0
    hbm_bytes = {"gemm_o": 2.0 * M * D + 8.0 * M * D + xs_out + 2.0 * D * D, "attention": 4.0 * M * D * 2,
                 "ln_modulate": 4.0 * M * D + 2.0 * M * D}
    fold_id = {"gemm_qkv": 2, "gemm_gelu": 2, "gemm_o": 1, "gemm_dn": 1}
    epi_id = {"gemm_qkv": 1, "gemm_gelu": 2, "gemm_o": 4, "gemm_dn": 4}
    kernels, roofs = {}, {}
    # QKV projection + self-attention as ONE launch (256-token samples: gemm_qkv_attn256_kernel): no attention launches; the class carries both flops
    fused_attn = tot_cnt[_lib.PROF_CLASSES.index("attention")] == 0 and tot_cnt[_lib.PROF_CLASSES.index("gemm_qkv")] > 0
    if fused_attn:
        flops["gemm_qkv"] += flops["attention"]
    for c, name in enumerate(_lib.PROF_CLASSES):
        if tot_cnt[c] == 0:
            continue
        avg = tot_ms[c] / tot_cnt[c]
        k = {"launches_per_forward": tot_cnt[c] // reps, "avg_ms": round(avg, 5), "ms_per_forward": round(tot_ms[c] / reps, 4)}
        if name in flops:
            k["tflops"] = round(flops[name] / (avg * 1e-3) / 1e12, 2)
        if name in hbm_bytes:
            k["gbps"] = round(hbm_bytes[name] / (avg * 1e-3) / 1e9, 1)
        kernels[name] = k
        # rocprofv3 symbol: <epilogue id, LN folding (1 producer, 2 consumer), residual rows through the operand ring (one tile per workgroup),
        # W from registers (0: not shipped), K >= 2048 name tag of the one-tile kernels (mlp.out has a symbol of its own since round 6)>
        xr = 1 if name in ("gemm_o", "gemm_dn") and (M // 256) * (D // 256) <= 256 and M % 256 == 0 else 0
        kname = "gemm_bf16_nt_256f_kernel"               # the 256-tile persistent kernel (full-line operand stream)
        sym = ("gemm_qkv_attn256_kernel<%d>" % (2 if folded else 0)) if (name == "gemm_qkv" and fused_attn) else \
            ("%s<%d, %d, %d, 0, %d>" % (kname, epi_id[name], fold_id[name] if folded else 0, xr, 1 if (xr and name == "gemm_dn") else 0)) if name in epi_id else \
            {"attention": {0: "attn_fwd_kernel<%d, false>" % (D // cfg.score.num_heads), 1: "attn_fwd_resident_kernel<%d>" % (D // cfg.score.num_heads),
                           2: "attn_fwd_head_kernel<%d, %d>" % (D // cfg.score.num_heads, (T + 63) // 64)}[
                               int(_lib.lib().ldt_attention_route(B, cfg.score.num_heads, T, T, D // cfg.score.num_heads))],
             "ln_modulate": "ln_mod_vec_kernel<%d>" % (D // 256)}.get(name)
        if sym is None:
            continue
        if name in hbm_bytes:                           # HBM-bound kernels: algorithmic bytes / time against 8 TB/s
            r = {"bound": "hbm", "achieved": k["gbps"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(k["gbps"] / PEAK_HBM_GBS, 4),
                 "bytes_per_launch": hbm_bytes[name]}
        else:
            r = {"bound": "mfma", "achieved": k["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(k["tflops"] / PEAK_BF16_TFLOPS, 4), "flops_per_launch": flops[name]}
        tr_, prov = measured_traffic(sym)
        mu, mu_src = measured_mfma_util(sym)
        r.update(mfma_util=mu, mfma_util_source=mu_src)
        r.update(traffic=tr_, traffic_source=prov, kernel=sym, op=name, avg_launch_ms=k["avg_ms"], ms_per_forward=k["ms_per_forward"],
                 ln_folding=folded)
        if name == "gemm_qkv" and fused_attn:
            r["fused"] = "QKV projection + self-attention of the block in one launch: flops_per_launch = projection + attention"
        roofs[name] = r
    dom = max((n for n in roofs if n.startswith("gemm_")), key=lambda n: kernels[n]["ms_per_forward"])
    # mlp.out and MLP-up + GELU take 3.0-3.1 ms of a forward each and swap places from box to box: within 2 % of the maximum the line names
    # mlp.out (the class DESIGN.md and the committed profiles call dominant), so that successive bench lines price the same kernel
    if "gemm_dn" in roofs and dom != "gemm_dn" and kernels["gemm_dn"]["ms_per_forward"] >= 0.98 * kernels[dom]["ms_per_forward"]:
        roofs["gemm_dn"]["dominant_by_tie_break"] = "%s took %.4f ms per forward in this run, mlp.out %.4f (within 2 %%)" % (
            dom, kernels[dom]["ms_per_forward"], kernels["gemm_dn"]["ms_per_forward"])
        dom = "gemm_dn"
    return roofs[dom], roofs, kernels


def attention_standalone(trainer, cfg, B, reps=20):
    """The north-star's attention figure: the self-attention kernel of the bench shape (B x heads problems of T x T x 64 on the
    Score's own q | k | v row layout) timed STAND-ALONE — in the forward it is the epilogue of the QKV GEMM since round 4, so no
    launch of its own exists there.  Algorithmic bytes = read Q, K, V + write O in bf16 = 4 M D 2, HIP events on the
    launch stream, random data."""
    from ldt_amd import ops
    model = trainer.model
    T, D, H = cfg.score.z_scale, cfg.score.hidden_size, cfg.score.num_heads
    M = B * T
    qkv = torch.randn(M, 3 * D, device=trainer.device).to(torch.bfloat16)
    o = torch.empty(B, H, T, D // H, device=trainer.device, dtype=torch.bfloat16)
    run = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, T, T, D // H, out=o)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 4.0 * M * D * 2
    gbs = nbytes / (ms * 1e-3) / 1e9
    from ldt_amd._lib import lib as _ldt_lib
    route = int(_ldt_lib().ldt_attention_route(B, H, T, T, D // H))       # the launcher's own decision (incl. LDT_ATTN_FORCE), not a guess
    sym = {0: "attn_fwd_kernel<%d, false>" % (D // H), 1: "attn_fwd_resident_kernel<%d>" % (D // H),
           2: "attn_fwd_head_kernel<%d, %d>" % (D // H, (T + 63) // 64)}[route]
    tr_, prov = measured_traffic(sym)
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 5), "kernel": sym, "traffic": tr_, "traffic_source": prov,
            "tflops": round(4.0 * M * T * D / (ms * 1e-3) / 1e12, 1),
            "scope": "stand-alone launches on random q | k | v (%d back to back); in the forward this loop runs as the epilogue of the QKV GEMM" % reps}


def rel_mse(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-300))


def trained_tiny_parity():
    """End-to-end points / Chamfer at FIXED bars on the well-conditioned fixture tests/golden/trained_tiny.npz: a tiny Score
    trained by the REFERENCE's own Trainer.update and sampled by the reference's Trainer.sample (oracle/gen_trained_tiny_golden.py),
    so sampled latents stay at the data scale.  Same check as tests/test_gpu_path.py::test_trainer_sample_trained_weights_fixed_bars."""
    import json
    from types import SimpleNamespace
    import numpy as np
    import ldt_amd
    from oracle import ldt_oracle as O
    gdir = os.path.join(ROOT, "tests", "golden")
    to_ns = lambda d: SimpleNamespace(**{k: (to_ns(v) if isinstance(v, dict) else v) for k, v in d.items()})
    cfg = to_ns(json.load(open(os.path.join(gdir, "tiny_cfg.json"))))
    zf = np.load(os.path.join(gdir, "trained_tiny.npz"))
    a = {k: torch.from_numpy(np.asarray(zf[k])) for k in zf.files if "::" not in k}
    sd = {pre: {k.split("::", 1)[1]: torch.from_numpy(np.asarray(zf[k])) for k in zf.files if k.startswith(pre + "::")} for pre in ("w", "c")}
    score = ldt_amd.Score(cfg.score); score.load_state_dict(sd["w"], strict=True)
    comp = ldt_amd.Compressor(cfg.compressor); comp.load_state_dict(sd["c"], strict=True)
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    traj = []
    pts, eps = tr.sample(a["x0"].shape[0], x0=a["x0"], noise=a["noises"], trajectory=traj)
    cd = float((O.chamfer_cd(pts.cpu(), a["points"]) / (a["points"] ** 2).sum(-1).mean(1)).max())
    # the fixture's own conditioning: the fp32 oracle's trajectory with nothing but the Score's GEMM weights rounded to bf16 once
    # (tests/test_gpu_path.py::_oracle_bf16_weight_sensitivity)
    is_w = lambda k: k.endswith("weight") and "adaLN" not in k and any(t in k for t in ("fc_q", "fc_kv", "fc_o", "mlp.fc", "mlp.out", "ln_in", "ln_out.ln"))
    sdq = {k: (v.to(torch.bfloat16).float() if is_w(k) else v) for k, v in sd["w"].items()}
    nl = [a["noises"][i] for i in range(a["noises"].shape[0])]
    with torch.no_grad():
        rec, recq = [], []
        _, eps_o = O.trainer_sample(sd["w"], sd["c"], cfg, a["x0"], nl, record=rec)
        _, eps_q = O.trainer_sample(sdq, sd["c"], cfg, a["x0"], nl, record=recq)
    xs = traj[0].cpu()
    curve = [rel_mse(xs[i], rec[i][3]) for i in range(xs.shape[0])]
    curve_q = [rel_mse(recq[i][3], rec[i][3]) for i in range(len(rec))]
    sens = rel_mse(eps_q, eps_o)
    out = {"fixture": "tests/golden/trained_tiny.npz (reference-trained tiny Score, reference-sampled: latents rms %.2f)" % float(a["latent_rms"]),
           "final_latent": rel_mse(eps.cpu(), a["eps"]), "points_rel_mse": rel_mse(pts.cpu(), a["points"]), "chamfer_norm": cd,
           "per_step_max": max(curve), "per_step_last": curve[-1],
           "oracle_bf16_weight_sensitivity": {"what": "the fp32 oracle's own trajectory with the Score's GEMM weights rounded to bf16 once: "
                                                      "what any bf16-weight path inherits from this fixture's schedule (N = 50)",
                                              "final_latent": sens, "per_step_max": max(curve_q)},
           "final_latent_over_sensitivity": rel_mse(eps.cpu(), a["eps"]) / sens,
           "tol": {"final_latent": 1e-4, "final_latent_over_sensitivity": 4.0, "points_rel_mse": 1e-3, "chamfer_norm": 1e-3}}
    out["pass"] = bool(out["final_latent"] <= 1e-4 and out["final_latent"] <= 4 * sens and out["points_rel_mse"] <= 1e-3 and cd <= 1e-3)
    return out


def c1_well_conditioned_parity(trainer, cfg_full):
    """End to end at the PRODUCTION width with fixed bars: config C1's shape (B=4, T=256, N=100, decode to 2048 points) on the
    well-conditioned fixture of oracle/fixtures.py (seeded Score, ln_out.ln.weight += pinv(ln_in.weight): latents stay at rms 0.1-2
    through the reverse SDE instead of inflating to ~400), same injected noise on both sides.  Same check as
    tests/test_gpu_fullsize.py::test_c1_shape_well_conditioned_fixed_bars."""
    import copy
    import ldt_amd
    from oracle import ldt_oracle as O
    from oracle.fixtures import condition_score_head
    torch.set_num_threads(host_cores())
    B, N = 4, 100
    cfg = copy.deepcopy(cfg_full)
    cfg.sde.sample_N = N
    T, z = cfg.score.z_scale, cfg.score.z_dim
    sd_w = condition_score_head(trainer.model.state_dict())
    sd_c = {k: v.detach().float().cpu() for k, v in trainer.compressor.state_dict().items()}
    score = ldt_amd.Score(cfg.score)
    score.load_state_dict(sd_w, strict=True)
    tr = ldt_amd.Trainer(cfg, score, trainer.compressor, trainer.device)
    x0, noises = O.draw_noises(99, B, T, z, N)
    traj = []
    pts, eps = tr.sample(B, x0=x0, noise=torch.stack(noises), trajectory=traj)
    rec = []
    with torch.no_grad():
        t0 = time.time()
        ref_pts, ref_eps = O.trainer_sample(sd_w, sd_c, cfg, x0, noises, record=rec,
                                            progress=lambda i: log("  oracle (well-conditioned C1) step %d/%d" % (i + 1, N)) if (i + 1) % 20 == 0 else None)
        t_cpu = time.time() - t0
    xs = traj[0].cpu()
    per_step = [rel_mse(xs[i], rec[i][3]) for i in range(N)]
    cd = float((O.chamfer_cd(pts.cpu(), ref_pts) / (ref_pts ** 2).sum(-1).mean(1)).max())
    out = {"fixture": "oracle/fixtures.py::condition_score_head: the seeded production-width Score (hidden %d x %d blocks) with "
                      "ln_out.ln.weight += 1.0 * pinv(ln_in.weight); C1 shape B=%d, T=%d, N=%d, decode to %d points"
                      % (cfg.score.hidden_size, cfg.score.num_blocks, B, T, N, ref_pts.shape[1]),
           "latent_rms_steps_0_50_98_99": [round(float(rec[i][3].pow(2).mean().sqrt()), 3) for i in (0, 50, 98, 99)],
           "per_step_max": max(per_step), "per_step_last": per_step[-1], "final_latent": rel_mse(eps.cpu(), ref_eps),
           "points_rel_mse": rel_mse(pts.cpu(), ref_pts), "chamfer_norm": cd, "oracle_seconds": round(t_cpu, 1),
           "tol": {"per_step": 1e-4, "final_latent": 1e-4, "points_rel_mse": 1e-3, "chamfer_norm": 1e-3}}
    out["pass"] = bool(out["per_step_max"] <= 1e-4 and out["final_latent"] <= 1e-4 and out["points_rel_mse"] <= 1e-3 and cd <= 1e-3)
    del score, tr
    torch.cuda.empty_cache()
    return out


def c1_baseline_and_parity(trainer, cfg_full):
    """Config C1 EXACTLY (BASELINE.md §3 / SURVEY §8d): B=4, T=256, N=100 ancestral steps + decode.
    The CPU oracle's wall time is the cpu_baseline (x 1/10 for N=1000: the loop body is step-invariant); its trajectory
    is the parity reference for the SAME run on the GPU (same weights, same injected x0 / per-step noise)."""
    import copy
    import ldt_amd
    from oracle import ldt_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    B, N = 4, 100
    cfg = copy.deepcopy(cfg_full)
    cfg.sde.sample_N = N
    T, z = cfg.score.z_scale, cfg.score.z_dim
    sd_s = {k: v.detach().float().cpu() for k, v in trainer.model.state_dict().items()}
    sd_c = {k: v.detach().float().cpu() for k, v in trainer.compressor.state_dict().items()}
    x0, noises = O.draw_noises(1234, B, T, z, N)
    log("C1 on the CPU oracle: %d threads, B=%d, T=%d, N=%d" % (cores, B, T, N))
    rec = []
    with torch.no_grad():
        t0 = time.time()
        ref_pts, ref_eps = O.trainer_sample(sd_s, sd_c, cfg, x0, noises, record=rec,
                                            progress=lambda i: log("  oracle C1 step %d/%d" % (i + 1, N)) if (i + 1) % 20 == 0 else None)
        t_cpu = time.time() - t0
        g = torch.Generator().manual_seed(0)               # conditioning of the decode map at these (random-weight) latents
        pert = O.compressor_decode(sd_c, cfg.compressor, ref_eps * (1 + 2 ** -9 * torch.randn(ref_eps.shape, generator=g)))
    log("C1 oracle done: %.1f s" % t_cpu)
    tr100 = ldt_amd.Trainer(cfg, trainer.model, trainer.compressor, trainer.device)
    traj = []
    pts, eps = tr100.sample(B, x0=x0, noise=torch.stack(noises), trajectory=traj)
    torch.cuda.synchronize()
    t0 = time.time()
    tr100.sample(B, x0=x0, noise=torch.stack(noises))
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    xs = traj[0].cpu()                                      # [N, B, T, z]: x after every step
    per_step = [rel_mse(xs[i], rec[i][3]) for i in range(N)]
    r2 = (ref_pts ** 2).sum(-1).mean(1)
    cd = float((O.chamfer_cd(pts.cpu(), ref_pts) / r2).max())
    floor_pts = rel_mse(pert, ref_pts)
    floor_cd = float((O.chamfer_cd(pert, ref_pts) / r2).max())
    # What the line's top-level parity figures are: the FIXED-BAR end-to-end run at the production width
    # (`end_to_end_full_width`: C1's shape on the well-conditioned fixture) — per-step / final latents, decoded points, Chamfer, pass.
    # The C1 run with the plain seeded (random) weights — the very run the cpu_baseline times — keeps its LATENT curve at top level
    # (`c1_latents`, bar 1e-4); its decoded-cloud figures depend on the oracle's own conditioning at latents of rms ~400 and are
    # informational only (`ill_conditioned_random_weights`, no pass).
    c1_lat = {"config": "C1: B=4, T=%d, N=%d ancestral, injected x0 + per-step noise (seed 1234): the run cpu_baseline times" % (T, N),
              "per_step_max": max(per_step), "per_step_last": per_step[-1], "final_latent": rel_mse(eps.cpu(), ref_eps),
              "tol": 1e-4, "gpu_seconds": round(t_gpu, 3)}
    c1_lat["pass"] = bool(c1_lat["per_step_max"] <= 1e-4 and c1_lat["final_latent"] <= 1e-4)
    fw = c1_well_conditioned_parity(trainer, cfg_full)
    tt = trained_tiny_parity()
    parity = {"what": "top-level figures = end_to_end_full_width (fixed bars, production width, C1's shape, well-conditioned fixture); "
                      "metric: relative MSE |a-b|^2/|b|^2; Chamfer / mean squared radius",
              "per_step_max": fw["per_step_max"], "final_latent": fw["final_latent"], "points_rel_mse": fw["points_rel_mse"],
              "chamfer_norm": fw["chamfer_norm"], "tol": fw["tol"],
              "c1_latents": c1_lat,
              "end_to_end_full_width": fw, "end_to_end_trained": tt,
              "ill_conditioned_random_weights": {
                  "what": "decoded cloud of the C1 run with random weights (latents inflated to rms ~400-600, decoder softmaxes saturated): "
                          "informational, not a pass criterion; `decode_conditioning` = the fp32 oracle's own decode under one bf16 rounding "
                          "(2^-9 relative) of its latents",
                  "points_rel_mse": rel_mse(pts.cpu(), ref_pts), "chamfer_norm": cd,
                  "decode_conditioning": {"points_rel_mse": floor_pts, "chamfer_norm": floor_cd}}}
    parity["pass"] = bool(fw["pass"] and c1_lat["pass"] and tt["pass"])
    base = {"value": (B / t_cpu) / 10.0, "unit": "shapes/sec", "cores": cores, "kind": "port",
            "sample": "config C1 exactly: oracle (PyTorch-CPU fp32 restatement of the reference, pinned to reference-captured goldens), "
                      "%d threads, B=%d shapes, T=%d tokens, N=%d ancestral steps + decode = %.1f s wall; x 1/10 for N=1000 "
                      "(per-step cost is step-invariant)" % (cores, B, T, N, t_cpu),
            "c1_seconds": round(t_cpu, 2), "c1_shapes_per_sec_at_N100": B / t_cpu}
    return base, parity


# ----------------------------------------------------------------------------------------------------------- extras
def _timed(fn, reps=2, stats=None):
    """seconds per call (median over `reps` individually synchronised calls after one warm-up) and the last result;
    `stats` (a dict) receives min / max / reps."""
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    if stats is not None:
        stats.update(reps=reps, min_s=ts[0], max_s=ts[-1], median_s=med)
    return med, r


def extra_c4(tokens=256, batch=1024, chunk=1024):
    """BASELINE configs[3]: Compressor encode + decode only (2048 -> T tokens -> 2048), batch 1024, 1 GPU, and the
    cross-attention kernel on its own in both orientations.  CPU baseline: the oracle on a bounded sample of clouds."""
    import ldt_amd
    from ldt_amd import ops
    from oracle import ldt_oracle as O
    cfg = ldt_amd.airplane_config(latent_tokens=tokens)
    torch.manual_seed(0)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    sd_c = {k: v.detach().float().clone() for k, v in comp.state_dict().items()}
    comp = comp.cuda()
    g = torch.Generator().manual_seed(2)
    pts_h = torch.randn(batch, 2048, 3, generator=g)
    pts_h = pts_h - pts_h.mean(1, keepdim=True)
    pts_h = pts_h / pts_h.norm(dim=-1).amax(1)[:, None, None]
    pts = pts_h.cuda()
    st_enc, st_dec = {}, {}
    t_enc, eps = _timed(lambda: torch.cat([comp(pts[i:i + chunk])["all_eps"] for i in range(0, batch, chunk)]), reps=12, stats=st_enc)
    t_dec, dec = _timed(lambda: comp.sample((batch, 2048), given_eps=eps), reps=12, stats=st_dec)
    assert bool(torch.isfinite(dec).all())
    cc = cfg.compressor
    d, L, T = cc.hidden_dim, cc.n_layers, tokens
    # SURVEY §8d decode FLOPs per shape: L x (q/o + MLP on 2048 rows, cross-attention 2048 x T, ln + kv on T rows) x 2
    dec_flops = L * (2048 * (2 * d * d + 2 * d * 4 * d) + 2 * 2048 * T * d + T * (cc.z_dim * d + d * 2 * d)) * 2.0
    dec_tf = dec_flops * batch / t_dec / 1e12
    # unfused lower bound of the decode (SURVEY §8d "unfused: 2 MB x passes per shape and layer"): the fp32 set O
    # (2048 x d) read + written by each of the block's 7 kernels -> HBM-bound time at 8 TB/s
    unfused_bytes = L * 7 * 2 * 2048 * d * 4.0
    unfused_clouds_s = PEAK_HBM_GBS * 1e9 / unfused_bytes
    # The floors of the FUSED decode path itself, per cloud (a fused path is not graded against an unfused bound).  Per level
    # two launches touch the (2048 x d) set: attention + fc_o + residual (q bf16 in, x fp32 in + out) and LN + MLP + residual + the next
    # level's q (x fp32 in + out, q bf16 out) = 20 B per row and channel; the matrix pipe: dec_flops at the dense bf16 peak; VALU issue on 1024
    # SIMDs at 2.4 GHz: 16 cycles per 64 attention scores (hb() below) + 1048 cycles per 2048 hidden activations of the MLP (the kernel's
    # measured mix: 230 VALU at 4 cycles + 32 v_exp at 8 per wave and 32 x 64 chunk, profiles/r05_fused_mlp_analysis.txt).
    fused_fl = {"hbm_us": L * 2048 * d * 20.0 / (PEAK_HBM_GBS * 1e9) * 1e6, "mfma_us": dec_flops / (PEAK_BF16_TFLOPS * 1e12) * 1e6,
                "valu_issue_us": L * (cc.num_heads * 2048.0 * T / 64.0 * 16.0 + 2048.0 * 4 * d / 2048.0 * 1048.0) / 1024.0 / 2.4e9 * 1e6}
    # cross-attention kernel alone (d = 128, 4 heads x 32), 128 clouds per launch
    Bm, H, dh = 128, cc.num_heads, d // cc.num_heads

    def attn_time(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    q = torch.randn(Bm * 2048, d, device="cuda").to(torch.bfloat16)
    kv = torch.randn(Bm * T, 2 * d, device="cuda").to(torch.bfloat16)
    o = torch.empty(Bm, H, 2048, dh, device="cuda", dtype=torch.bfloat16)
    t1 = attn_time(lambda: ops.attention_fwd(q, kv[:, :d], kv[:, d:], Bm, H, 2048, T, dh, out=o))
    b1 = (2 * Bm * 2048 * d + 2 * Bm * T * d) * 2.0
    q2 = torch.randn(Bm * T, d, device="cuda").to(torch.bfloat16)
    kv2 = torch.randn(Bm * 2048, 2 * d, device="cuda").to(torch.bfloat16)
    o2 = torch.empty(Bm, H, T, dh, device="cuda", dtype=torch.bfloat16)
    t2 = attn_time(lambda: ops.attention_fwd(q2, kv2[:, :d], kv2[:, d:], Bm, H, T, 2048, dh, out=o2))
    # CPU oracle on a bounded sample
    nb = 8
    torch.set_num_threads(host_cores())
    with torch.no_grad():
        noise = [torch.randn(nb, T, cc.z_dim) for _ in range(L)]
        t0 = time.time(); r = O.compressor_encode(sd_c, cc, pts_h[:nb], noise); t_ce = time.time() - t0
        t0 = time.time(); dec_ref = O.compressor_decode(sd_c, cc, r["all_eps"]); t_cd = time.time() - t0
        # parity at full size, through the big-batch kernels: the whole batch is encoded / decoded again with the oracle's clouds, posterior
        # noise and latents in its first `nb` slots, and those slots are compared (relative MSE; FPS indices exactly)
        rel = lambda a, b: float(((a.double() - b.double()) ** 2).sum() / (b.double() ** 2).sum())
        gn = torch.Generator().manual_seed(5)
        full_noise = [torch.cat([n, torch.randn(batch - nb, T, cc.z_dim, generator=gn)]).cuda() for n in noise]
        og = comp(pts, post_noise=full_noise)
        eps_in = eps.clone(); eps_in[:nb] = r["all_eps"].cuda()
        dg = comp.sample((batch, 2048), given_eps=eps_in)
        par = {"clouds": nb, "of_batch": batch, "fps_idx_equal": bool(torch.equal(og["fps_idx"][:nb].cpu().long(), r["fps_idx"].long())),
               "encode_all_eps_rel_mse": rel(og["all_eps"][:nb].cpu(), r["all_eps"]), "encode_set_rel_mse": rel(og["set"][:nb].cpu(), r["set"]),
               "decode_rel_mse": rel(dg[:nb].cpu(), dec_ref), "tol": {"all_eps": 1e-4, "set": 1e-4, "decode": 1e-4}}
        par["pass"] = bool(par["fps_idx_equal"] and par["encode_all_eps_rel_mse"] < 1e-4 and par["encode_set_rel_mse"] < 1e-4 and par["decode_rel_mse"] < 1e-4)

    def hb(bytes_, t, nq, nk):
        """The kernel against the THREE floors of its launch: HBM (algorithmic bytes at 8 TB/s), the matrix pipe
        (QK^T + PV flops at the dense bf16 peak) and VALU issue — per score one v_exp_f32 (8 issue cycles per wave-instruction,
        MI355X_MICROARCH.md 'vector-instruction ISSUE cost') + half each of v_pk_fma (scale, - max), v_pk_add (row sum), v_cvt_pk_bf16
        and v_max3 (4 cycles each): 16 cycles per 64 scores on one of the chip's 1024 SIMDs at the 2.4 GHz peak clock.  The largest
        floor names the bound; `frac` = that floor / measured (the phases do not overlap perfectly: the sum of the three is the
        no-overlap time)."""
        gbs = bytes_ / t / 1e9
        scores = float(Bm) * H * nq * nk
        fl = {"hbm_us": bytes_ / (PEAK_HBM_GBS * 1e9) * 1e6, "mfma_us": 4.0 * scores * dh / (PEAK_BF16_TFLOPS * 1e12) * 1e6,
              "valu_issue_us": scores / 64.0 * 16.0 / 1024.0 / 2.4e9 * 1e6}
        bound = max(fl, key=fl.get)
        # Round 6 (profiles/r06_attention_dh32.txt): the kernel's actual per-score instruction stream at Dh = 32 — per joint 64-key step and wave
        # 32 v_exp (8 issue cycles) + 16 v_pk_fma + 16 v_pk_add + 15 v_max3 + 8 v_max (4 each) + 16 v_cvt_pk_bf16 (4.5) + 8 MFMA issue slots (8)
        # + ~50 cycles of LDS reads / moves = 660 cycles per 2048 scores (the 512 of `valu_issue_us` leave out conversions, maxima and MFMA
        # slots).  Timing-only builds show the kernel compute-bound on exactly this stream (72 of its 81 us with the K / V traffic removed).
        mix_us = scores / 2048.0 * 660.0 / 1024.0 / 2.4e9 * 1e6
        return {"us": round(t * 1e6, 1), "bound": bound[:-3], "floors_us": {k: round(v, 1) for k, v in fl.items()},
                "frac": round(fl[bound] / (t * 1e6), 4), "no_overlap_sum_us": round(sum(fl.values()), 1),
                "instruction_mix_issue_us": round(mix_us, 1), "frac_of_instruction_mix_issue": round(mix_us / (t * 1e6), 4),
                "hbm": {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "bytes_per_launch": bytes_},
                "scores_per_launch": scores, "clouds_per_launch": Bm}

    return {"workload": "BASELINE configs[3]: Compressor encode+decode only, batch %d, 2048 pts <-> %d tokens, 1 GPU" % (batch, T),
            "encode_clouds_per_s": round(batch / t_enc, 1), "decode_clouds_per_s": round(batch / t_dec, 1),
            "encode_decode_clouds_per_s": round(batch / (t_enc + t_dec), 1),
            "timing": {"what": "median of 12 individually synchronised calls after one warm-up call; clouds/s at the slowest / fastest call",
                       "encode_clouds_per_s_min_max": [round(batch / st_enc["max_s"], 1), round(batch / st_enc["min_s"], 1)],
                       "decode_clouds_per_s_min_max": [round(batch / st_dec["max_s"], 1), round(batch / st_dec["min_s"], 1)]},
            "decode_roofline": {"bound": "mfma", "achieved": round(dec_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(dec_tf / PEAK_BF16_TFLOPS, 4), "flops_per_cloud": dec_flops},
            "decode_fused_floors": {"what": "floors of the fused decode path per cloud (HBM: 20 B per row, channel and level at 8 TB/s; matrix pipe; VALU issue)",
                                    "floors_us_per_cloud": {k: round(v, 2) for k, v in fused_fl.items()}, "bound": max(fused_fl, key=fused_fl.get)[:-3],
                                    "measured_us_per_cloud": round(t_dec / batch * 1e6, 2),
                                    "frac": round(max(fused_fl.values()) / (t_dec / batch * 1e6), 4),
                                    "no_overlap_sum_us": round(sum(fused_fl.values()), 2)},
            "decode_unfused_reference": {"what": "NOT a bound of this path: what an unfused chain (7 kernels per block each reading + writing the fp32 "
                                                 "(2048 x %d) set) could reach at 8 TB/s" % d,
                                         "clouds_per_s": round(unfused_clouds_s, 1), "bytes_per_cloud": unfused_bytes,
                                         "measured_over_it": round(batch / t_dec / unfused_clouds_s, 3)},
            "cross_attn_q2048_kvT": hb(b1, t1, 2048, T), "cross_attn_qT_kv2048": hb((2 * Bm * T * d + 2 * Bm * 2048 * d) * 2.0, t2, T, 2048),
            "parity": par,
            "cpu_baseline": {"encode_clouds_per_s": round(nb / t_ce, 3), "decode_clouds_per_s": round(nb / t_cd, 3), "cores": host_cores(),
                             "kind": "port", "sample": "oracle compressor_encode / compressor_decode on %d of the %d clouds" % (nb, batch)}}


def vipc_parity(score, tokens, batch, cond, sd_s, n_steps=25, nb=8):
    """BASELINE configs[4]'s per-GPU share at the production width vs the oracle: a teacher-forced
    Score.forward(condition=) on the whole batch and `n_steps` free-running steps of the fused conditional loop on injected
    noise (first `nb` samples compared: trajectories are independent).  Same check as
    tests/test_gpu_fullsize.py::test_fullsize_vipc_conditioned_vs_oracle."""
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = ldt_amd.airplane_config(latent_tokens=tokens, sample_N=n_steps)
    z = cfg.score.z_dim
    g = torch.Generator().manual_seed(9)
    x = torch.randn(batch, tokens, z, generator=g)
    t = torch.rand(batch, generator=g) * 0.98 + 0.01
    pts_tm, img = cond[0].cpu().transpose(1, 2).contiguous(), cond[1].cpu()
    out = score(x.cuda(), t.cuda(), condition=cond)
    with torch.no_grad():
        ref = O.score_forward(sd_s, cfg.score, x, t, condition=(pts_tm, img))
    torch.manual_seed(4)
    comp = ldt_amd.Compressor(cfg.compressor); comp.init()
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    x0, noises = O.draw_noises(77, batch, tokens, z, n_steps)
    traj = []
    _, eps = tr.sample(batch, condition=cond, x0=x0, noise=torch.stack(noises), trajectory=traj)
    sde = O.VPSDE(cfg.sde)
    fn = O.score_fn_from_model(sde, lambda xx, tt: O.score_forward(sd_s, cfg.score, xx, tt, condition=(pts_tm[:nb], img[:nb])))
    rec = []
    with torch.no_grad():
        ref_eps = O.sample_discrete(sde, fn, x0[:nb], [n[:nb] for n in noises], n_steps, record=rec)
    xs = traj[0][:, :nb].cpu()
    curve = [rel_mse(xs[i], rec[i][3]) for i in range(n_steps)]
    par = {"what": "B=%d, T=%d, S=%d condition tokens, hidden %d x %d blocks: teacher-forced forward (all samples) and %d free-running "
                   "steps of the fused conditional loop (first %d samples) vs the CPU oracle" % (batch, tokens, cond[0].shape[2], cfg.score.hidden_size,
                                                                                             cfg.score.num_blocks, n_steps, nb),
           "teacher_forced": rel_mse(out.cpu(), ref), "per_step_max": max(curve), "final_latent": rel_mse(eps[:nb].cpu(), ref_eps),
           "tol": 1e-4}
    par["pass"] = bool(max(par["teacher_forced"], par["per_step_max"], par["final_latent"]) <= 1e-4)
    return par


def extra_sampling(score, tokens, batch, n_steps, vipc, cpu_steps=6):
    """A sampling workload at `tokens` latent tokens: shapes/s of Trainer.sample(batch) with n_steps SDE steps (+ decode), its
    whole-job MFMA fraction, and a bounded CPU-oracle baseline (B=4, first `cpu_steps` steps, extrapolated linearly)."""
    import ldt_amd
    from oracle import ldt_oracle as O
    cfg = ldt_amd.airplane_config(latent_tokens=tokens, sample_N=n_steps)
    torch.manual_seed(1)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    tr = ldt_amd.Trainer(cfg, score, comp, "cuda:0")
    cond, S = None, 32
    if vipc:                                              # synthetic ConditionNet outputs
        g = torch.Generator().manual_seed(5)
        cond = (torch.randn(batch, cfg.score.hidden_size, S, generator=g).cuda(), torch.randn(batch, cfg.score.t_dim, generator=g).cuda())
    dt, (pts, eps) = _timed(lambda: tr.sample(batch, condition=cond), reps=1)
    assert bool(torch.isfinite(eps).all())
    flops = score_flops_per_sample_step(cfg) * n_steps * batch
    if vipc:                                              # + per-sample AdaLN rows (6 D t_dim per block + final) per sample-step
        D, L = cfg.score.hidden_size, cfg.score.num_blocks
        flops += 2.0 * (L * 6 * D + 2 * D) * cfg.score.t_dim * n_steps * batch
    tf = flops / dt / 1e12
    base, sd_s = bounded_cpu_baseline(score, cfg, tokens, n_steps, cond, cpu_steps)
    par = vipc_parity(score, tokens, batch, cond, sd_s) if vipc else None
    return {**({"parity": par} if par else {}),
            "shapes_per_s": round(batch / dt, 3), "ms_per_sde_step": round(1e3 * dt / n_steps, 3), "seconds_per_call": round(dt, 3),
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                         "scope": "whole job (Score flops of SURVEY §8d / wall time)"},
            "cpu_baseline": base}


def extra_ln_kernels(trainer, B, n_steps):
    """The headline workload on the LayerNorm-kernel path: the same `Trainer.sample(B)` call with LN folding off — what a checkpoint whose
    LayerNorm inputs trip the fold guard (mean^2 / variance > 16 on some row, Score.FOLD_MAX_MEAN_RATIO) falls back to.  One timed call
    after one warm-up call; the model's folding state is restored afterwards."""
    model = trainer.model
    seen = model.collect_fold_ratio()                     # the in-loop monitor's running maximum over the timed (folded) calls
    was = model._fold_disabled
    model._fold_disabled = True
    try:
        assert not model.can_fold(B, model.z_scale if hasattr(model, "z_scale") else 256), "folding still on"
        dt, _ = _timed(lambda: trainer.sample(B), reps=1)
    finally:
        model._fold_disabled = was
    return {"workload": "BASELINE configs[1] with the LayerNorm kernels (LN folding off: the fold guard's fallback path)",
            "shapes_per_s": round(B / dt, 4), "ms_per_sde_step": round(1e3 * dt / n_steps, 3), "seconds_per_call": round(dt, 3),
            "fold_ratio_seen_on_the_folded_path": seen, "fold_guard_bound": model.FOLD_MAX_MEAN_RATIO, "fold_guard_tripped": bool(was)}


def bounded_cpu_baseline(score, cfg, tokens, n_steps, cond, cpu_steps=6, Bc=4):
    """CPU oracle on a bounded sample of a sampling workload: B = 4 shapes, the first `cpu_steps` SDE steps (the loop body is
    step-invariant), extrapolated linearly to `n_steps`; decode excluded.  `cond` = the GPU run's (pts, img) condition or None."""
    from oracle import ldt_oracle as O
    torch.set_num_threads(host_cores())
    sd_s = {k: v.detach().float().cpu() for k, v in score.state_dict().items()}
    x0, noises = O.draw_noises(7, Bc, tokens, cfg.score.z_dim, cpu_steps)
    sde = O.VPSDE(cfg.sde)
    cnd = None if cond is None else (cond[0][:Bc].cpu().transpose(1, 2).contiguous(), cond[1][:Bc].cpu())
    fn = O.score_fn_from_model(sde, lambda x, t: O.score_forward(sd_s, cfg.score, x, t, condition=cnd))
    with torch.no_grad():
        t0 = time.time()
        O.sample_discrete(sde, fn, x0, noises, n_steps, max_steps=cpu_steps)
        t_step = (time.time() - t0) / cpu_steps
    return {"value": Bc / (n_steps * t_step), "unit": "shapes/sec", "cores": host_cores(), "kind": "port",
            "sample": "oracle, B=%d, T=%d, first %d of %d steps (%.3f s/step), extrapolated linearly, decode excluded"
                      % (Bc, tokens, cpu_steps, n_steps, t_step)}, sd_s


def main():
    args = parse()
    if needs_self_launch(args, os.environ):              # before ANY GPU call in this process
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ                      # under torch.distributed.run: the collective path runs even at world 1
    # pre-flight, before any model is built (fail in seconds, not after the warm-up): one visible device per local rank, the launcher's
    # world size is the one asked for, every rank of this node got a device of its own (device_count() does not initialise the GPU)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world if launched else 1)))
    ndev = torch.cuda.device_count()
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)
    assert ndev >= local_world, "%d rank(s) on this node but only %d visible GPU(s)" % (local_world, ndev)
    assert 0 <= local_rank < local_world and local_rank < ndev, "LOCAL_RANK=%d outside [0, %d) / %d device(s)" % (local_rank, local_world, ndev)
    torch.cuda.set_device(local_rank)
    ranks_seen = 1
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")       # RCCL over xGMI; MASTER_* / RANK from the launcher
        assert dist.get_world_size() == args.gpus, "process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)
        # one int per rank through RCCL: (rank, local device) pairs — proves N ranks met, on N distinct devices of this node
        me = torch.tensor([rank, local_rank], device="cuda:%d" % local_rank, dtype=torch.int32)
        seen = torch.empty((world, 2), device=me.device, dtype=torch.int32)
        dist.all_gather_into_tensor(seen, me)
        seen = seen.cpu().tolist()
        assert sorted(r for r, _ in seen) == list(range(world)), "ranks met over RCCL: %s" % seen
        assert len({d for _, d in seen}) == min(world, local_world) or world > local_world, "two ranks share a device: %s" % seen
        ranks_seen = len(seen)
    device = "cuda:%d" % local_rank

    import ldt_amd
    cfg = ldt_amd.airplane_config(latent_tokens=args.tokens, sample_N=args.sde_steps)
    torch.manual_seed(0)                                 # same weights + same CPU generator stream on every rank
    score = ldt_amd.Score(cfg.score)
    comp = ldt_amd.Compressor(cfg.compressor)
    comp.init()
    trainer = ldt_amd.Trainer(cfg, score, comp, device)
    B = args.batch_per_gpu * world
    cond, S = None, 32
    if args.config == "c5":
        # BASELINE configs[4]: synthetic ConditionNet outputs for the GLOBAL batch (pts_condition ~ N(0,1) (B, hidden, S),
        # img_condition ~ N(0,1) (B, t_dim)), drawn identically on every rank; Trainer.sample hands each rank its rows
        g = torch.Generator().manual_seed(5)
        cond = (torch.randn(B, cfg.score.hidden_size, S, generator=g).to(device), torch.randn(B, cfg.score.t_dim, generator=g).to(device))

    def barrier():
        if launched:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    log("model built on %s (world %d), B=%d T=%d N=%d" % (device, world, B, args.tokens, args.sde_steps))
    for i in range(args.warmup):
        trainer.sample(B, condition=cond)
        torch.cuda.synchronize()
        log("warmup %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pts, eps = trainer.sample(B, condition=cond)
    barrier()
    dt = time.perf_counter() - t0
    log("timed region: %.2f s for %d sample() calls" % (dt, args.steps))
    if launched:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert pts.shape == (B, cfg.data.tr_max_sample_points, 3) and bool(torch.isfinite(eps).all())

    if rank == 0:
        value = B * args.steps / dt
        flops_call = score_flops_per_sample_step(cfg) * args.sde_steps * B
        if cond is not None:                             # + per-sample AdaLN rows (6 D t_dim per block + final) per sample-step
            flops_call += 2.0 * (cfg.score.num_blocks * 6 * cfg.score.hidden_size + 2 * cfg.score.hidden_size) * cfg.score.t_dim * args.sde_steps * B
        wl = ("BASELINE configs[1]: ShapeNet-airplane sampling, batch %d/GPU, %d latent tokens x %d, %d ancestral SDE steps (24-block d=1024 Score) "
              "+ decode to %d points" if cond is None else
              "BASELINE configs[4]: ViPC-conditioned sampling (synthetic ConditionNet outputs: 32 condition tokens + image vector per shape), "
              "batch %d/GPU, %d latent tokens x %d, %d ancestral SDE steps (24-block d=1024 Score, per-sample AdaLN, cross-attention on even "
              "blocks) + decode to %d points") % (args.batch_per_gpu, args.tokens, cfg.score.z_dim, args.sde_steps, cfg.data.tr_max_sample_points)
        line = {
            "metric": "shapes/sec (2048-pt, %d-step SDE sample)" % args.sde_steps, "value": round(value, 4),
            "unit": "shapes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl, "name": args.config,
                       "global_batch": B, "batch_per_gpu": args.batch_per_gpu, "latent_tokens": args.tokens,
                       "sde_steps": args.sde_steps, "points": cfg.data.tr_max_sample_points,
                       "parallelism": "dp%d batch slices, one all-gather" % world,
                       "collective": ("%s, world %d" % (dist.get_backend(), dist.get_world_size())) if launched else "none (single process)",
                       "rccl_ranks_seen": ranks_seen if launched else 0},
            "rccl_ranks_seen": ranks_seen if launched else 0,
            "achieved_tflops_whole_job": round(flops_call * args.steps / dt / 1e12, 1),
            "whole_job_mfma_frac": round(flops_call * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        if not args.no_roofline and cond is None:
            roof, roofs, kernels = roofline_pass(trainer, cfg, args.batch_per_gpu)
            log("roofline pass done: %s" % json.dumps(kernels))
            line["roofline"] = dict(roof)                # (a copy: the flat `x_` scalars below go into this object only)
            line["roofline_attention"] = roofs.get("attention") or dict(attention_standalone(trainer, cfg, args.batch_per_gpu),
                                                                        fused_into="gemm_qkv: %s" % roofs["gemm_qkv"]["kernel"])
            line["roofline_kernels"] = roofs
            line["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline and cond is None:
            base, parity = c1_baseline_and_parity(trainer, cfg)
            line["cpu_baseline"] = base
            line["parity"] = parity
            line["speedup_vs_cpu"] = round(value / base["value"], 1)
            log("parity: %s" % json.dumps({k: parity[k] for k in ("per_step_max", "final_latent", "chamfer_norm", "pass")}))
        if cond is not None:                             # configs[4]: whole-job roofline; bounded oracle baseline + full-width parity at N = 1
            tf = flops_call * args.steps / dt / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": None,
                                "scope": "whole job (Score flops of SURVEY §8d + per-sample AdaLN rows / wall time); per-kernel rooflines: --config c2"}
            if world == 1 and not args.no_cpu_baseline:
                base, sd_s = bounded_cpu_baseline(score, cfg, args.tokens, args.sde_steps, cond)
                line["cpu_baseline"] = base
                line["parity"] = vipc_parity(score, args.tokens, B, cond, sd_s)
                line["speedup_vs_cpu"] = round(value / base["value"], 1)
        if world == 1 and not args.no_extras and cond is None:
            extra = {}
            for name, fn in (("c2_layernorm_kernels", lambda: extra_ln_kernels(trainer, B, args.sde_steps)),
                             ("c4_compressor_b1024", lambda: extra_c4()),
                             ("c5_vipc_share_b32_t32", lambda: dict(workload="BASELINE configs[4] per-GPU share: ViPC-conditioned sampling, 32 shapes/GPU, "
                                                                    "32 latent tokens, 32 condition tokens, 1000 steps", **extra_sampling(score, 32, 32, 1000, True))),
                             ("shipped_t32_b64", lambda: dict(workload="shipped airplane YAML (32 latent tokens), batch 64, 1000 steps",
                                                              **extra_sampling(score, 32, 64, 1000, False)))):
                if time.time() - _T0 > args.budget_s:
                    extra[name] = {"skipped": "run older than --budget-s %.0f s" % args.budget_s}
                    continue
                try:
                    extra[name] = fn()
                    log("extra %s: %s" % (name, json.dumps(extra[name])[:400]))
                except Exception as e:                    # noqa: BLE001 — an extra block must not cost the headline line
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["extra"] = extra
        # ---- what a reader of the driver's projection of this line needs, as scalars.  The projection keeps the contract
        # keys (`roofline`, `cpu_baseline`, `config` with their scalar members) and only the NAMES of the others, so every figure is put at
        # top level AND inside `roofline` (prefix `x_`).
        flat = {}
        ra = line.get("roofline_attention") or {}
        for k_ in ("achieved", "frac", "avg_launch_ms", "kernel"):
            if k_ in ra:
                flat["attention_%s" % ("gbps" if k_ == "achieved" else "hbm_frac" if k_ == "frac" else k_)] = ra[k_]
        ex = line.get("extra") or {}
        c4 = ex.get("c4_compressor_b1024") or {}
        for k_ in ("encode_clouds_per_s", "decode_clouds_per_s"):
            if k_ in c4:
                flat["c4_" + k_] = c4[k_]
        for nm, sub in (("q2048_kvT", c4.get("cross_attn_q2048_kvT")), ("qT_kv2048", c4.get("cross_attn_qT_kv2048"))):
            if sub:
                flat["c4_cross_attn_%s_us" % nm] = sub["us"]
                flat["c4_cross_attn_%s_frac_of_bound" % nm] = sub["frac"]
        for nm, key in (("c5_share", "c5_vipc_share_b32_t32"), ("t32", "shipped_t32_b64"), ("c2_layernorm_kernels", "c2_layernorm_kernels")):
            if "shapes_per_s" in (ex.get(key) or {}):
                flat["%s_shapes_per_s" % nm] = ex[key]["shapes_per_s"]
        par = line.get("parity") or {}
        for k_ in ("per_step_max", "final_latent", "points_rel_mse", "chamfer_norm", "pass"):
            if k_ in par:
                flat["parity_" + k_] = par[k_]
        if "c1_latents" in par:
            flat["parity_c1_latents_per_step_max"] = par["c1_latents"]["per_step_max"]
        line.update(flat)
        if "roofline" in line:
            line["roofline"].update({"x_" + k_: v_ for k_, v_ in flat.items()})
        log("summary: %s" % json.dumps(flat))
        print(json.dumps(line))
    if launched:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
