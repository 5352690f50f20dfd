"""Multi-GPU sharding of the sampling loop: one process per GPU, contiguous batch slices by GLOBAL
sample index, no data-path collective inside the loop, ONE all-gather (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests) of the finished shapes at the end — SURVEY.md §8e.  The reference itself is
single-GPU (README.md:53); this module is new."""
import torch
import torch.distributed as dist


def initialized():
    return dist.is_available() and dist.is_initialized()


def world():
    if initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(num_samples, rank, world_size):
    """Contiguous slice [lo, hi) of rank `rank`; the batch is padded to a multiple of world_size so every
    rank runs the same shapes (padding rows are dropped after the gather)."""
    per = (num_samples + world_size - 1) // world_size
    lo = rank * per
    return lo, lo + per, per


def all_gather_rows(local, num_samples):
    """local [per, ...] on every rank -> [num_samples, ...] (same on every rank): the single collective."""
    rank, ws = world()
    if not initialized():
        return local[:num_samples]
    local = local.contiguous()                  # (a launched world of 1 still goes through the collective: same code path as N > 1)
    out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out[:num_samples]


def check_same_draws(x0, seed, device):
    """Every rank draws the full-batch x0 and the Philox key from its own CPU generator and keeps its rows, so the result
    is independent of the world size only if the generators were seeded identically (the reference's common_init,
    tools/utils.py:269-276).  One 16-byte all-gather of (key, bit pattern of sum(x0)) — control plane, not the data path —
    turns a silent mismatch into an error."""
    rank, ws = world()
    if not initialized():
        return
    dev = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([0 if seed is None else int(seed), int(x0.double().sum().view(torch.int64).item())], dtype=torch.int64, device=dev)
    allv = torch.empty((ws * 2,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, mine)
    allv = allv.view(ws, 2)
    if not bool((allv == allv[0]).all()):
        raise RuntimeError("Trainer.sample: ranks drew different x0 / noise keys (rank %d: %s; rank 0: %s) — seed the CPU generator "
                           "identically on every rank (torch.manual_seed) or pass x0= / seed=" % (rank, mine.tolist(), allv[0].tolist()))


def all_reduce_sum_(t):
    """In-place sum over ranks of a small device tensor (LangevinCorrector's batch-mean norms — the only cross-sample
    quantity on the sampling path, diffusion_continuous.py:204-205).  gloo (CPU tests) reduces a host copy."""
    rank, ws = world()
    if ws == 1:
        return t
    if dist.get_backend() == "nccl":
        dist.all_reduce(t)
    else:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    return t
