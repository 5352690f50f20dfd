"""Generation-quality metrics of the reference's validation loop (evaluation/evaluation_metrics.py), MI355X path.

Same function names / argument order / result keys as the reference module, so `from ldt_amd.metrics import
compute_all_metrics` replaces `from evaluation import compute_all_metrics` (trainer/Latent_SDE_Trainer.py:12,217):

    distChamfer(a, b)                  :23-33     -> (dl, dr)                     ldt_chamfer
    emd_approx(sample, ref)            :40-46     -> match_cost / n               ldt_emd_approx  (approxmatch.cu)
    EMD_CD(...)                        :69-109    -> {'mmd-CD', 'mmd-EMD'}
    _pairwise_EMD_CD_ / _pairwise_CD_  :112-199   -> (N_sample, N_ref) matrices   ldt_chamfer_pairwise / ldt_emd_approx
    knn(Mxx, Mxy, Myy, k)              :202-231   1-NN two-sample test
    lgan_mmd_cov(all_dist)             :234-246   MMD / COV
    compute_all_metrics / compute_CD_metrics / compute_MMD_metrics  :249-321

The O(N_sample * N_ref * n * m) work — every cloud pair's Chamfer and EMD — runs in two HIP kernels, one workgroup per
pair with both clouds in LDS; the reference loops over pairs in Python and materialises (B, n, m) matrices.  The
bookkeeping on the resulting (N, N) matrices (min / unique / top-k) is plain torch on the device.  `batch_size` is
accepted for signature parity and ignored (no intermediate is large enough to need chunking).  There is no CPU path.
"""
import torch

from . import ops


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError("metrics: tensors are on %s; the HIP path has no CPU fallback (.cuda() them)" % t.device)
    return t.float().contiguous()


def distChamfer(a, b):
    return ops.chamfer(_dev(a), _dev(b))


def L2_ChamferEval_1000(array1, array2):
    """completion_trainer/Latent_SDE_Trainer.py:41-44: (mean dist1 + mean dist2) * 1000 over the whole batch."""
    dist1, dist2 = distChamfer(array1, array2)
    return (torch.mean(dist1) + torch.mean(dist2)) * 1000


def F1Score(array1, array2, threshold=0.001):
    """completion_trainer/Latent_SDE_Trainer.py:47-53: per-cloud F1 of the two nearest-neighbour precisions at `threshold`
    (squared distance); 0/0 -> 0.  Returns (fscore, precision_1, precision_2)."""
    dist1, dist2 = distChamfer(array1, array2)
    precision_1 = torch.mean((dist1 < threshold).float(), dim=1)
    precision_2 = torch.mean((dist2 < threshold).float(), dim=1)
    fscore = 2 * precision_1 * precision_2 / (precision_1 + precision_2)
    fscore[torch.isnan(fscore)] = 0
    return fscore, precision_1, precision_2


def emd_approx(sample, ref):
    """emd_approx_cuda (:40-46): approximate-matching cost / n, one value per cloud pair (sample[b], ref[b])."""
    B, N, N_ref = sample.size(0), sample.size(1), ref.size(1)
    assert N == N_ref, "Not sure what would EMD do in this case"
    return ops.emd_approx(_dev(sample), _dev(ref)) / float(N)


def EMD_CD(sample_pcs, ref_pcs, batch_size=None, accelerated_cd=True, reduced=True, accelerated_emd=True):
    N_sample, N_ref = sample_pcs.shape[0], ref_pcs.shape[0]
    assert N_sample == N_ref, "REF:%d SMP:%d" % (N_ref, N_sample)
    dl, dr = distChamfer(sample_pcs, ref_pcs)
    cd = dl.mean(dim=1) + dr.mean(dim=1)
    emd = emd_approx(sample_pcs, ref_pcs)
    if reduced:
        cd, emd = cd.mean(), emd.mean()
    return {"mmd-CD": cd, "mmd-EMD": emd}


def _pairwise_CD_(sample_pcs, ref_pcs, batch_size=None, verbose=True):
    return ops.chamfer_pairwise(_dev(sample_pcs), _dev(ref_pcs))                 # (N_sample, N_ref)


def _pairwise_EMD_CD_(sample_pcs, ref_pcs, batch_size=None, accelerated_cd=True, accelerated_emd=True):
    s, r = _dev(sample_pcs), _dev(ref_pcs)
    assert s.shape[1] == r.shape[1], "Not sure what would EMD do in this case"
    return ops.chamfer_pairwise(s, r), ops.emd_approx(s, r, pairwise=True) / float(s.shape[1])


def knn(Mxx, Mxy, Myy, k, sqrt=False):
    """:202-231 (upstream: GAN-Metrics) — leave-one-out k-NN two-sample test on the stacked distance matrix of
    {x: label 1} U {y: label 0}: a point is predicted "x" when at least k/2 of its k nearest others are x."""
    n0, n1 = Mxx.shape[0], Myy.shape[0]
    is_x = torch.zeros(n0 + n1, dtype=Mxx.dtype, device=Mxx.device)
    is_x[:n0] = 1.
    D = torch.cat([torch.cat([Mxx, Mxy], dim=1), torch.cat([Mxy.t(), Myy], dim=1)], dim=0)
    if sqrt:
        D = D.abs().sqrt()
    D = D.clone()
    D.fill_diagonal_(float("inf"))                               # a point is not its own neighbour
    nearest = D.topk(k, dim=0, largest=False).indices            # [k, n0+n1], per column
    votes = is_x[nearest].sum(dim=0)
    pred = (votes >= k / 2.).to(Mxx.dtype)
    tp, fp = (pred * is_x).sum(), (pred * (1 - is_x)).sum()
    fn, tn = ((1 - pred) * is_x).sum(), ((1 - pred) * (1 - is_x)).sum()
    return {"tp": tp, "fp": fp, "fn": fn, "tn": tn, "precision": tp / (tp + fp + 1e-10), "recall": tp / (tp + fn + 1e-10),
            "acc": (pred == is_x).float().mean()}


def lgan_mmd_cov(all_dist):
    """:234-246 — all_dist (N_sample, N_ref).  MMD: mean over reference clouds of the distance to their nearest sample;
    COV: fraction of reference clouds that are the nearest reference of some sample."""
    n_ref = all_dist.shape[1]
    matched = all_dist.argmin(dim=1).unique().numel()
    return {"mmd": all_dist.min(dim=0).values.mean(), "cov": torch.tensor(matched / float(n_ref)).to(all_dist)}


def _mmd_cov(results, M_rs, tag):
    results.update({"%s-%s" % (k, tag): v for k, v in lgan_mmd_cov(M_rs.t()).items()})


def _one_nn(results, M_rr, M_rs, M_ss, tag):
    results.update({"1-NN-%s-%s" % (tag, k): v for k, v in knn(M_rr, M_rs, M_ss, 1, sqrt=False).items() if "acc" in k})


def _report(results):
    for k, v in results.items():
        print("[%s] %.8f" % (k, v.item()))


def _mmd_part(sample_pcs, ref_pcs):
    results = {}
    M_rs_cd, M_rs_emd = _pairwise_EMD_CD_(ref_pcs, sample_pcs)
    _mmd_cov(results, M_rs_cd, "CD")
    _mmd_cov(results, M_rs_emd, "EMD")
    _report(results)
    return results, M_rs_cd, M_rs_emd


def compute_MMD_metrics(sample_pcs, ref_pcs, batch_size=None, accelerated_cd=True, accelerated_emd=True):
    """:280-296 — MMD / COV under CD and EMD (rows = reference clouds, columns = samples, then transposed)."""
    return _mmd_part(sample_pcs.cuda(), ref_pcs.cuda())[0]


def compute_all_metrics(sample_pcs, ref_pcs, batch_size=None, accelerated_cd=True, accelerated_emd=True):
    """:249-277 — MMD / COV plus the 1-NN accuracies under CD and EMD (keys as upstream)."""
    ref_pcs, sample_pcs = ref_pcs.cuda(), sample_pcs.cuda()
    results, M_rs_cd, M_rs_emd = _mmd_part(sample_pcs, ref_pcs)
    M_rr_cd, M_rr_emd = _pairwise_EMD_CD_(ref_pcs, ref_pcs)
    M_ss_cd, M_ss_emd = _pairwise_EMD_CD_(sample_pcs, sample_pcs)
    _one_nn(results, M_rr_cd, M_rs_cd, M_ss_cd, "CD")
    _one_nn(results, M_rr_emd, M_rs_emd, M_ss_emd, "EMD")
    return results


def compute_CD_metrics(sample_pcs, ref_pcs, batch_size=None):
    """:299-321 — the Chamfer-only subset."""
    results = {}
    M_rs_cd = _pairwise_CD_(ref_pcs, sample_pcs)
    _mmd_cov(results, M_rs_cd, "CD")
    _report(results)
    _one_nn(results, _pairwise_CD_(ref_pcs, ref_pcs), M_rs_cd, _pairwise_CD_(sample_pcs, sample_pcs), "CD")
    return results
