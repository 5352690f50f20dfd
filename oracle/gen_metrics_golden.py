"""TEST INFRASTRUCTURE ONLY — writes tests/golden/metrics_cd.npz from the imported reference.

The reference's own `compute_CD_metrics`, `_pairwise_CD_`, `lgan_mmd_cov`, `knn` and the exact-assignment `emd_approx`
fallback (evaluation/evaluation_metrics.py:47-64,165-246,299-321) run on CPU for two small sets of clouds.  The
approximate-matching EMD (CUDA extension, evaluation/pytorch_structural_losses) cannot run here: unpinned.

    python oracle/gen_metrics_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import as R  # noqa: E402
from oracle.gen_golden import save  # noqa: E402


def main():
    R.setup()
    with R.quiet():
        import evaluation.evaluation_metrics as E
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(17)
    n_pts = 96
    # "reference" set: noisy spheres / cubes; "sample" set: a different mixture, so that MMD/COV/1-NN are not trivial
    def clouds(k, kind):
        p = torch.randn(k, n_pts, 3, generator=g)
        if kind == "sphere":
            p = p / p.norm(dim=-1, keepdim=True) * (0.6 + 0.3 * torch.rand(k, 1, 1, generator=g))
        else:
            p = p.clamp(-1, 1) * (0.4 + 0.4 * torch.rand(k, 1, 3, generator=g))
        return p + 0.02 * torch.randn(k, n_pts, 3, generator=g)

    ref = torch.cat([clouds(7, "sphere"), clouds(6, "cube")], 0)
    smp = torch.cat([clouds(4, "sphere"), clouds(8, "cube")], 0)
    with R.quiet():
        M_rs = E._pairwise_CD_(ref, smp, 5)
        M_rr = E._pairwise_CD_(ref, ref, 5)
        M_ss = E._pairwise_CD_(smp, smp, 5)
        res = E.compute_CD_metrics(smp, ref, 5)
        mc = E.lgan_mmd_cov(M_rs.t())
        k1 = E.knn(M_rr, M_rs, M_ss, 1, sqrt=False)
        k3 = E.knn(M_rr, M_rs, M_ss, 3, sqrt=True)
        emd_exact = E.emd_approx(smp[:6], ref[:6])          # the reference's CPU fallback: exact assignment, mean distance
    save("metrics_cd", ref=ref, smp=smp, M_rs=M_rs, M_rr=M_rr, M_ss=M_ss,
         mmd_cd=res["mmd-CD"], cov_cd=res["cov-CD"], one_nn_cd_acc=res["1-NN-CD-acc"],
         lgan_mmd=mc["mmd"], lgan_cov=mc["cov"], knn1_acc=k1["acc"], knn1_tp=k1["tp"], knn1_fp=k1["fp"], knn1_fn=k1["fn"],
         knn1_tn=k1["tn"], knn3_sqrt_acc=k3["acc"], knn3_sqrt_precision=k3["precision"], knn3_sqrt_recall=k3["recall"],
         emd_exact=emd_exact)


if __name__ == "__main__":
    main()
