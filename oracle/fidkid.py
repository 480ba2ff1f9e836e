"""TEST INFRASTRUCTURE (oracle): numpy / scipy restatement of the FID / KID statistics of the reference's evaluation.

  * KID: `FIDKID._calc_kid` (npcd/utils/fidkid.py:58-82 of the reference): cubic polynomial kernel MMD over `num_subsets` random
    subsets of size m = min(n_real, n_fake, max_subset_size), sampled with np.random.choice; reported x 1000 (:105).
  * FID: the base class `mmgen.core.evaluation.metrics.FID._calc_fid` -- mmgeneration is a third-party dependency ABSENT from
    /root/reference (README.md:30 pins v0.7.2); its published algorithm is restated here: Frechet distance between two Gaussians,
    ||mu_f - mu_r||^2 + Tr(C_f) + Tr(C_r) - 2 Tr((C_f C_r)^(1/2)) with scipy.linalg.sqrtm, the real part taken, a 1e-6 ridge on both
    covariances when the square root is not finite; the three numbers it returns are (fid, mean term, trace term).
Parity status: the KID restatement follows the reference's own lines; the FID half is anchored on the published formula only
("parity unpinned" for mmgen's numerical corner cases: no fixture can be generated without the package)."""
import numpy as np
import scipy.linalg


def calc_fid(fake_mean, fake_cov, real_mean, real_cov, eps=1e-6):
    cov_sqrt, _ = scipy.linalg.sqrtm(fake_cov.dot(real_cov), disp=False)
    if not np.isfinite(cov_sqrt).all():
        offset = np.eye(fake_cov.shape[0]) * eps
        cov_sqrt = scipy.linalg.sqrtm((fake_cov + offset).dot(real_cov + offset))
    if np.iscomplexobj(cov_sqrt):
        cov_sqrt = cov_sqrt.real
    diff = fake_mean - real_mean
    mean_norm = diff.dot(diff)
    trace = np.trace(fake_cov) + np.trace(real_cov) - 2 * np.trace(cov_sqrt)
    return float(mean_norm + trace), float(mean_norm), float(trace)


def calc_kid(real_feat, fake_feat, num_subsets, max_subset_size, rng=np.random):
    n = real_feat.shape[1]
    m = min(min(real_feat.shape[0], fake_feat.shape[0]), max_subset_size)
    t = 0
    for _ in range(num_subsets):
        x = fake_feat[rng.choice(fake_feat.shape[0], m, replace=False)]
        y = real_feat[rng.choice(real_feat.shape[0], m, replace=False)]
        a = (x @ x.T / n + 1) ** 3 + (y @ y.T / n + 1) ** 3
        b = (x @ y.T / n + 1) ** 3
        t += (a.sum() - np.diag(a).sum()) / (m - 1) - b.sum() * 2 / m
    return float(t / num_subsets / m)
