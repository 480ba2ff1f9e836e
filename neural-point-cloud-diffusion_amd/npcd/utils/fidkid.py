"""FID / KID of the reference's diffusion evaluation (npcd/utils/fidkid.py:34-108 -> mmgen FID, pinned at mmgeneration v0.7.2 by the
reference's README.md:30; used by npcd/eval/diffusion_evaluation.py:122-126,179,183).

What is here: the STATISTICS -- feature accumulation, the reference statistics pickle (`mean`, `cov`, `feats_np`), the Frechet
distance and the kernel inception distance (cubic polynomial kernel, random subsets, x 1000) -- in float64.  What is NOT here: the
Inception-v3 network and its weights (`data/inception-2015-12-05.pt`, the per-dataset `*_inception_stylegan.pkl`): they are not part
of the reference repository and not available in this environment, so `feed` takes FEATURES, or images plus a caller-supplied
`feature_extractor`; without one it raises instead of guessing.  Hook for npcd.eval.sample_and_render:
`feed=lambda images: fidkid.feed(images * 2 - 1, "fakes")` (diffusion_evaluation.py:179)."""
import pickle
from typing import Callable, Optional

import numpy as np
import torch


def frechet_distance(fake_mean, fake_cov, real_mean, real_cov, eps: float = 1e-6):
    """(fid, mean term, trace term) = ||mu_f - mu_r||^2 + Tr C_f + Tr C_r - 2 Tr (C_f C_r)^(1/2), float64.  The trace of the matrix square
    root is the sum of the square roots of the eigenvalues of C_f^(1/2) C_r C_f^(1/2) (symmetric positive semi-definite: real, no
    complex square root to discard as in the scipy.linalg.sqrtm form of mmgen); a 1e-6 ridge when that is not finite."""
    fm, fc = torch.as_tensor(fake_mean, dtype=torch.float64), torch.as_tensor(fake_cov, dtype=torch.float64)
    rm, rc = torch.as_tensor(real_mean, dtype=torch.float64), torch.as_tensor(real_cov, dtype=torch.float64)

    def tr_sqrt(a, b):
        w, v = torch.linalg.eigh((a + a.T) / 2)
        root = (v * w.clamp_min(0).sqrt()) @ v.T
        m = root @ b @ root
        return torch.linalg.eigvalsh((m + m.T) / 2).clamp_min(0).sqrt().sum()

    t = tr_sqrt(fc, rc)
    if not bool(torch.isfinite(t)):
        ridge = torch.eye(fc.shape[0], dtype=torch.float64) * eps
        t = tr_sqrt(fc + ridge, rc + ridge)
    diff = fm - rm
    mean_norm = float(diff.dot(diff))
    trace = float(torch.trace(fc) + torch.trace(rc) - 2 * t)
    return mean_norm + trace, mean_norm, trace


def kernel_inception_distance(real_feat: np.ndarray, fake_feat: np.ndarray, num_subsets: int = 100, max_subset_size: int = 1000, rng=np.random) -> float:
    """reference fidkid.py:58-82 (without the x 1000 of :105): mean over `num_subsets` random subsets of the unbiased cubic-kernel MMD."""
    n = real_feat.shape[1]
    m = min(min(real_feat.shape[0], fake_feat.shape[0]), max_subset_size)
    t = 0.0
    for _ in range(num_subsets):
        x = torch.from_numpy(fake_feat[rng.choice(fake_feat.shape[0], m, replace=False)]).double()
        y = torch.from_numpy(real_feat[rng.choice(real_feat.shape[0], m, replace=False)]).double()
        a = (x @ x.T / n + 1) ** 3 + (y @ y.T / n + 1) ** 3
        b = (x @ y.T / n + 1) ** 3
        t += float((a.sum() - torch.diagonal(a).sum()) / (m - 1) - b.sum() * 2 / m)
    return t / num_subsets / m


class FIDKID:
    """Same surface as the reference's FIDKID (fidkid.py:34-108): `prepare()` loads the reference statistics, `feed(batch, mode)` with
    mode "reals" / "fakes", `summary()` -> (fid, mean term, cov term, kid x 1000) and `_result_dict` / `_result_str`."""
    name = "FIDKID"

    def __init__(self, num_images: int, num_subsets: int = 100, max_subset_size: int = 1000, inception_pkl: Optional[str] = None,
                 feature_extractor: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, **_):
        self.num_images, self.num_subsets, self.max_subset_size = num_images, num_subsets, max_subset_size
        self.inception_pkl, self.feature_extractor = inception_pkl, feature_extractor
        self.real_feats, self.fake_feats = [], []
        self.real_mean = self.real_cov = self.real_feats_np = None
        self.num_real_feeded = self.num_fake_feeded = 0
        self._result_str, self._result_dict = None, None

    def prepare(self):
        if self.inception_pkl is not None:
            with open(self.inception_pkl, "rb") as f:
                ref = pickle.load(f)
            self.real_mean, self.real_cov, self.real_feats_np = ref["mean"], ref["cov"], ref["feats_np"]
            self.num_real_feeded = self.real_feats_np.shape[0]

    def _features(self, batch: torch.Tensor) -> torch.Tensor:
        if batch.dim() == 2:
            return batch.detach().double().cpu()                     # already features [n, D]
        if self.feature_extractor is None:
            raise RuntimeError("FIDKID.feed was given images but no feature_extractor: the Inception-v3 network of the reference's evaluation "
                               "(data/inception-2015-12-05.pt) is not part of this package -- pass feature_extractor= or feed features [n, D]")
        with torch.no_grad():
            return self.feature_extractor(batch).detach().double().cpu()

    def feed(self, batch: torch.Tensor, mode: str):
        """mode "reals" / "fakes"; at most num_images per side are kept (mmgen Metric.feed)."""
        if mode not in ("reals", "fakes"):
            raise ValueError(mode)
        if mode == "reals" and self.real_feats_np is not None:
            return 0
        have = self.num_real_feeded if mode == "reals" else self.num_fake_feeded
        take = min(batch.shape[0], self.num_images - have)
        if take <= 0:
            return 0
        feats = self._features(batch[:take])
        if mode == "reals":
            self.real_feats.append(feats); self.num_real_feeded += take
        else:
            self.fake_feats.append(feats); self.num_fake_feeded += take
        return take

    def summary(self):
        if self.real_feats_np is None:
            feats = torch.cat(self.real_feats, dim=0)
            assert feats.shape[0] >= self.num_images
            self.real_feats_np = feats[:self.num_images].numpy()
            self.real_mean = np.mean(self.real_feats_np, 0)
            self.real_cov = np.cov(self.real_feats_np, rowvar=False)
        fake = torch.cat(self.fake_feats, dim=0)
        assert fake.shape[0] == self.num_images
        fake_np = fake.numpy()
        fid, mean, cov = frechet_distance(np.mean(fake_np, 0), np.cov(fake_np, rowvar=False), self.real_mean, self.real_cov)
        kid = kernel_inception_distance(self.real_feats_np, fake_np, self.num_subsets, self.max_subset_size) * 1000
        self._result_str = f"{fid:.4f} ({mean:.5f}/{cov:.5f}), {kid:.4f}"
        self._result_dict = dict(fid=fid, fid_mean=mean, fid_cov=cov, kid=kid)
        return fid, mean, cov, kid
