"""FID / KID statistics (npcd.utils.fidkid, reference npcd/utils/fidkid.py:34-108 on mmgen's FID) against the numpy / scipy oracle
(oracle/fidkid.py) on synthetic features: the Frechet distance in its closed form for commuting covariances, against the
scipy.linalg.sqrtm restatement for general ones, KID with a replayed random stream, the pickle path and the loud failure without an
Inception network."""
import pickle

import numpy as np
import pytest
import torch

from oracle import fidkid as ofk


def _feats(n, d, seed, shift=0.0, scale=1.0):
    g = np.random.RandomState(seed)
    a = g.randn(d, d) / d ** 0.5
    return (g.randn(n, d) @ a) * scale + shift


def test_frechet_distance_closed_form_and_oracle():
    from npcd.utils.fidkid import frechet_distance
    d = 12
    # commuting (diagonal) covariances: sum (sqrt a - sqrt b)^2 + |mu|^2
    a, b = np.linspace(0.5, 3.0, d), np.linspace(2.0, 0.1, d)
    mu = np.linspace(-1, 1, d)
    fid, mean, cov = frechet_distance(mu, np.diag(a), np.zeros(d), np.diag(b))
    assert mean == pytest.approx(float(mu @ mu), rel=1e-12)
    assert cov == pytest.approx(float(((a ** 0.5 - b ** 0.5) ** 2).sum()), rel=1e-10)
    assert fid == pytest.approx(mean + cov, rel=1e-12)
    # general covariances vs the scipy.linalg.sqrtm form (mmgen v0.7.2's published algorithm)
    x, y = _feats(400, d, 1), _feats(300, d, 2, shift=0.3, scale=1.4)
    args = (x.mean(0), np.cov(x, rowvar=False), y.mean(0), np.cov(y, rowvar=False))
    got, ref = frechet_distance(*args), ofk.calc_fid(*args)
    for g_, r_ in zip(got, ref):
        assert g_ == pytest.approx(r_, rel=1e-8, abs=1e-10)
    # identical distributions: zero; rank-deficient covariance (fewer samples than dimensions) stays finite
    z = frechet_distance(args[0], args[1], args[0], args[1])
    assert abs(z[0]) < 1e-8
    small = _feats(5, d, 3)
    f2 = frechet_distance(small.mean(0), np.cov(small, rowvar=False), args[2], args[3])
    assert np.isfinite(f2[0]) and f2[0] == pytest.approx(ofk.calc_fid(small.mean(0), np.cov(small, rowvar=False), args[2], args[3])[0], rel=1e-5)


def test_kid_replays_the_reference_sampling():
    from npcd.utils.fidkid import kernel_inception_distance
    real, fake = _feats(150, 16, 4), _feats(120, 16, 5, shift=0.2)
    got = kernel_inception_distance(real, fake, num_subsets=7, max_subset_size=50, rng=np.random.RandomState(11))
    ref = ofk.calc_kid(real, fake, 7, 50, rng=np.random.RandomState(11))
    assert got == pytest.approx(ref, rel=1e-10)
    same = kernel_inception_distance(real, real.copy(), num_subsets=20, max_subset_size=150, rng=np.random.RandomState(0))
    assert abs(same) < abs(got)                                      # an unbiased estimate around zero for equal distributions


def test_fidkid_object_surface(tmp_path):
    from npcd.utils.fidkid import FIDKID
    real, fake = _feats(64, 8, 6), _feats(64, 8, 7, shift=0.5)
    pkl = tmp_path / "ref.pkl"
    with open(pkl, "wb") as f:
        pickle.dump({"mean": real.mean(0), "cov": np.cov(real, rowvar=False), "feats_np": real}, f)
    m = FIDKID(num_images=64, num_subsets=5, max_subset_size=32, inception_pkl=str(pkl))
    m.prepare()
    assert m.num_real_feeded == 64 and m.feed(torch.from_numpy(real), "reals") == 0      # the pickle stands for the reals
    for i in range(0, 80, 16):
        m.feed(torch.from_numpy(fake[i:i + 16]) if i < 64 else torch.zeros(16, 8), "fakes")
    np.random.seed(3)
    fid, mean, cov, kid = m.summary()
    np.random.seed(3)
    assert fid == pytest.approx(ofk.calc_fid(fake.mean(0), np.cov(fake, rowvar=False), real.mean(0), np.cov(real, rowvar=False))[0], rel=1e-8)
    assert kid == pytest.approx(ofk.calc_kid(real, fake, 5, 32) * 1000, rel=1e-10)
    assert set(m._result_dict) == {"fid", "fid_mean", "fid_cov", "kid"} and m._result_str.count("(") == 1
    # images without an Inception network: loud
    with pytest.raises(RuntimeError, match="feature_extractor"):
        FIDKID(num_images=4).feed(torch.zeros(2, 3, 8, 8), "fakes")
    # ... and with a caller-supplied extractor the same object works on images
    m2 = FIDKID(num_images=4, num_subsets=2, max_subset_size=4, feature_extractor=lambda im: im.flatten(1)[:, :6])
    g = torch.Generator().manual_seed(0)
    m2.feed(torch.randn(4, 3, 4, 4, generator=g), "reals"); m2.feed(torch.randn(4, 3, 4, 4, generator=g), "fakes")
    assert np.isfinite(m2.summary()[0])
