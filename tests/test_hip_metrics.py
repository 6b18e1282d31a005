"""HIP metric kernels (la_metrics.hip) against the oracle and against tests/golden/metrics.npz (made by the reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as mr

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('case', ['a', 'b'])
def test_feature_stats_and_fid_vs_reference_golden(dev, case):
    from latentaugment_amd import metrics
    real, gen = GOLD[f'{case}_real'], GOLD[f'{case}_gen']
    sr = metrics.FeatureStats(capture_mean_cov=True, capture_all=True)
    sg = metrics.FeatureStats(capture_mean_cov=True)
    for i in range(0, real.shape[0], 50):
        sr.append_torch(torch.from_numpy(real[i:i + 50]).to(dev))
    for i in range(0, gen.shape[0], 64):
        sg.append(gen[i:i + 64])
    mu_r, sig_r = sr.get_mean_cov()
    np.testing.assert_allclose(mu_r, GOLD[f'{case}_mu_real'], rtol=1e-12, atol=1e-13)      # float64 accumulators: summation order only
    np.testing.assert_allclose(sig_r, GOLD[f'{case}_sigma_real'], rtol=1e-9, atol=1e-11)
    assert sr.get_all().shape == real.shape
    fid = metrics.compute_fid_from_stats(mu_r, sig_r, *sg.get_mean_cov())
    assert fid == pytest.approx(float(GOLD[f'{case}_fid']), rel=1e-8)


@pytest.mark.parametrize('case', ['a', 'b'])
def test_precision_recall_vs_reference_golden(dev, case):
    from latentaugment_amd import metrics
    real, gen = GOLD[f'{case}_real'], GOLD[f'{case}_gen']
    p, r, det = metrics.compute_pr_from_features(real, gen, nhood_size=3, return_details=True)
    ref = mr.precision_recall_from_features(real, gen, nhood_size=3)
    # radii: float32 sums of exact fp16 products in another order, then rounded to float16 like the reference's
    for name in ('precision', 'recall'):
        np.testing.assert_allclose(det[name + '_kth'], ref[name + '_kth'], rtol=2e-3, atol=0)
        assert (det[name + '_pred'] != ref[name + '_pred']).mean() <= 0.005      # a radius one fp16 ulp off can flip a borderline probe
    assert p == pytest.approx(float(GOLD[f'{case}_precision']), abs=0.005)
    assert r == pytest.approx(float(GOLD[f'{case}_recall']), abs=0.005)
    d = metrics.compute_distances(real[:40], gen[:50])
    np.testing.assert_allclose(d.numpy(), GOLD[f'{case}_dist40x50'], rtol=1e-4, atol=1e-4)


def test_pr_properties_large(dev):
    """Size-independent properties at a realistic size: a set is fully inside its own manifold (precision = recall = 1),
    far-away probes are outside (0), and the kth radius of a duplicated set is 0 for k = 1."""
    from latentaugment_amd import metrics
    g = torch.Generator().manual_seed(3)
    x = torch.randn([3000, 256], generator=g)
    p, r = metrics.compute_pr_from_features(x, x, nhood_size=3)
    assert p == 1.0 and r == 1.0
    p, r = metrics.compute_pr_from_features(x, x + 100.0, nhood_size=3)
    assert p == 0.0 and r == 0.0
    xx = torch.cat([x[:500], x[:500]])
    _, _, det = metrics.compute_pr_from_features(xx, xx, nhood_size=1, return_details=True)
    assert float(np.abs(det['precision_kth']).max()) < 0.2        # self + duplicate: both ~0 up to fp32 cancellation of |a|^2+|b|^2-2ab


def test_metrics_misuse(dev):
    from latentaugment_amd import _lib, metrics
    with pytest.raises(_lib.LatentAugHipError):
        metrics.compute_pr_from_features(torch.randn(3, 16), torch.randn(3, 16), nhood_size=5)      # fewer points than neighbours
    st = metrics.FeatureStats(capture_mean_cov=True, max_items=10)
    st.append(np.zeros([8, 4], np.float32))
    st.append(np.zeros([8, 4], np.float32))
    assert st.num_items == 10 and st.is_full()
