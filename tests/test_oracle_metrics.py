"""Oracle (oracle/metrics_ref.py) against tests/golden/metrics.npz, which the reference's own compute_fid / compute_pr produced."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as mr

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))


@pytest.mark.parametrize('case', ['a', 'b'])
def test_fid_and_feature_stats_vs_reference(case):
    real, gen = GOLD[f'{case}_real'], GOLD[f'{case}_gen']
    sr, sg = mr.FeatureStatsRef(capture_mean_cov=True), mr.FeatureStatsRef(capture_mean_cov=True)
    for i in range(0, real.shape[0], 50):
        sr.append(real[i:i + 50])
    for i in range(0, gen.shape[0], 64):
        sg.append(gen[i:i + 64])
    mu_r, sig_r = sr.get_mean_cov()
    np.testing.assert_allclose(mu_r, GOLD[f'{case}_mu_real'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(sig_r, GOLD[f'{case}_sigma_real'], rtol=1e-10, atol=1e-12)
    fid = mr.fid_from_stats(mu_r, sig_r, *sg.get_mean_cov())
    assert fid == pytest.approx(float(GOLD[f'{case}_fid']), rel=1e-9)


@pytest.mark.parametrize('case', ['a', 'b'])
def test_precision_recall_vs_reference(case):
    real, gen = GOLD[f'{case}_real'], GOLD[f'{case}_gen']
    for rb, cb in ((10000, 10000), (97, 130)):
        out = mr.precision_recall_from_features(real, gen, nhood_size=3, row_batch_size=rb, col_batch_size=cb)
        assert out['precision'] == pytest.approx(float(GOLD[f'{case}_precision']), abs=1e-7)
        assert out['recall'] == pytest.approx(float(GOLD[f'{case}_recall']), abs=1e-7)
    d = mr.pairwise_distances(torch.from_numpy(real[:40]).half(), torch.from_numpy(gen[:50]).half())
    np.testing.assert_allclose(d.numpy(), GOLD[f'{case}_dist40x50'], rtol=1e-6, atol=1e-6)


def test_feature_stats_max_items_and_capture_all():
    st = mr.FeatureStatsRef(capture_all=True, capture_mean_cov=True, max_items=70)
    x = np.random.RandomState(0).randn(100, 8).astype(np.float32)
    for i in range(0, 100, 32):
        st.append(x[i:i + 32])
    assert st.num_items == 70 and st.is_full() and st.get_all().shape == (70, 8)
    np.testing.assert_allclose(st.get_mean_cov()[0], x[:70].astype(np.float64).mean(0), rtol=1e-12)
