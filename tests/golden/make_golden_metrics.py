#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by RUNNING THE REFERENCE's metric code in the build container.

Run from the repo root:   python tests/golden/make_golden_metrics.py      (needs /root/reference, read-only)

Executed from the reference (imported, never copied): metrics/frechet_inception_distance.py::compute_fid,
metrics/precision_recall.py::compute_pr / compute_distances, metrics/metric_utils.py::FeatureStats.
The detector networks are NVIDIA-hosted downloads, so the three `compute_feature_stats_for_*` providers are replaced by
functions that return reference FeatureStats objects filled (through the reference's own `append`) with the synthetic
features stored in the fixture.  One shim: `torch.cdist` has no float16 CPU kernel, so inside precision_recall it is
wrapped to compute in float32 on the float16-rounded features the reference hands it.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'models', 'stylegan3'))

from metrics import frechet_inception_distance, metric_utils, precision_recall  # noqa: E402


def features(seed, n, d, shift=0.0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn([n, d], generator=g)
    mix = torch.randn([d, d], generator=g) / d ** 0.5
    return ((base @ mix) * scale + shift).numpy().astype(np.float32)


def stats_provider(feats, batch=37):
    def provide(opts=None, capture_all=False, capture_mean_cov=False, max_items=None, **_):
        st = metric_utils.FeatureStats(capture_all=capture_all, capture_mean_cov=capture_mean_cov, max_items=max_items)
        for i in range(0, feats.shape[0], batch):
            st.append(feats[i:i + batch])
        return st
    return provide


class _HalfSafeTorch:
    """torch with cdist computed in float32 when handed float16 (no Half CPU kernel)."""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def cdist(a, b, *args, **kw):
        return torch.cdist(a.float(), b.float(), *args, **kw)


def main():
    out = {}
    for case, (n_real, n_gen, d, shift, scale) in {'a': (300, 260, 64, 0.3, 1.2), 'b': (513, 700, 96, 0.05, 0.9)}.items():
        real, gen = features(10, n_real, d), features(11, n_gen, d, shift, scale)
        opts = types.SimpleNamespace(rank=0, num_gpus=1, device=torch.device('cpu'), mode_dict=None, dataset_kwargs_gen=True)
        metric_utils.compute_feature_stats_for_dataset = stats_provider(real)
        metric_utils.compute_feature_stats_for_aug_dataset = stats_provider(gen)
        fid = frechet_inception_distance.compute_fid(opts, max_real=None, num_gen=None)
        st = stats_provider(real)(capture_mean_cov=True)
        mu, sigma = st.get_mean_cov()
        precision_recall.torch = _HalfSafeTorch()
        pr = {}
        for rb, cb in ((10000, 10000), (128, 100)):          # one batch, and ragged row / column batches
            pr[(rb, cb)] = precision_recall.compute_pr(opts, max_real=None, num_gen=None, nhood_size=3, row_batch_size=rb, col_batch_size=cb)
        assert pr[(10000, 10000)] == pr[(128, 100)], pr
        dist = precision_recall.compute_distances(torch.from_numpy(real[:40]).half(), torch.from_numpy(gen[:50]).half(), 1, 0, 16)
        precision_recall.torch = torch
        out.update({f'{case}_real': real, f'{case}_gen': gen, f'{case}_fid': np.float64(fid), f'{case}_mu_real': mu,
                    f'{case}_sigma_real': sigma, f'{case}_precision': np.float64(pr[(10000, 10000)][0]),
                    f'{case}_recall': np.float64(pr[(10000, 10000)][1]), f'{case}_dist40x50': dist.numpy()})
        print(case, 'fid', fid, 'precision/recall', pr[(10000, 10000)])
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)


if __name__ == '__main__':
    main()
