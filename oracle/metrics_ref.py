"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the numeric core of the reference's quality metrics.

  FeatureStatsRef                 metrics/metric_utils.py:79-155 (float64 running mean / second moment, optional capture_all)
  fid_from_stats                  metrics/frechet_inception_distance.py:41-45
  pairwise_distances              metrics/precision_recall.py:19-32 (torch.cdist, Euclidean, GEMM form for > 25 points)
  precision_recall_from_features  metrics/precision_recall.py:36-85 (k-th neighbour radii, manifold membership)

Not restated: the detector networks (NVIDIA-hosted Inception / VGG16 pickles, metric_utils.py:46-60) -- network downloads that
are not available offline: **parity unpinned** for the feature extractor; everything downstream of the features is pinned by
tests/golden/metrics.npz (made by running the reference's own compute_fid / compute_pr on injected features).

Arithmetic notes: the reference casts the P&R features to float16 and calls torch.cdist on the device in float16; half cdist
does not exist on the CPU, so the pinned definition is "features rounded to float16, distances in float32, radii rounded to
float16 (precision_recall.py:78), membership test dist <= radius".
"""
import numpy as np
import scipy.linalg
import torch


class FeatureStatsRef:
    def __init__(self, capture_all=False, capture_mean_cov=False, max_items=None):
        self.capture_all, self.capture_mean_cov, self.max_items = capture_all, capture_mean_cov, max_items
        self.num_items = 0
        self.num_features = None
        self.all_features = []
        self.raw_mean = None
        self.raw_cov = None

    def is_full(self):
        return self.max_items is not None and self.num_items >= self.max_items

    def append(self, x):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == 2
        if self.max_items is not None and self.num_items + x.shape[0] > self.max_items:
            if self.num_items >= self.max_items:
                return
            x = x[:self.max_items - self.num_items]
        if self.num_features is None:
            self.num_features = x.shape[1]
            self.raw_mean = np.zeros([self.num_features], dtype=np.float64)
            self.raw_cov = np.zeros([self.num_features, self.num_features], dtype=np.float64)
        assert x.shape[1] == self.num_features
        self.num_items += x.shape[0]
        if self.capture_all:
            self.all_features.append(x)
        if self.capture_mean_cov:
            x64 = x.astype(np.float64)
            self.raw_mean += x64.sum(axis=0)
            self.raw_cov += x64.T @ x64

    def get_all(self):
        return np.concatenate(self.all_features, axis=0)

    def get_mean_cov(self):
        mean = self.raw_mean / self.num_items
        cov = self.raw_cov / self.num_items - np.outer(mean, mean)
        return mean, cov


def fid_from_stats(mu_real, sigma_real, mu_gen, sigma_gen):
    m = np.square(mu_gen - mu_real).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_gen, sigma_real), disp=False)
    return float(np.real(m + np.trace(sigma_gen + sigma_real - s * 2)))


def pairwise_distances(rows, cols):
    """Euclidean distances [n_rows, n_cols] in float32 (torch.cdist, as the reference calls it)."""
    return torch.cdist(rows.float().unsqueeze(0), cols.float().unsqueeze(0))[0]


def precision_recall_from_features(real, gen, nhood_size=3, row_batch_size=10000, col_batch_size=10000):
    real = torch.as_tensor(real).to(torch.float16).float()
    gen = torch.as_tensor(gen).to(torch.float16).float()
    out = {}
    for name, manifold, probes in (('precision', real, gen), ('recall', gen, real)):
        kth = []
        for mb in manifold.split(row_batch_size):
            dist = torch.cat([pairwise_distances(mb, cb) for cb in manifold.split(col_batch_size)], dim=1)
            kth.append(dist.kthvalue(nhood_size + 1).values.to(torch.float16))
        kth = torch.cat(kth)
        pred = []
        for pb in probes.split(row_batch_size):
            dist = torch.cat([pairwise_distances(pb, cb) for cb in manifold.split(col_batch_size)], dim=1)
            pred.append((dist <= kth).any(dim=1))
        out[name] = float(torch.cat(pred).float().mean())
        out[name + '_kth'] = kth.float().numpy()
        out[name + '_pred'] = torch.cat(pred).numpy()
    return out
