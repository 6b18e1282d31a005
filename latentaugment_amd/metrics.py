"""Host-side mirror of the reference's quality metrics (metrics/) over the HIP C ABI -- everything downstream of the
detector features.

  FeatureStats                 metrics/metric_utils.py:79-155   (same fields, append / append_torch / get_all / get_mean_cov,
                                                                 save / load of the reference's pickle layout)
  compute_fid_from_stats       metrics/frechet_inception_distance.py:41-45
  compute_distances            metrics/precision_recall.py:19-32
  compute_pr_from_features     metrics/precision_recall.py:72-85

The detectors themselves (Inception-v3 and VGG16 pickles hosted by NVIDIA, metric_utils.py:46-60) cannot be fetched offline:
callers supply the features, e.g. from `synthesis.FeatureEngine` or from a detector they have on disk.
torch only owns the device memory; the moments, distances, radii and membership tests are HIP kernels (la_metrics.hip).
"""
import pickle

import numpy as np
import scipy.linalg
import torch

from . import _lib


class FeatureStats:
    """Running feature statistics.  Device tensors handed to `append_torch` are accumulated ON the GPU (float64
    accumulators, `la_feature_moments_f64`); numpy input goes through the same kernel after an upload."""

    def __init__(self, capture_all=False, capture_mean_cov=False, max_items=None, device='cuda:0'):
        self.capture_all = capture_all
        self.capture_mean_cov = capture_mean_cov
        self.max_items = max_items
        self.num_items = 0
        self.num_features = None
        self.all_features = None
        self._dev = torch.device(device)
        self._mean = None      # float64 device accumulators
        self._cov = None

    def set_num_features(self, num_features):
        if self.num_features is not None:
            assert num_features == self.num_features
            return
        self.num_features = num_features
        self.all_features = []
        if self.capture_mean_cov:
            self._mean = torch.zeros([num_features], dtype=torch.float64, device=self._dev)
            self._cov = torch.zeros([num_features, num_features], dtype=torch.float64, device=self._dev)

    def is_full(self):
        return (self.max_items is not None) and (self.num_items >= self.max_items)

    def append_torch(self, x, num_gpus=1, rank=0):
        assert isinstance(x, torch.Tensor) and x.ndim == 2
        assert 0 <= rank < num_gpus
        if num_gpus > 1:      # interleave the ranks' samples, as the reference does with broadcasts (metric_utils.py:120-128)
            ys = [torch.empty_like(x) for _ in range(num_gpus)]
            torch.distributed.all_gather(ys, x.contiguous())
            x = torch.stack(ys, dim=1).flatten(0, 1)
        _lib.require_gpu(x)
        x = x.detach().to(torch.float32).contiguous()
        if (self.max_items is not None) and (self.num_items + x.shape[0] > self.max_items):
            if self.num_items >= self.max_items:
                return
            x = x[:self.max_items - self.num_items].contiguous()
        self.set_num_features(x.shape[1])
        self.num_items += x.shape[0]
        if self.capture_all:
            self.all_features.append(x.cpu().numpy())
        if self.capture_mean_cov:
            lib = _lib.load()
            with torch.cuda.device(x.device):
                _lib.check(lib.la_feature_moments_f64(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(self._mean), _lib.ptr(self._cov),
                                                      _lib.stream_ptr()), 'feature_moments')

    def append(self, x):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == 2
        self.append_torch(torch.from_numpy(x).to(self._dev))

    def get_all(self):
        assert self.capture_all
        return np.concatenate(self.all_features, axis=0)

    def get_all_torch(self):
        return torch.from_numpy(self.get_all())

    @property
    def raw_mean(self):
        return None if self._mean is None else self._mean.cpu().numpy()

    @property
    def raw_cov(self):
        return None if self._cov is None else self._cov.cpu().numpy()

    def get_mean_cov(self):
        assert self.capture_mean_cov
        mean = self.raw_mean / self.num_items
        cov = self.raw_cov / self.num_items
        cov = cov - np.outer(mean, mean)
        return mean, cov

    def save(self, pkl_file):
        """The reference's cache layout: a pickle of the object's fields (metric_utils.py:138-140)."""
        d = dict(capture_all=self.capture_all, capture_mean_cov=self.capture_mean_cov, max_items=self.max_items,
                 num_items=self.num_items, num_features=self.num_features, all_features=self.all_features,
                 raw_mean=self.raw_mean, raw_cov=self.raw_cov)
        with open(pkl_file, 'wb') as f:
            pickle.dump(d, f)

    @staticmethod
    def load(pkl_file, device='cuda:0'):
        from .formats import _restricted_load          # cache files are data (dict of numpy / python values): never a full unpickle
        with open(pkl_file, 'rb') as f:
            s = _restricted_load(f)
        obj = FeatureStats(capture_all=s['capture_all'], capture_mean_cov=s.get('capture_mean_cov', False),
                           max_items=s['max_items'], device=device)
        obj.num_items, obj.num_features, obj.all_features = s['num_items'], s['num_features'], s['all_features']
        if s.get('raw_mean') is not None and obj.capture_mean_cov:
            obj._mean = torch.from_numpy(np.asarray(s['raw_mean'], dtype=np.float64)).to(obj._dev)
            obj._cov = torch.from_numpy(np.asarray(s['raw_cov'], dtype=np.float64)).to(obj._dev)
        return obj


def compute_fid_from_stats(mu_real, sigma_real, mu_gen, sigma_gen):
    """Frechet distance of two Gaussians (the matrix square root runs on the host with scipy, as in the reference)."""
    m = np.square(mu_gen - mu_real).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_gen, sigma_real), disp=False)
    return float(np.real(m + np.trace(sigma_gen + sigma_real - s * 2)))


def _f16_padded(x, dev):
    """float16 [n][D'] on the device, D' = D rounded up to a multiple of 16 with zero columns (distances unchanged)."""
    x = torch.as_tensor(x).to(dev).to(torch.float16)
    assert x.ndim == 2
    pad = -x.shape[1] % 16
    if pad:
        x = torch.nn.functional.pad(x, [0, pad])
    return x.contiguous()


def compute_distances(row_features, col_features, num_gpus=1, rank=0, col_batch_size=None, device='cuda:0'):
    """Euclidean distance matrix [rows, cols] (float32, on the host like the reference's rank-0 result)."""
    assert num_gpus == 1 and rank == 0, 'the metric path is single-process in every reference driver (SURVEY 2c)'
    dev = torch.device(device)
    lib = _lib.load()
    r, c = _f16_padded(row_features, dev), _f16_padded(col_features, dev)
    dist = torch.empty([r.shape[0], c.shape[0]], dtype=torch.float32, device=dev)
    ws = torch.empty([lib.la_pr_workspace_floats(r.shape[0], c.shape[0])], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):          # the stream must be `dev`'s, not the current device's
        _lib.check(lib.la_cdist_f16(_lib.ptr(r), r.shape[0], _lib.ptr(c), c.shape[0], r.shape[1], _lib.ptr(dist), _lib.ptr(ws),
                                    _lib.stream_ptr()), 'cdist')
    return dist.cpu()


def compute_pr_from_features(real_features, gen_features, nhood_size=3, row_batch_size=10000, col_batch_size=10000,
                             device='cuda:0', return_details=False):
    """(precision, recall) of `gen_features` against `real_features`.  The batch sizes are accepted for interface parity; the
    kernels stream over the columns and never build the distance matrix, so the result does not depend on them."""
    dev = torch.device(device)
    lib = _lib.load()
    feats = {'real': _f16_padded(real_features, dev), 'gen': _f16_padded(gen_features, dev)}
    results, details = {}, {}
    for name, mk, pk in (('precision', 'real', 'gen'), ('recall', 'gen', 'real')):
        manifold, probes = feats[mk], feats[pk]
        nm, npb, D = manifold.shape[0], probes.shape[0], manifold.shape[1]
        ws = torch.empty([lib.la_pr_workspace_floats(max(nm, npb), nm)], dtype=torch.float32, device=dev)
        kth = torch.empty([nm], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.la_pr_kth_f16(_lib.ptr(manifold), nm, _lib.ptr(manifold), nm, D, nhood_size, _lib.ptr(kth), _lib.ptr(ws),
                                         _lib.stream_ptr()), 'pr_kth')
        kth = kth.to(torch.float16).to(torch.float32)          # the reference keeps the radii in float16 (precision_recall.py:78)
        member = torch.empty([npb], dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.la_pr_member_f16(_lib.ptr(probes), npb, _lib.ptr(manifold), nm, D, _lib.ptr(kth), _lib.ptr(member),
                                            _lib.ptr(ws), _lib.stream_ptr()), 'pr_member')
        results[name] = float(member.to(torch.float32).mean())
        details[name + '_kth'] = kth.cpu().numpy()
        details[name + '_pred'] = member.cpu().numpy().astype(bool)
    if return_details:
        return results['precision'], results['recall'], details
    return results['precision'], results['recall']
