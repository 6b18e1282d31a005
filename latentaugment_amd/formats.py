"""On-disk formats either side of the hot path -- host-side I/O only, no arithmetic on the path.

  * LatentCodeDataset / ImgDataset : zip of per-slice pickles, as written by data/write_tozip.py:30-68 and read by
    augments/utils/util_dataset.py:150-279 of the reference (member names "<split>/<patient>/<slice>.pickle").
  * DatasetStats + compute_stats    : the real-data banks W / X (util_dataset.py:35-147, util_latent_aug.py:503-563),
    including the slice schedule 00010..00120 step `step` and the cache pickle layout (`DatasetStats.save/load`).
  * load_network_pkl                : G_ema / D out of a StyleGAN network pickle (util_latent_aug.py:466-484).  The
    reference unpickles with `pickle.load`, which re-executes the class source embedded by
    torch_utils/persistence.py:118-126,179-202.  This loader NEVER executes embedded source: persistent objects are
    rebuilt as inert records (class name + state) and flattened to a state_dict with the names of legacy.py:171-203.
"""
import collections
import io
import os
import pickle
import zipfile

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# zip datasets

def _ext(fname):
    return os.path.splitext(fname)[1].lower()


class _ZipPickles:
    def __init__(self, path, split):
        if _ext(path) != '.zip':
            raise IOError('Path must point to a zip')
        self._path, self._split, self._zipfile = path, split, None
        names = set(self._zip().namelist())
        self._fnames = sorted(f for f in names if _ext(f) == '.pickle' and split in f)
        if not self._fnames:
            raise IOError('No files found in the specified path')

    def _zip(self):
        if self._zipfile is None:
            self._zipfile = zipfile.ZipFile(self._path)
        return self._zipfile

    def open_file(self, fname):
        return self._zip().open(fname, 'r')

    def __len__(self):
        return len(self._fnames)

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_zipfile'] = None
        return d


class LatentCodeDataset(_ZipPickles):
    """util_dataset.py:150-208: each member is a pickled ndarray [num_ws, w_dim] (the inverted latent of one slice)."""

    def __init__(self, path, split, w_dim=512, num_ws=14):
        super().__init__(path, split)
        w0 = self._load_w(0)[0]
        if w_dim is not None and w0.shape[1] != w_dim:
            raise IOError('W does not match the specified latent dimension.')
        if num_ws is not None and w0.shape[0] != num_ws:
            raise IOError('W does not match the specified broadcasting.')

    def _load_w(self, idx):
        fname = self._fnames[idx]
        with self.open_file(fname) as f:
            w = _restricted_load(f)                # plain ndarray pickles written by the inversion step
        return np.asarray(w).astype('float32'), fname

    def __getitem__(self, idx):
        return self._load_w(idx)

    def lookup(self, fname):
        """Latent of one file name, as LatentAugment.sample_from_inversion reads it (latent_aug.py:314-318)."""
        with self.open_file(fname) as f:
            return np.asarray(_restricted_load(f)).astype('float32')


class ImgDataset(_ZipPickles):
    """util_dataset.py:210-279: each member is a pickled dict modality -> 2-D array (raw 0..255); item = CHW float32."""

    def __init__(self, path, split, modalities, resolution=256):
        super().__init__(path, split)
        self._modalities = list(modalities)
        assert len(self._modalities) > 0
        img = self._load_raw_image(0)[0]
        if resolution is not None and (img.shape[1] != resolution or img.shape[2] != resolution):
            raise IOError('Image files do not match the specified resolution')

    def _load_raw_image(self, idx):
        fname = self._fnames[idx]
        with self.open_file(fname) as f:
            p = _restricted_load(f)
        s = p[self._modalities[0]]
        out = np.zeros((len(self._modalities), s.shape[0], s.shape[1]), dtype='float32')
        for i, m in enumerate(self._modalities):
            out[i] = np.asarray(p[m]).astype('float32')
        return out, fname

    def __getitem__(self, idx):
        return self._load_raw_image(idx)


# ------------------------------------------------------------------------------------------------------------------
# banks

class DatasetStats:
    """Real-data bank accumulator with the reference's slice schedule and cache layout (util_dataset.py:35-147)."""
    _NDIM = {'latent': 3, 'features': 4, 'features_jit': 2, 'img': 4}

    def __init__(self, manifold, capture_all=False, max_items=None, step=1):
        if manifold not in self._NDIM:
            raise NotImplementedError('Unrecognised manifold! Add it!')
        self.manifold, self.capture_all, self.max_items, self.step = manifold, capture_all, max_items, step
        self.num_items = 0
        self.all_x = []
        self.schedule = sorted(f'{i:05d}' for i in np.arange(start=10, stop=120 + 1, step=step))
        self.ndim = self._NDIM[manifold]

    def append(self, x, fname):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == self.ndim
        if self.max_items is not None and self.num_items + x.shape[0] > self.max_items:
            if self.num_items >= self.max_items:
                return -1
            x = x[:self.max_items - self.num_items]
        if not self.capture_all:
            stem = os.path.splitext(os.path.basename(fname))[0]
            if stem[-5:] not in self.schedule:             # keep a slice iff its 5-digit id is on the schedule
                return 0
        self.all_x.append(x)
        self.num_items += x.shape[0]
        return x.shape[0]

    def get_all(self):
        return np.concatenate(self.all_x, axis=0)

    def get_all_torch(self):
        return torch.from_numpy(self.get_all().astype(np.float32))

    def save(self, pkl_file):
        with open(pkl_file, 'wb') as f:
            pickle.dump(self.__dict__, f)

    @staticmethod
    def load(pkl_file):
        with open(pkl_file, 'rb') as f:
            s = _restricted_load(f)                       # EasyDict-free: plain dict of numpy / python values
        obj = DatasetStats(manifold=s['manifold'], capture_all=s['capture_all'], max_items=s['max_items'], step=s['step'])
        obj.__dict__.update(s)
        return obj


def compute_stats(dataset, manifold, cache_dir, cache_tag='', step=10, max_items=100000, feature_fn=None):
    """util_latent_aug.py:503-563.  'latent' / 'img' banks, and 'features_jit' (the TorchScript-LPIPS bank, :565-580) when
    `feature_fn(x [1,C,R,R] float32 raw image) -> [1,F]` is given (it crops one modality, repeats it to 3 channels and
    runs the feature net; see LatentAug._lpips_bank_feature_fn)."""
    if manifold not in ('latent', 'img', 'features_jit'):
        raise NotImplementedError(f"manifold {manifold!r}: only 'latent', 'img' and 'features_jit' banks are built on this path")
    if manifold == 'features_jit' and feature_fn is None:
        raise ValueError("manifold 'features_jit' needs feature_fn")
    num_items = len(dataset) if max_items is None else min(len(dataset), max_items)
    os.makedirs(cache_dir, exist_ok=True)
    tag = f'{manifold}-step_{step}-maxitems_{num_items}'
    if cache_tag:
        tag = f'{cache_tag}-{tag}'
    cache_file = os.path.join(cache_dir, tag + '.pkl')
    if os.path.isfile(cache_file):
        return DatasetStats.load(cache_file)
    stats = DatasetStats(manifold=manifold, max_items=num_items, step=step)
    for i in range(len(dataset)):
        x, fname = dataset[i]
        x = np.asarray(x, dtype=np.float32)[None]          # the reference iterates a batch-size-1 DataLoader
        if manifold == 'img':
            x = x / 127.5 - 1                              # synthetic images live in [-1, 1]  (:544)
        elif manifold == 'features_jit':
            stem = os.path.splitext(os.path.basename(fname))[0]
            if stem[-5:] not in stats.schedule:            # do not run the net on slices the schedule drops anyway
                continue
            x = np.asarray(feature_fn(x), dtype=np.float32)
        if stats.append(x, fname) < 0:
            break
    stats.save(cache_file)
    return stats


# ------------------------------------------------------------------------------------------------------------------
# network pickles, without executing the embedded class source

class PersistentRecord:
    """Inert stand-in for an object pickled by torch_utils.persistence: class name + state, nothing executed."""

    def __init__(self, meta):
        self.class_name = meta.get('class_name')
        self.state = dict(meta.get('state') or {})

    def __repr__(self):
        return f'<PersistentRecord {self.class_name}>'


def _reconstruct_persistent_obj(meta):
    return PersistentRecord(dict(meta))


class _EasyDictStub(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


_SAFE_GLOBALS = {
    ('collections', 'OrderedDict'): collections.OrderedDict,
    ('builtins', 'dict'): dict, ('builtins', 'list'): list, ('builtins', 'tuple'): tuple, ('builtins', 'set'): set,
    ('builtins', 'int'): int, ('builtins', 'float'): float, ('builtins', 'bool'): bool, ('builtins', 'str'): str,
    ('builtins', 'slice'): slice, ('builtins', 'bytearray'): bytearray,
    ('dnnlib.util', 'EasyDict'): _EasyDictStub, ('dnnlib', 'EasyDict'): _EasyDictStub,
    ('torch_utils.persistence', '_reconstruct_persistent_obj'): _reconstruct_persistent_obj,
}
_SAFE_TORCH_NAMES = {'_rebuild_tensor_v2', '_rebuild_tensor', '_rebuild_parameter', '_rebuild_parameter_with_state',
                     'FloatStorage', 'HalfStorage', 'DoubleStorage', 'LongStorage', 'IntStorage',
                     'BoolStorage', 'ByteStorage', 'UntypedStorage', 'Size', 'device', 'dtype', 'float32', 'float16',
                     'float64', 'int64', 'int32', 'uint8', 'bool', 'Tensor', 'Parameter', '_rebuild_from_type_v2',
                     'StorageType', 'TypedStorage'}
# numpy reconstruction helpers, by exact (module suffix, name): nothing else of numpy is reachable from a pickle
_SAFE_NUMPY = {('numpy', 'ndarray'), ('numpy', 'dtype'),
               ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
               ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
               ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer')}


def _load_tensor_from_bytes(b):
    """Stand-in for torch.storage._load_from_bytes (which is a full `torch.load(..., weights_only=False)`, i.e. an
    unrestricted unpickle of attacker-controlled bytes): persistence pickles carry their tensors through it, so the entry
    must exist -- but only tensors may come out of it."""
    return torch.load(io.BytesIO(b), weights_only=True, map_location='cpu')


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS:
            return _SAFE_GLOBALS[(module, name)]
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            return _load_tensor_from_bytes
        if (module, name) in _SAFE_NUMPY:
            import importlib
            return getattr(importlib.import_module(module), name)
        if module.startswith('torch') and name in _SAFE_TORCH_NAMES:
            import importlib
            return getattr(importlib.import_module(module), name)
        raise pickle.UnpicklingError(f'refusing to load global {module}.{name}: not on the allow-list of the safe loader')


def _restricted_load(f):
    return _RestrictedUnpickler(f).load()


def _flatten(rec, prefix, out, attrs):
    """Walk a PersistentRecord tree laid out like torch.nn.Module.__dict__ into a flat state_dict."""
    st = rec.state
    for k, v in (st.get('_parameters') or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if isinstance(v, torch.Tensor) else torch.as_tensor(v)
    for k, v in (st.get('_buffers') or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if isinstance(v, torch.Tensor) else torch.as_tensor(v)
    for k, v in st.items():
        if not k.startswith('_') and (isinstance(v, (int, float, str, bool, list, tuple)) or v is None):
            attrs[prefix + k] = v
    for k, m in (st.get('_modules') or {}).items():
        if isinstance(m, PersistentRecord):
            _flatten(m, prefix + k + '.', out, attrs)
        elif isinstance(m, torch.nn.Module):
            for kk, vv in m.state_dict().items():
                out[prefix + k + '.' + kk] = vv


def record_to_state_dict(rec):
    """(state_dict, attrs) of a network record; attrs carries z_dim / w_dim / num_ws / img_resolution / img_channels."""
    out, attrs = collections.OrderedDict(), {}
    _flatten(rec, '', out, attrs)
    return out, attrs


class LoadedNetwork(dict):
    """state_dict of a pickled network plus the scalar attributes the hot path reads (util_latent_aug.py:119-121)."""

    def __init__(self, state_dict, attrs, class_name):
        super().__init__(state_dict)
        self.attrs, self.class_name = attrs, class_name
        for k in ('z_dim', 'w_dim', 'c_dim', 'num_ws', 'img_resolution', 'img_channels'):
            if k in attrs:
                setattr(self, k, attrs[k])


def load_network_pkl(path_or_file):
    """{'G_ema': LoadedNetwork, 'D': LoadedNetwork, ...} from a network-snapshot pickle, executing no embedded code."""
    f = open(path_or_file, 'rb') if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    try:
        data = _restricted_load(f)
    finally:
        if f is not path_or_file:
            f.close()
    out = {}
    for k, v in data.items():
        if isinstance(v, PersistentRecord):
            sd, attrs = record_to_state_dict(v)
            out[k] = LoadedNetwork(sd, attrs, v.class_name)
        else:
            out[k] = v
    return out


def find_network_pkl(model_dir, dataset, dataset_name, modalities, exp_stylegan, network_pkl_stylegan):
    """Path rule of load_stylegan (util_latent_aug.py:466-471)."""
    mods = modalities if isinstance(modalities, str) else ','.join(modalities)
    dir_model = os.path.join(model_dir, dataset, 'training-runs', dataset_name, mods)
    exp = [x for x in os.listdir(dir_model) if exp_stylegan in x]
    assert len(exp) == 1, f'expected exactly one experiment matching {exp_stylegan!r} in {dir_model}'
    return os.path.join(dir_model, exp[0], network_pkl_stylegan)
