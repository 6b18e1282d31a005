"""Test helpers: write interim zips and network pickles in the reference's on-disk formats (no reference code needed)."""
import collections
import os
import pickle
import sys
import types
import zipfile

import numpy as np
import pytest


def write_zip(path, members):
    with zipfile.ZipFile(path, 'w') as z:
        for name, obj in members.items():
            z.writestr(name, pickle.dumps(obj))


def make_interim(tmp, num_ws=6, w_dim=32, res=16, patients=3):
    """Zips laid out as data/write_tozip.py writes them: <split>/<patient>/<slice id>.pickle, slice ids 00010..00120."""
    rng = np.random.RandomState(0)
    lat, img = {}, {}
    for p in range(patients):
        for sl in range(10, 121, 5):
            name = f'train/p{p:03d}/s_{sl:05d}.pickle'
            lat[name] = rng.randn(num_ws, w_dim).astype('float32')
            img[name] = {'A': rng.randint(0, 256, (res, res)).astype('float32'), 'B': rng.randint(0, 256, (res, res)).astype('float32')}
    lat['val/p900/s_00010.pickle'] = rng.randn(num_ws, w_dim).astype('float32')
    write_zip(os.path.join(tmp, 'w.zip'), lat)
    write_zip(os.path.join(tmp, 'img.zip'), img)
    return lat, img


# ---- network pickles in the reference's persistence format (torch_utils/persistence.py:118-126), written without its code
class _P:
    """Pickles like a persistence-decorated object: reduce -> (torch_utils.persistence._reconstruct_persistent_obj, (meta,))."""

    def __init__(self, class_name, state):
        self.class_name, self.state = class_name, state

    def __reduce__(self):
        fn = sys.modules['torch_utils.persistence']._reconstruct_persistent_obj
        return (fn, (dict(type='class', version=6, module_src='raise RuntimeError("embedded source must never run")',
                          class_name=self.class_name, state=self.state),))


def _to_persistent(m):
    st = {k: v for k, v in m.__dict__.items() if not k.startswith('_') and isinstance(v, (int, float, str, bool, type(None)))}
    st['_parameters'] = collections.OrderedDict(m._parameters)
    st['_buffers'] = collections.OrderedDict(m._buffers)
    st['_modules'] = collections.OrderedDict((k, _to_persistent(v)) for k, v in m._modules.items())
    st['training'] = False
    return _P(type(m).__name__, st)


@pytest.fixture
def fake_persistence_module():
    tu = types.ModuleType('torch_utils')
    pe = types.ModuleType('torch_utils.persistence')

    def _reconstruct_persistent_obj(meta):
        raise AssertionError('the dump side never reconstructs')
    _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
    _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
    pe._reconstruct_persistent_obj = _reconstruct_persistent_obj
    tu.persistence = pe
    old = {k: sys.modules.get(k) for k in ('torch_utils', 'torch_utils.persistence')}
    sys.modules['torch_utils'], sys.modules['torch_utils.persistence'] = tu, pe
    yield
    for k, v in old.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


