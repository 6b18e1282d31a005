"""On-disk formats (zip-of-pickles datasets, bank construction + cache, network pickles) -- CPU only."""
import collections
import io
import os
import pickle
import sys
import types
import zipfile

import numpy as np
import pytest
import torch

from latentaugment_amd import formats
from oracle import sg2_networks as nets
from helpers_formats import _to_persistent, fake_persistence_module, make_interim, write_zip  # noqa: F401


def test_zip_datasets_and_banks(tmp_path):
    lat, img = make_interim(str(tmp_path))
    ds = formats.LatentCodeDataset(str(tmp_path / 'w.zip'), split='train', w_dim=32, num_ws=6)
    assert len(ds) == 3 * 23                                   # the val member is filtered out by the split
    w, fname = ds[0]
    assert fname == sorted(k for k in lat if 'train' in k)[0] and w.shape == (6, 32) and w.dtype == np.float32
    np.testing.assert_array_equal(ds.lookup(fname), lat[fname])
    with pytest.raises(IOError):
        formats.LatentCodeDataset(str(tmp_path / 'w.zip'), split='train', w_dim=31, num_ws=6)
    with pytest.raises(IOError):
        formats.LatentCodeDataset(str(tmp_path / 'w.zip'), split='test')
    # schedule: ids 00010..00120 step 20 -> 6 slices per patient; step 5 -> 23
    st = formats.compute_stats(ds, 'latent', str(tmp_path / 'cache'), step=20)
    assert st.schedule == ['00010', '00030', '00050', '00070', '00090', '00110']
    W = st.get_all_torch()
    assert W.shape == (3 * 6, 6, 32)
    np.testing.assert_array_equal(W[0].numpy(), lat['train/p000/s_00010.pickle'])
    assert formats.compute_stats(ds, 'latent', str(tmp_path / 'cache'), step=5).get_all_torch().shape[0] == 3 * 23
    # cache file has the reference's tag and reloads identically
    cache = tmp_path / 'cache' / f'latent-step_20-maxitems_{len(ds)}.pkl'
    assert cache.is_file()
    np.testing.assert_array_equal(formats.compute_stats(ds, 'latent', str(tmp_path / 'cache'), step=20).get_all(), W.numpy())
    dimg = formats.ImgDataset(str(tmp_path / 'img.zip'), split='train', modalities=['A', 'B'], resolution=16)
    x, _ = dimg[0]
    assert x.shape == (2, 16, 16)
    X = formats.compute_stats(dimg, 'img', str(tmp_path / 'cache'), step=20).get_all_torch()
    assert X.shape == (18, 2, 16, 16) and float(X.min()) >= -1.0 and float(X.max()) <= 1.0
    np.testing.assert_allclose(X[0, 0].numpy(), img['train/p000/s_00010.pickle']['A'] / 127.5 - 1, rtol=1e-6)
    with pytest.raises(IOError):
        formats.ImgDataset(str(tmp_path / 'img.zip'), split='train', modalities=['A', 'B'], resolution=32)


def test_network_pickle_loader_runs_no_embedded_code(tmp_path, fake_persistence_module):
    G = nets.make_generator(img_resolution=16, img_channels=2, channel_base=256, channel_max=16, w_dim=32, mapping_layers=2,
                            noise_strength=0.1)
    D = nets.make_discriminator(img_resolution=16, img_channels=2, channel_base=256, channel_max=16)
    blob = pickle.dumps(dict(G=_to_persistent(G), D=_to_persistent(D), G_ema=_to_persistent(G), training_set_kwargs=dict(a=1)))
    path = tmp_path / 'network-snapshot-000001.pkl'
    path.write_bytes(blob)
    data = formats.load_network_pkl(str(path))
    Ge = data['G_ema']
    assert Ge.class_name == 'Generator' and Ge.z_dim == 32 and Ge.w_dim == 32 and Ge.num_ws == G.num_ws
    assert Ge.img_resolution == 16 and Ge.img_channels == 2
    want = G.state_dict()
    assert set(want) == set(Ge.keys())
    for k, v in want.items():
        assert torch.equal(v, Ge[k]), k
    assert set(D.state_dict()) == set(data['D'].keys())
    assert data['training_set_kwargs'] == dict(a=1)
    # a pickle that reaches for anything off the allow-list is refused
    evil = pickle.dumps(os.system)
    with pytest.raises(pickle.UnpicklingError):
        formats.load_network_pkl(io.BytesIO(evil))
    # the path rule of load_stylegan
    d = tmp_path / 'models' / 'DS' / 'training-runs' / 'DSNAME' / 'A,B' / '00003-stylegan2-x'
    d.mkdir(parents=True)
    (d / 'net.pkl').write_bytes(blob)
    assert formats.find_network_pkl(str(tmp_path / 'models'), 'DS', 'DSNAME', ['A', 'B'], '00003', 'net.pkl') == str(d / 'net.pkl')


class _Boom:
    """Pickles to a call of os.getcwd -- harmless, but proves code execution if it ever runs inside the loader."""
    fired = False

    def __reduce__(self):
        return (_boom_fire, ())


def _boom_fire():
    _Boom.fired = True
    return 'fired'


class _ViaLoadFromBytes:
    def __init__(self, inner):
        self.inner = inner

    def __reduce__(self):
        return (torch.storage._load_from_bytes, (self.inner,))


def test_safe_loader_refuses_nested_pickle_through_load_from_bytes():
    """torch.storage._load_from_bytes is a full torch.load(weights_only=False): a pickle that wraps a second pickle in it
    must not get an unrestricted unpickle of the inner bytes (round-1 advisor finding), while real tensors carried that
    way still load."""
    inner = pickle.dumps(_Boom())
    evil = pickle.dumps(_ViaLoadFromBytes(inner))
    _Boom.fired = False
    with pytest.raises(Exception) as ei:
        formats._restricted_load(io.BytesIO(evil))
    assert not _Boom.fired, 'inner payload was executed'
    assert isinstance(ei.value, (pickle.UnpicklingError, RuntimeError))
    # a genuine tensor through the same entry point
    buf = io.BytesIO()
    torch.save(torch.arange(5.0), buf)
    good = pickle.dumps(_ViaLoadFromBytes(buf.getvalue()))
    assert torch.equal(formats._restricted_load(io.BytesIO(good)), torch.arange(5.0))
    # numpy: only the reconstruction helpers are reachable, not arbitrary attributes of numpy's core modules
    class _NpAttr:
        def __reduce__(self):
            return (np.zeros, ((2,),))
    with pytest.raises(pickle.UnpicklingError):
        formats._restricted_load(io.BytesIO(pickle.dumps(_NpAttr())))
    # ndarrays and the zip-member dicts still round-trip
    a = {'A': np.arange(6, dtype=np.float32).reshape(2, 3), 'n': np.float64(3.5)}
    b = formats._restricted_load(io.BytesIO(pickle.dumps(a)))
    np.testing.assert_array_equal(a['A'], b['A'])
    assert float(b['n']) == 3.5
