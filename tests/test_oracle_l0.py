"""Oracle L0 ops vs golden vectors produced by the reference's own op layer (tests/golden/l0_ops.npz)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import sg2_ops as ops

TOL = dict(rtol=1e-5, atol=1e-6)


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'l0_ops.npz'))


def t(a, grad=False):
    return torch.from_numpy(np.asarray(a)).clone().requires_grad_(grad)


def test_setup_filter(g):
    np.testing.assert_allclose(ops.setup_filter([1, 3, 3, 1]).numpy(), g['G1_filter_1331'], **TOL)
    np.testing.assert_allclose(ops.setup_filter([1, 3, 3, 1], gain=4).numpy(), g['G1_filter_1331_gain4'], **TOL)
    np.testing.assert_allclose(ops.setup_filter([1, 2, 1], flip_filter=True).numpy(), g['G1_filter_121_flip'], **TOL)
    f = ops.setup_filter([1, 3, 3, 1])
    assert f.shape == (4, 4) and abs(float(f.sum()) - 1) < 1e-6


def test_bias_act(g):
    n = int(g['G2_count'])
    assert n == 18
    for k in range(n):
        act, gain, clamp = [str(s) for s in g[f'G2_{k}_meta']]
        gain = None if gain == 'None' else float(gain)
        clamp = None if clamp == 'None' else float(clamp)
        x, b = t(g[f'G2_{k}_x'], True), t(g[f'G2_{k}_b'], True)
        y = ops.bias_act(x, b, act=act, gain=gain, clamp=clamp)
        gx, gb = torch.autograd.grad(y, [x, b], t(g[f'G2_{k}_dy']))
        np.testing.assert_allclose(y.detach().numpy(), g[f'G2_{k}_y'], **TOL)
        np.testing.assert_allclose(gx.numpy(), g[f'G2_{k}_gx'], **TOL)
        np.testing.assert_allclose(gb.numpy(), g[f'G2_{k}_gb'], rtol=1e-4, atol=1e-5)
    y = ops.bias_act(t(g['G2_fc_x']), t(g['G2_fc_b']), act='lrelu')
    np.testing.assert_allclose(y.numpy(), g['G2_fc_y'], **TOL)


def test_upfirdn2d_family(g):
    f = ops.setup_filter([1, 3, 3, 1])
    n = int(g['G3_count'])
    assert n == 24
    for k in range(n):
        name, kw = [str(s) for s in g[f'G3_{k}_meta']]
        kw = ast.literal_eval(kw)
        x = t(g[f'G3_{k}_x'], True)
        y = getattr(ops, name)(x, f, **kw)
        assert tuple(y.shape) == g[f'G3_{k}_y'].shape, (name, kw)
        (gx,) = torch.autograd.grad(y, [x], t(g[f'G3_{k}_dy']))
        np.testing.assert_allclose(y.detach().numpy(), g[f'G3_{k}_y'], **TOL)
        np.testing.assert_allclose(gx.numpy(), g[f'G3_{k}_gx'], **TOL)


def test_conv2d_resample(g):
    f = ops.setup_filter([1, 3, 3, 1])
    n = int(g['G4_count'])
    assert n == 10
    for k in range(n):
        groups, kw = [str(s) for s in g[f'G4_{k}_meta']]
        kw = ast.literal_eval(kw)
        x, w = t(g[f'G4_{k}_x'], True), t(g[f'G4_{k}_w'], True)
        y = ops.conv2d_resample(x, w, f=f, groups=int(groups), **kw)
        gx, gw = torch.autograd.grad(y, [x, w], t(g[f'G4_{k}_dy']))
        np.testing.assert_allclose(y.detach().numpy(), g[f'G4_{k}_y'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gx.numpy(), g[f'G4_{k}_gx'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gw.numpy(), g[f'G4_{k}_gw'], rtol=1e-4, atol=1e-4)


def test_fma(g):
    a, b, c = t(g['G8_a'], True), t(g['G8_b'], True), t(g['G8_c'], True)
    y = ops.fma(a, b, c)
    ga, gb, gc = torch.autograd.grad(y, [a, b, c], t(g['G8_dy']))
    np.testing.assert_allclose(y.detach().numpy(), g['G8_y'], **TOL)
    np.testing.assert_allclose(ga.numpy(), g['G8_ga'], **TOL)
    np.testing.assert_allclose(gb.numpy(), g['G8_gb'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gc.numpy(), g['G8_gc'], rtol=1e-4, atol=1e-5)


def test_modconv_fused_equals_unfused():
    gen = torch.Generator().manual_seed(3)
    f = ops.setup_filter([1, 3, 3, 1])
    for up in (1, 2):
        x = torch.randn([3, 6, 8, 8], generator=gen)
        w = torch.randn([5, 6, 3, 3], generator=gen)
        s = torch.randn([3, 6], generator=gen) + 1
        nz = torch.randn([8 * up, 8 * up], generator=gen) * 0.1
        a = ops.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=f, flip_weight=(up == 1),
                                 fused_modconv=True)
        b = ops.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=f, flip_weight=(up == 1),
                                 fused_modconv=False)
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-5)


def test_philox4x32_10_known_answers():
    """oracle/noise_ref.py (the checker of la_noise_normal_f32) against the known-answer vectors of the Random123 distribution
    (kat_vectors: philox4x32 10 rounds): zero counter / key, all ones, and the digits-of-pi counter."""
    import numpy as np
    from oracle import noise_ref
    kat = [((0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, '408f276d 41c83b0e a20bc7c6 6d5451fd'),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), 'd16cfe09 94fdcceb 5001e420 24126ea1')]
    for c, k, want in kat:
        got = noise_ref.philox4x32_10(*[np.uint32(v) for v in c], *k)
        assert ' '.join(f'{int(v):08x}' for v in got) == want
    z = noise_ref.noise_normal(16, 16384, seed=99, layer=2)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.var()) - 1.0) < 0.01
