"""GPU parity tests, op level: HIP kernels (through the C ABI) vs the oracle and the reference-generated goldens.

Tolerances: the HIP path computes in fp32 (fp32 MFMA = exact fp32 FMA chains) but sums in a different order from
torch-CPU, so comparisons are rtol 1e-4 / atol scaled to the output magnitude (stated per test).
"""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sg2_ops as ref_ops  # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a ROCm device')
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'l0_ops.npz'))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_library_loaded_is_in_tree():
    from latentaugment_amd import _lib
    lib = _lib.load()
    assert lib.la_abi_version() == 1
    assert os.path.dirname(_lib.LIB_PATH).endswith('latentaugment_amd')
    # the path that was actually dlopen'ed is the in-tree PRODUCT build (the development build, liblatentaug_hip_dev.so, is only ever
    # loaded by scripts/ and by the child processes of one test: the parity suite itself must not run on it)
    assert _lib.LOADED_PATH == os.path.realpath(_lib.LIB_PATH), _lib.LOADED_PATH
    with open('/proc/self/maps') as f:
        mapped = {line.split()[-1] for line in f if 'liblatentaug_hip' in line}
    assert mapped == {os.path.realpath(_lib.LIB_PATH)}, mapped


def test_bias_act_golden(dev, g):
    from latentaugment_amd import ops
    for k in range(int(g['G2_count'])):
        act, gain, clamp = [str(s) for s in g[f'G2_{k}_meta']]
        gain = None if gain == 'None' else float(gain)
        clamp = None if clamp == 'None' else float(clamp)
        x = torch.tensor(g[f'G2_{k}_x'], device=dev, requires_grad=True)
        b = torch.tensor(g[f'G2_{k}_b'], device=dev, requires_grad=True)
        y = ops.bias_act(x, b, act=act, gain=gain, clamp=clamp)
        gx, gb = torch.autograd.grad(y, [x, b], torch.tensor(g[f'G2_{k}_dy'], device=dev))
        close(y, g[f'G2_{k}_y'])
        close(gx, g[f'G2_{k}_gx'])
        close(gb, g[f'G2_{k}_gb'], rtol=1e-4, atol=1e-4)
    y = ops.bias_act(torch.tensor(g['G2_fc_x'], device=dev), torch.tensor(g['G2_fc_b'], device=dev), act='lrelu')
    close(y, g['G2_fc_y'])


def test_bias_act_ragged_and_empty(dev):
    from latentaugment_amd import ops
    x = torch.randn([3, 5, 7, 3], device=dev)          # H*W = 21: not a multiple of 4
    b = torch.randn([5], device=dev)
    close(ops.bias_act(x, b, act='lrelu', clamp=0.5), ref_ops.bias_act(x.cpu(), b.cpu(), act='lrelu', clamp=0.5))
    e = torch.empty([0, 5, 4, 4], device=dev)
    assert ops.bias_act(e, b).shape == (0, 5, 4, 4)
    with pytest.raises(KeyError):
        ops.bias_act(x, b, act='gelu')      # (not in the reference's activation table: it indexes a dict the same way)


def test_bias_act_all_activations_first_and_second_order(dev):
    """Every activation of the reference's table (bias_act.py:20-30) with default and explicit gain / clamp / alpha: forward, the gradient
    of the op (plugin grad = 1) and the gradient of that gradient (grad = 2) against vectors made by RUNNING the reference's
    `bias_act(..., impl='ref')` + autograd in float64 on float32 inputs (tests/golden/make_golden_bias_act.py).  Tolerance: float32
    evaluation of exp / tanh / log1p (1e-5 relative to the tensor's scale); swish forms its derivatives from the saved input."""
    import ast
    from latentaugment_amd import ops
    gb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bias_act_full.npz'))
    seen = set()
    for rep in gb['cases']:
        name, act, gain, clamp, alpha, def_alpha, def_gain, idx, ref, has2 = ast.literal_eval(str(rep))
        seen.add(act)
        assert ops._ACTS[act][0] == idx and ops._ACTS[act][3] == ref and ops._ACTS[act][4] == has2      # the table mirrors the reference's
        assert abs(ops._ACTS[act][1] - def_alpha) < 1e-12 and abs(ops._ACTS[act][2] - def_gain) < 1e-12
        kw = dict(act=act, gain=None if gain < 0 else gain, clamp=None if clamp < 0 else clamp, alpha=None if alpha < 0 else alpha)
        x = torch.tensor(gb[f'{name}_x'], device=dev, requires_grad=True)
        b = torch.tensor(gb[f'{name}_b'], device=dev, requires_grad=True)
        dy = torch.tensor(gb[f'{name}_dy'], device=dev)
        ddx = torch.tensor(gb[f'{name}_ddx'], device=dev)
        y = ops.bias_act(x, b, dim=1, **kw)
        (dx,) = torch.autograd.grad(y, [x], dy, create_graph=True)
        sc = max(1.0, float(np.abs(gb[f'{name}_y']).max()))
        close(y, gb[f'{name}_y'], rtol=1e-5, atol=1e-5 * sc)
        close(dx, gb[f'{name}_dx'], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(gb[f'{name}_dx']).max())))
        if has2:
            (d2,) = torch.autograd.grad(dx, [x], ddx)
            close(d2, gb[f'{name}_d2'], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(gb[f'{name}_d2']).max())))
        else:
            assert float(np.abs(gb[f'{name}_d2']).max()) == 0.0
        # db of the first-order gradient = dx summed over the other axes
        (gb1,) = torch.autograd.grad(ops.bias_act(x, b, dim=1, **kw), [b], dy)
        close(gb1, gb[f'{name}_dx'].sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-4)
    assert seen == set(ops._ACTS)


def test_upfirdn2d_separable_filters_vs_reference(dev):
    """1-D (separable) filters -- what setup_filter keeps for >= 8 taps (upfirdn2d.py:101-104) -- through upfirdn2d / upsample2d /
    downsample2d / filter2d: forward and input gradient against vectors made by RUNNING the reference's impl='ref' in float64
    (tests/golden/make_golden_upfirdn_sep.py).  The op runs one pass per axis, as the reference's plugin path does (:188-201)."""
    from latentaugment_amd import ops
    gs = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'upfirdn_sep.npz'))
    for rep in gs['cases']:
        name, tname, op, kwrep = ast.literal_eval(str(rep))
        kw = ast.literal_eval(kwrep)
        f = ops.setup_filter(list(gs[f'{name}_taps']))
        assert f.ndim == 1 and f.numel() >= 8
        x = torch.tensor(gs[f'{name}_x'], device=dev, requires_grad=True)
        y = getattr(ops, op)(x, f, **kw)
        (dx,) = torch.autograd.grad(y, [x], torch.tensor(gs[f'{name}_dy'], device=dev))
        close(y, gs[f'{name}_y'], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(gs[f'{name}_y']).max())))
        close(dx, gs[f'{name}_dx'], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(gs[f'{name}_dx']).max())))


def test_upfirdn2d_and_bias_act_take_non_contiguous_inputs(dev):
    """channels_last and sliced views give the same results as their contiguous copies (the reference's plugin reads any strides,
    upfirdn2d.cpp:25-30; this binding copies once)."""
    from latentaugment_amd import ops
    f = ops.setup_filter([1, 3, 3, 1])
    x = torch.randn([2, 6, 12, 10], device=dev)
    for v in (x.to(memory_format=torch.channels_last), x.transpose(2, 3), x[:, ::2]):
        close(ops.upsample2d(v, f), ops.upsample2d(v.contiguous(), f), rtol=0, atol=0)
        b = torch.randn([v.shape[1]], device=dev)
        close(ops.bias_act(v, b, act='swish'), ops.bias_act(v.contiguous(), b, act='swish'), rtol=0, atol=0)


def test_upfirdn2d_golden(dev, g):
    from latentaugment_amd import ops
    f = ops.setup_filter([1, 3, 3, 1])
    close(f, g['G1_filter_1331'])
    for k in range(int(g['G3_count'])):
        name, kw = [str(s) for s in g[f'G3_{k}_meta']]
        kw = ast.literal_eval(kw)
        x = torch.tensor(g[f'G3_{k}_x'], device=dev, requires_grad=True)
        y = getattr(ops, name)(x, f, **kw)
        assert tuple(y.shape) == g[f'G3_{k}_y'].shape, (name, kw)
        (gx,) = torch.autograd.grad(y, [x], torch.tensor(g[f'G3_{k}_dy'], device=dev))
        close(y, g[f'G3_{k}_y'])
        close(gx, g[f'G3_{k}_gx'])


def test_upfirdn2d_large_roundtrip_property(dev):
    """<upfirdn(x), y> == <x, upfirdn^T(y)> at a BASELINE-size plane (adjoint identity, size independent)."""
    from latentaugment_amd import ops
    f = ops.setup_filter([1, 3, 3, 1])
    x = torch.randn([2, 3, 256, 256], device=dev, requires_grad=True)
    y = ops.upsample2d(x, f)
    assert y.shape == (2, 3, 512, 512)
    r = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, [x], r)
    lhs = float((y.double() * r.double()).sum())
    rhs = float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_pairwise_l2_golden(dev, golden_dir):
    from latentaugment_amd import ops
    gc = np.load(os.path.join(golden_dir, 'criteria.npz'))
    for tag in ('2d', '3d', '4d'):
        X, Y = torch.tensor(gc[f'G5_{tag}_X'], device=dev), torch.tensor(gc[f'G5_{tag}_Y'], device=dev)
        close(ops.l2_loss_vectorized(X, Y), gc[f'G5_{tag}_mean'], rtol=1e-5, atol=1e-6)
        close(ops.l2_loss_vectorized(X, Y, compute_mean=False), gc[f'G5_{tag}_full'], rtol=1e-5, atol=1e-3)


def _modconv_case(dev, B, cin, cout, res, up, noise_strength, seed, splitk=True, prec=0, tol=1.0):
    """Run the HIP modconv fwd + bwd for one layer and compare with autograd through the oracle."""
    import ctypes as C
    from latentaugment_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(seed)
    rin = res // 2 if up else res
    x = torch.randn([B, cin, rin, rin], generator=gen)
    w = torch.randn([cout, cin, 3, 3], generator=gen)
    s = torch.randn([B, cin], generator=gen) * 0.5 + 1.0
    bias = torch.randn([cout], generator=gen) * 0.1
    noise = torch.randn([res, res], generator=gen)
    gy = torch.randn([B, cout, res, res], generator=gen)
    f = ref_ops.setup_filter([1, 3, 3, 1])
    clamp = 256.0
    # ---- oracle
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    yr = ref_ops.modulated_conv2d(xr, w, sr, noise=noise * noise_strength, up=2 if up else 1, padding=1,
                                  resample_filter=f, flip_weight=not up, fused_modconv=True)
    yr = ref_ops.bias_act(yr, bias, act='lrelu', clamp=clamp)
    gxr, gsr = torch.autograd.grad(yr, [xr, sr], gy)
    # ---- HIP
    st = _lib.stream_ptr()
    xd, wd_, sd, bd, nd, gyd = [t.to(dev).contiguous() for t in (x, w, s, bias, noise, gy)]
    wf = torch.empty([9, cin, cout], device=dev)
    wb = torch.empty([9, cout, cin], device=dev)
    wsq = torch.empty([cout, cin], device=dev)
    _lib.check(lib.la_pack_conv_weights_f32(_lib.ptr(wd_), _lib.ptr(wf), _lib.ptr(wb), _lib.ptr(wsq), cout, cin, 9, st))
    d = torch.rsqrt((sd.square() @ wsq.t()) + 1e-8).contiguous()     # test-side demod (the engine has its own kernel)
    wqf = wqb = None
    if prec:
        wqf = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 0, 3)], dtype=torch.uint8, device=dev)
        wqb = torch.empty([lib.la_modconv_bf16_pack_bytes(cin, cout, 1, 3)], dtype=torch.uint8, device=dev)
        _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(wd_), _lib.ptr(wqf), cout, cin, 9, 0, 3, st))
        _lib.check(lib.la_pack_conv_weights_bf16_f32(_lib.ptr(wd_), _lib.ptr(wqb), cout, cin, 9, 1, 3, st))
    y = torch.empty([B, cout, res, res], device=dev)
    fir = np.ascontiguousarray(f.numpy())
    scratch = torch.empty([B * cout * (res + 1) * (res + 1)], device=dev)
    sq2 = float(np.sqrt(2))
    skn = int(lib.la_modconv_workspace_bytes(B, cin, cout, res, 1 if up else 0))
    if not splitk:      # only room for the pre-split input: forces the direct (non split-K) kernels
        rin_ = res // 2 if up else res
        pad32 = lambda c: (c + 31) // 32 * 32      # the pre-split copy is channel-interleaved in whole 32-channel chunks
        skn = (8 * B * max(pad32(cin) * rin_ * rin_, pad32(cout) * (res + 1) * (res + 1)) + 16 + 1024 + 32 * B * max(cin, cout)
               + 4 * B * max(cin, cout) * 64 + 1024 + 255) // 256 * 256 if prec else 0
    skw = torch.empty([max(skn, 1)], dtype=torch.uint8, device=dev) if skn else None
    if up:
        _lib.check(lib.la_modconv3x3_up2_fwd_f32(_lib.ptr(xd), cin * rin * rin, _lib.ptr(wf), _lib.ptr(wqf), prec, _lib.ptr(sd), cin, _lib.ptr(d),
                                                 cout, _lib.ptr(nd), 0, noise_strength, _lib.ptr(bd), 3, 0.2, sq2, clamp,
                                                 fir.ctypes.data, _lib.ptr(scratch), _lib.ptr(y), _lib.ptr(skw), skn, B, cin, cout, res, st))
    else:
        _lib.check(lib.la_modconv3x3_fwd_f32(_lib.ptr(xd), cin * rin * rin, _lib.ptr(wf), _lib.ptr(wqf), prec, _lib.ptr(sd), cin, _lib.ptr(d), cout,
                                             _lib.ptr(nd), 0, noise_strength, _lib.ptr(bd), 3, 0.2, sq2, clamp, _lib.ptr(y),
                                             _lib.ptr(skw), skn, B, cin, cout, res, st))
    scale = float(yr.abs().max())
    close(y, yr, rtol=1e-4 * tol, atol=1e-5 * scale * tol)
    # backward: act' and demod applied on the test side (the engine's seam kernel does this), then the HIP contraction
    # (slope taken from the oracle's output so that a sign flip of a ~0 activation -- legitimate at any finite precision --
    #  does not turn the backward comparison into a discontinuity test)
    y = yr.detach().to(dev)
    slope = torch.where(y > 0, sq2, 0.2 * sq2) * (y.abs() < clamp)
    g1 = gyd * slope
    gz = (g1 * d[:, :, None, None]).contiguous()
    tiles = lib.la_modconv_ds_tiles(rin)
    gx = torch.empty([B, cin, rin, rin], device=dev)
    dsp = torch.zeros([B, cin, tiles], device=dev)
    if up:
        _lib.check(lib.la_modconv3x3_up2_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), prec, _lib.ptr(sd), cin, _lib.ptr(xd), cin * rin * rin,
                                                 fir.ctypes.data, _lib.ptr(scratch), _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin,
                                                 cout, res, st))
    else:
        _lib.check(lib.la_modconv3x3_bwd_f32(_lib.ptr(gz), _lib.ptr(wb), _lib.ptr(wqb), prec, _lib.ptr(sd), cin, _lib.ptr(xd), cin * rin * rin,
                                             _lib.ptr(gx), _lib.ptr(dsp), _lib.ptr(skw), skn, B, cin, cout, res, st))
    close(gx, gxr, rtol=1e-4 * tol, atol=1e-5 * float(gxr.abs().max()) * tol)
    # style gradient = modulation term (partials) + demodulation term (test-side, mirrors la_style_backward_conv)
    zd = torch.where(y > 0, y / sq2, y / (0.2 * sq2)) - bd[None, :, None, None] - nd[None, None] * noise_strength
    ddn = (g1 * zd).sum(dim=[2, 3])
    ds = dsp.sum(dim=2) - sd * ((ddn * d * d) @ wsq)
    close(ds, gsr, rtol=2e-4 * tol, atol=2e-5 * float(gsr.abs().max()) * tol)


@pytest.mark.parametrize('case', [
    dict(B=2, cin=16, cout=16, res=8, up=False, noise_strength=0.0),
    dict(B=3, cin=32, cout=64, res=16, up=False, noise_strength=0.3),
    dict(B=2, cin=24, cout=132, res=32, up=False, noise_strength=0.1),     # ragged channels (not multiples of the tile)
    dict(B=1, cin=64, cout=128, res=64, up=False, noise_strength=0.0),
    dict(B=2, cin=16, cout=16, res=8, up=True, noise_strength=0.2),
    dict(B=2, cin=64, cout=32, res=32, up=True, noise_strength=0.0),
    dict(B=1, cin=128, cout=64, res=64, up=True, noise_strength=0.1),
    dict(B=2, cin=512, cout=512, res=4, up=False, noise_strength=0.0),     # the 4x4 block
])
def test_modconv_vs_oracle(dev, case):
    _modconv_case(dev, seed=7, **case)                    # split-K where the layer qualifies (<= 32x32)
    if case['res'] <= 32:
        _modconv_case(dev, seed=7, splitk=False, **case)  # same layer through the direct kernel


@pytest.mark.parametrize('case', [
    dict(B=2, cin=16, cout=16, res=8, up=False, noise_strength=0.0),
    dict(B=2, cin=24, cout=132, res=32, up=False, noise_strength=0.1),     # ragged channels
    dict(B=1, cin=64, cout=128, res=64, up=False, noise_strength=0.0),
    dict(B=2, cin=40, cout=200, res=64, up=False, noise_strength=0.1),     # halo kernel with a ragged channel chunk and M tile
    dict(B=2, cin=32, cout=32, res=64, up=False, noise_strength=0.1),      # 32-row halo tiles (1 x 4 wave grid)
    dict(B=1, cin=24, cout=20, res=64, up=False, noise_strength=0.0),      # ... ragged in both M and C
    dict(B=2, cin=64, cout=32, res=32, up=True, noise_strength=0.0),
    dict(B=1, cin=128, cout=64, res=64, up=True, noise_strength=0.1),
    dict(B=2, cin=48, cout=32, res=128, up=True, noise_strength=0.1),      # 64x64 input: the four phases run as ONE merged launch
    dict(B=1, cin=128, cout=40, res=128, up=True, noise_strength=0.0),     # 128-row backward tile over a 129^2 gradient scratch, ragged channel chunk
    dict(B=2, cin=512, cout=512, res=4, up=False, noise_strength=0.0),
])
def test_modconv_split_bf16_vs_oracle(dev, case):
    """Split-bf16 contraction: x3 (6 MFMAs) must hold the SAME tolerance as the exact fp32 path; x2 (3 MFMAs) 10x looser."""
    _modconv_case(dev, seed=11, prec=1, **case)
    _modconv_case(dev, seed=11, prec=3, **case)              # scaled split-fp16 x2: same tolerance as exact fp32
    _modconv_case(dev, seed=11, prec=2, tol=10.0, **case)
    if case['res'] <= 32:
        _modconv_case(dev, seed=11, prec=1, splitk=False, **case)
