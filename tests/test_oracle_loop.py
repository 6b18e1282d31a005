"""Oracle criteria / crops / loop vs goldens produced by the reference's LatentAug (tests/golden)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import feature_net, latent_aug_ref as lar, sg2_networks as nets


@pytest.fixture(scope='module')
def gc(golden_dir):
    return np.load(os.path.join(golden_dir, 'criteria.npz'))


@pytest.fixture(scope='module')
def gl(golden_dir):
    return np.load(os.path.join(golden_dir, 'latent_loop.npz'))


def t(a, grad=False):
    return torch.from_numpy(np.asarray(a)).clone().requires_grad_(grad)


def test_l2_loss_vectorized(gc):
    for tag in ('2d', '3d', '4d'):
        X, Y = t(gc[f'G5_{tag}_X'], True), t(gc[f'G5_{tag}_Y'])
        d = lar.l2_loss_vectorized(X, Y)
        (gx,) = torch.autograd.grad(d, [X])
        np.testing.assert_allclose(d.detach().numpy(), gc[f'G5_{tag}_mean'], rtol=1e-5)
        np.testing.assert_allclose(lar.l2_loss_vectorized(X, Y, compute_mean=False).detach().numpy(),
                                   gc[f'G5_{tag}_full'], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(gx.numpy(), gc[f'G5_{tag}_gx'], rtol=1e-4, atol=1e-7)


def test_center_crop(gc):
    for res, want in ((256, 181), (512, 362), (1024, 724), (32, 22)):
        assert lar.center_crop_size(res) == want
        img = torch.arange(res * res, dtype=torch.float32).reshape(1, 1, res, res)
        cc = lar.center_crop(img, want)
        assert tuple(cc.shape) == tuple(gc[f'G6_cc_{res}_shape'])
        assert float(cc[0, 0, 0, 0]) == float(gc[f'G6_cc_{res}_first'])
        assert float(cc[0, 0, -1, -1]) == float(gc[f'G6_cc_{res}_last'])


def test_crop_params_and_transform(gc):
    for seed in (0, 6, 99):
        for res, crop in ((256, 64), (32, 8)):
            random.seed(seed)
            pos = lar.get_crop_params(res, crop)
            assert tuple(pos) == tuple(gc[f'G6_params_{seed}_{res}_{crop}'])
            img = torch.arange(res * res, dtype=torch.float32).reshape(1, 1, res, res)
            c = lar.apply_aug_transform(img, res, crop, 'center_random_crop', pos)
            assert tuple(c.shape) == tuple(gc[f'G6_crop_{seed}_{res}_{crop}_shape'])
            assert float(c[0, 0, 0, 0]) == float(gc[f'G6_crop_{seed}_{res}_{crop}_first'])


def test_adam_matches_torch():
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn([4, 1, 16], generator=g)
    p_t = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_t], betas=(0.9, 0.999), lr=0.01)
    st = lar.AdamState(p0, 0.01)
    p = p0.clone()
    for _ in range(7):
        grad = torch.randn([4, 1, 16], generator=g)
        opt.zero_grad()
        p_t.grad = grad.clone()
        opt.step()
        p = st.step(p, grad)
        np.testing.assert_allclose(p.numpy(), p_t.detach().numpy(), rtol=1e-6, atol=1e-7)


CASES = {
    'latent': dict(w_latent=0.5),
    'pix': dict(w_pix=2.0),
    'disc': dict(w_disc=1.0),
    'lpips': dict(w_lpips=3.0),
    'all': dict(w_latent=0.3, w_pix=1.0, w_disc=0.5, w_lpips=2.0),
    'soft': dict(w_latent=0.3, w_pix=1.0, soft_aug=True, alpha=0.7),
}


def _nets(gl):
    res, cbase, cmax, wdim = int(gl['res']), int(gl['cbase']), int(gl['cmax']), int(gl['wdim'])
    G = nets.make_generator(img_resolution=res, img_channels=2, channel_base=cbase, channel_max=cmax, seed=0,
                            noise_strength=0.1, w_dim=wdim, mapping_layers=2)
    D = nets.make_discriminator(img_resolution=res, img_channels=2, channel_base=cbase, channel_max=cmax, seed=0)
    return G, D


@pytest.mark.parametrize('name', list(CASES))
def test_loop_matches_reference(gl, name):
    G, D = _nets(gl)
    fnet = feature_net.TinyFeatureNet(seed=5)
    kw = dict(CASES[name])
    ref = lar.LatentAugRef(G, D, W=t(gl['W']), X=t(gl['X']), fea=[t(gl['fea0']), t(gl['fea1'])], feature_net=fnet,
                           res=int(gl['res']), num_epochs=int(gl['epochs']), opt_lr=float(gl['lr']),
                           crop_size=int(gl['crop']), **kw)
    random.seed(6)
    pos = lar.get_crop_params(int(gl['res']), int(gl['crop']))
    assert tuple(pos) == tuple(gl[f'{name}_crop_pos'])
    torch.manual_seed(123)
    img, w_aug = ref.forward(t(gl['w0']), crop_pos=pos)
    np.testing.assert_allclose(w_aug.numpy(), gl[f'{name}_w_aug'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(img.numpy(), gl[f'{name}_img'], rtol=1e-3, atol=2e-4)


def test_loop_unfused_modconv_equals_fused(gl):
    G, D = _nets(gl)
    kw = dict(w_latent=0.3, w_pix=1.0)
    outs = []
    for fused in (True, False):
        ref = lar.LatentAugRef(G, D, W=t(gl['W']), X=t(gl['X']), res=int(gl['res']), num_epochs=3, opt_lr=0.01,
                               crop_size=8, final_noise_mode='const', fused_modconv=fused, **kw)
        outs.append(ref.forward(t(gl['w0']), crop_pos=(0, 0)))
    np.testing.assert_allclose(outs[0][1].numpy(), outs[1][1].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(outs[0][0].numpy(), outs[1][0].numpy(), rtol=1e-3, atol=2e-4)


def test_ganrand(gl):
    G, _ = _nets(gl)
    with torch.no_grad():
        G.mapping.w_avg.copy_(t(gl['w_avg']))
    torch.manual_seed(321)
    ws = G.mapping(t(gl['ganrand_z']), None, truncation_psi=0.7)
    img = G.synthesis(ws)
    np.testing.assert_allclose(ws.numpy(), gl['ganrand_ws'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(img.numpy(), gl['ganrand_img'], rtol=1e-3, atol=2e-4)


def test_state_dict_names_match_reference_legacy_map():
    """Names at models/stylegan3/legacy.py:171-203 (G) and :271-288 (D)."""
    G = nets.make_generator(img_resolution=16, img_channels=2, channel_base=128, channel_max=8, w_dim=16,
                            mapping_layers=8)
    keys = set(G.state_dict().keys())
    for k in ('mapping.w_avg', 'mapping.fc0.weight', 'mapping.fc7.bias', 'synthesis.b4.const',
              'synthesis.b4.conv1.weight', 'synthesis.b4.conv1.bias', 'synthesis.b4.conv1.noise_const',
              'synthesis.b4.conv1.noise_strength', 'synthesis.b4.conv1.affine.weight',
              'synthesis.b4.conv1.affine.bias', 'synthesis.b8.conv0.weight', 'synthesis.b8.conv0.affine.bias',
              'synthesis.b16.conv1.noise_const', 'synthesis.b16.torgb.weight', 'synthesis.b16.torgb.affine.weight',
              'synthesis.b8.conv0.resample_filter'):
        assert k in keys, k
    assert G.synthesis.b8.conv0.weight.shape == (8, 8, 3, 3)
    assert float(G.synthesis.b8.conv0.affine.bias[0]) == 1.0
    D = nets.make_discriminator(img_resolution=16, img_channels=2, channel_base=128, channel_max=8)
    dk = set(D.state_dict().keys())
    for k in ('b16.fromrgb.weight', 'b16.fromrgb.bias', 'b16.conv0.weight', 'b16.conv1.bias', 'b16.skip.weight',
              'b8.conv0.weight', 'b4.conv.weight', 'b4.fc.weight', 'b4.fc.bias', 'b4.out.weight', 'b4.out.bias'):
        assert k in dk, k
    assert 'b16.skip.bias' not in dk
    for res, nws in ((256, 14), (512, 16), (1024, 18)):
        assert 2 * int(np.log2(res)) - 2 == nws
