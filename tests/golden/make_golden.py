#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only).  Never runs on the GPU box; only the .npz outputs travel.

What is executed from the reference (imported, never copied):
  * models/stylegan3/torch_utils/ops/{bias_act,upfirdn2d,conv2d_resample,fma}.py  (CPU `_ref` branches)
  * augments/utils/util_latent_aug.py::LatentAug.forward / l2_loss_vectorized / calc_loss_*
  * augments/utils/util_dataset.py::get_params / get_transform / get_center_crop / crop
Modules the reference imports but that are absent here and unused on this path (cv2, openpyxl) are
satisfied by empty module objects; `torchvision.transforms` (absent) by the four tiny callables the
path uses (Compose, Lambda, CenterCrop with torchvision's published offset formula, RandomCrop).
The SG2 network classes are NOT in the reference (SURVEY "three facts" #1): the build's own
oracle networks are injected, re-wired onto the REFERENCE op layer, so the goldens pin
"reference ops + reference loop" around our L1 definition.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'models', 'stylegan3'))

torch.set_num_threads(4)
torch.backends.mkldnn.enabled = True


# ------------------------------------------------------------------ absent third-party modules
def _install_absent_modules():
    for name in ('cv2', 'openpyxl'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    try:
        import torchvision.transforms  # noqa: F401
    except ImportError:
        tv = types.ModuleType('torchvision')
        tr = types.ModuleType('torchvision.transforms')

        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        class Lambda:
            def __init__(self, fn):
                self.fn = fn

            def __call__(self, x):
                return self.fn(x)

        class CenterCrop:
            def __init__(self, size):
                self.size = int(size)

            def __call__(self, img):
                h, w = img.shape[-2:]
                top = int(round((h - self.size) / 2.0))
                left = int(round((w - self.size) / 2.0))
                return img[..., top:top + self.size, left:left + self.size]

        class RandomCrop:
            def __init__(self, size):
                self.size = int(size)

            def __call__(self, img):
                h, w = img.shape[-2:]
                i = random.randint(0, h - self.size)
                j = random.randint(0, w - self.size)
                return img[..., i:i + self.size, j:j + self.size]

        tr.Compose, tr.Lambda, tr.CenterCrop, tr.RandomCrop = Compose, Lambda, CenterCrop, RandomCrop
        tv.transforms = tr
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.transforms'] = tr


_install_absent_modules()

from torch_utils.ops import bias_act as ref_bias_act            # noqa: E402
from torch_utils.ops import upfirdn2d as ref_upfirdn2d          # noqa: E402
from torch_utils.ops import conv2d_resample as ref_conv2d_resample  # noqa: E402
from torch_utils.ops import fma as ref_fma                      # noqa: E402
from augments.utils import util_latent_aug as ref_ula           # noqa: E402
from augments.utils import util_dataset as ref_ud               # noqa: E402

from oracle import sg2_ops as our_ops                           # noqa: E402
from oracle import sg2_networks as our_nets                     # noqa: E402
from oracle import feature_net as our_fnet                      # noqa: E402


def T(a):
    return a.detach().cpu().numpy()


# ------------------------------------------------------------------ G1-G4, G8: L0 ops
def gen_l0():
    out = {}
    g = torch.Generator().manual_seed(11)
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    out['G1_filter_1331'] = T(f)
    out['G1_filter_1331_gain4'] = T(ref_upfirdn2d.setup_filter([1, 3, 3, 1], gain=4))
    out['G1_filter_121_flip'] = T(ref_upfirdn2d.setup_filter([1, 2, 1], flip_filter=True))

    # G2 bias_act
    k = 0
    for act in ('linear', 'lrelu', 'relu'):
        for gain in (None, 1.0, float(np.sqrt(2)) * 0.5):
            for clamp in (None, 0.7):
                x = torch.randn([2, 8, 16, 16], generator=g).requires_grad_(True)
                b = torch.randn([8], generator=g).requires_grad_(True)
                dy = torch.randn([2, 8, 16, 16], generator=g)
                y = ref_bias_act.bias_act(x, b, act=act, gain=gain, clamp=clamp)
                gx, gb = torch.autograd.grad(y, [x, b], dy)
                out[f'G2_{k}_meta'] = np.array([act, str(gain), str(clamp)])
                for n, v in (('x', x), ('b', b), ('dy', dy), ('y', y), ('gx', gx), ('gb', gb)):
                    out[f'G2_{k}_{n}'] = T(v)
                k += 1
    out['G2_count'] = np.array(k)
    # FC-style use: dim=1 on a 2-D tensor
    x = torch.randn([3, 8], generator=g)
    b = torch.randn([8], generator=g)
    out['G2_fc_x'], out['G2_fc_b'] = T(x), T(b)
    out['G2_fc_y'] = T(ref_bias_act.bias_act(x, b, act='lrelu'))

    # G3 upfirdn2d family
    k = 0
    cases = [
        dict(fn='upfirdn2d', up=1, down=1, padding=[1, 1, 1, 1], gain=4.0),       # FIR after up-conv
        dict(fn='upsample2d', up=2),                                              # img skip upsample
        dict(fn='downsample2d', down=2),                                          # D skip path
        dict(fn='filter2d'),
        dict(fn='upfirdn2d', up=1, down=1, padding=[2, 2, 2, 2], gain=1.0),       # D conv1 pre-filter
        dict(fn='upfirdn2d', up=2, down=1, padding=[2, 1, 2, 1], gain=4.0, flip_filter=True),
        dict(fn='upfirdn2d', up=1, down=2, padding=[1, 1, 1, 1], gain=1.0),
        dict(fn='upfirdn2d', up=2, down=2, padding=[-1, 2, 0, 1], gain=1.0),      # crop + mixed
    ]
    for shape in ([2, 3, 9, 9], [2, 3, 16, 16], [1, 2, 5, 12]):
        for cs in cases:
            cs = dict(cs)
            fn = getattr(ref_upfirdn2d, cs.pop('fn'))
            x = torch.randn(shape, generator=g).requires_grad_(True)
            y = fn(x, f, **cs)
            dy = torch.randn(y.shape, generator=g)
            (gx,) = torch.autograd.grad(y, [x], dy)
            out[f'G3_{k}_meta'] = np.array([fn.__name__, repr(cs)])
            for n, v in (('x', x), ('dy', dy), ('y', y), ('gx', gx)):
                out[f'G3_{k}_{n}'] = T(v)
            k += 1
    out['G3_count'] = np.array(k)

    # G4 conv2d_resample
    k = 0
    for groups in (1, 2):
        for cs in (dict(up=2, padding=1, flip_weight=False, ksz=3),    # synthesis conv0
                   dict(padding=1, flip_weight=True, ksz=3),           # synthesis conv1 / D conv0
                   dict(down=2, padding=1, flip_weight=True, ksz=3),   # D conv1
                   dict(down=2, padding=0, flip_weight=True, ksz=1),   # D skip
                   dict(padding=0, flip_weight=True, ksz=1)):          # torgb / fromrgb
            cs = dict(cs)
            ksz = cs.pop('ksz')
            cin, cout = 4, 6
            x = torch.randn([1 if groups > 1 else 2, cin * groups, 8, 8], generator=g).requires_grad_(True)
            w = torch.randn([cout * groups, cin, ksz, ksz], generator=g).requires_grad_(True)
            y = ref_conv2d_resample.conv2d_resample(x, w, f=f, groups=groups, **cs)
            dy = torch.randn(y.shape, generator=g)
            gx, gw = torch.autograd.grad(y, [x, w], dy)
            out[f'G4_{k}_meta'] = np.array([str(groups), repr(cs)])
            for n, v in (('x', x), ('w', w), ('dy', dy), ('y', y), ('gx', gx), ('gw', gw)):
                out[f'G4_{k}_{n}'] = T(v)
            k += 1
    out['G4_count'] = np.array(k)

    # G8 fma with broadcasting
    a = torch.randn([2, 4, 5, 5], generator=g).requires_grad_(True)
    b = torch.randn([2, 4, 1, 1], generator=g).requires_grad_(True)
    c = torch.randn([1, 1, 5, 5], generator=g).requires_grad_(True)
    y = ref_fma.fma(a, b, c)
    dy = torch.randn(y.shape, generator=g)
    ga, gb, gc = torch.autograd.grad(y, [a, b, c], dy)
    for n, v in (('a', a), ('b', b), ('c', c), ('dy', dy), ('y', y), ('ga', ga), ('gb', gb), ('gc', gc)):
        out[f'G8_{n}'] = T(v)
    np.savez_compressed(os.path.join(HERE, 'l0_ops.npz'), **out)
    print('l0_ops.npz', len(out), 'arrays')


# ------------------------------------------------------------------ G5, G6: criterion + crops
def gen_criteria():
    out = {}
    g = torch.Generator().manual_seed(12)
    l2 = ref_ula.LatentAug.l2_loss_vectorized
    for tag, xs, ys in (('2d', [3, 40], [7, 40]), ('3d', [2, 6, 32], [9, 6, 32]), ('4d', [2, 1, 13, 13], [5, 1, 13, 13])):
        X = torch.randn(xs, generator=g).requires_grad_(True)
        Y = torch.randn(ys, generator=g)
        d_mean = l2(X, Y)
        d_full = l2(X, Y, compute_mean=False)
        (gx,) = torch.autograd.grad(d_mean, [X])
        out[f'G5_{tag}_X'], out[f'G5_{tag}_Y'] = T(X), T(Y)
        out[f'G5_{tag}_mean'], out[f'G5_{tag}_full'], out[f'G5_{tag}_gx'] = T(d_mean), T(d_full), T(gx)
    # crops
    for res in (256, 512, 1024, 32):
        img = torch.arange(res * res, dtype=torch.float32).reshape(1, 1, res, res)
        cc = ref_ud.get_center_crop(load_size=res)(img)
        out[f'G6_cc_{res}_shape'] = np.array(cc.shape)
        out[f'G6_cc_{res}_first'] = T(cc[0, 0, 0, 0])
        out[f'G6_cc_{res}_last'] = T(cc[0, 0, -1, -1])
    for seed in (0, 6, 99):
        for res, crop in ((256, 64), (32, 8)):
            random.seed(seed)
            p = ref_ud.get_params(load_size=res, crop_size=crop, preprocess='center_random_crop')
            out[f'G6_params_{seed}_{res}_{crop}'] = np.array(p['crop_pos'])
            img = torch.arange(res * res, dtype=torch.float32).reshape(1, 1, res, res)
            tr = ref_ud.get_transform(load_size=res, crop_size=crop, preprocess='center_random_crop', params=p)
            c = tr(img)
            out[f'G6_crop_{seed}_{res}_{crop}_first'] = T(c[0, 0, 0, 0])
            out[f'G6_crop_{seed}_{res}_{crop}_shape'] = np.array(c.shape)
    np.savez_compressed(os.path.join(HERE, 'criteria.npz'), **out)
    print('criteria.npz', len(out), 'arrays')


# ------------------------------------------------------------------ G7: the loop
class _RefOpsAdapter:
    """Same surface as oracle.sg2_ops, but every L0 op is the REFERENCE's."""
    SQRT2 = our_ops.SQRT2
    act_defaults = staticmethod(our_ops.act_defaults)
    setup_filter = staticmethod(lambda taps=(1, 3, 3, 1), **kw: ref_upfirdn2d.setup_filter(list(taps), **kw))
    bias_act = staticmethod(lambda x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None:
                            ref_bias_act.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp))
    upsample2d = staticmethod(lambda x, f, **kw: ref_upfirdn2d.upsample2d(x, f, **kw))
    conv2d_resample = staticmethod(lambda x, w, **kw: ref_conv2d_resample.conv2d_resample(x, w, **kw))
    fma = staticmethod(lambda a, b, c: ref_fma.fma(a, b, c))

    @staticmethod
    def modulated_conv2d(*a, **kw):
        saved = (our_ops.conv2d_resample, our_ops.fma)
        our_ops.conv2d_resample = _RefOpsAdapter.conv2d_resample
        our_ops.fma = _RefOpsAdapter.fma
        try:
            return our_ops.modulated_conv2d(*a, **kw)
        finally:
            our_ops.conv2d_resample, our_ops.fma = saved


class _ScriptLikeVGG(torch.nn.Module):
    """Gives our tiny feature net the call signature of NVIDIA's vgg16.pt (util_latent_aug.py:395)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x, resize_images=False, return_lpips=True):
        return self.net(x)


def build_ref_module(G, D, W, X, fea, fnet, *, res, batch, epochs, lr, w_latent, w_pix, w_disc, w_lpips,
                     crop, soft_aug=False, alpha=1.0):
    m = ref_ula.LatentAug.__new__(ref_ula.LatentAug)
    torch.nn.Module.__init__(m)
    m.G, m.D = G, D
    m.z_dim, m.w_dim, m.num_ws = G.z_dim, G.w_dim, G.num_ws
    m.batch_size, m.world_size, m.res = batch, 1, res
    m.modalities = ['A', 'B']
    m.num_epochs, m.opt_lr = epochs, lr
    m.lpips_script = 'lpips_script'
    m.truncation_psi = 1.0
    m.w_pix, m.w_lpips, m.w_latent, m.w_disc = w_pix, w_lpips, w_latent, w_disc
    m.crop_size, m.preprocess = crop, 'center_random_crop'
    m.soft_aug, m.alpha = soft_aug, alpha
    m.verbose_log = m.verbose_flag = False
    if W is not None:
        m.register_buffer('W', W)
    if X is not None:
        m.register_buffer('X', X)
    if fea is not None:
        for i, mode in enumerate(m.modalities):
            m.register_buffer(f'fea_{mode}', fea[i])
        m.vgg16 = _ScriptLikeVGG(fnet)
    return m


def gen_loop():
    res, cbase, cmax, wdim, B = 32, 256, 16, 32, 2
    # networks: our definition on the reference's ops
    our_nets.ops = _RefOpsAdapter
    try:
        G = our_nets.make_generator(img_resolution=res, img_channels=2, channel_base=cbase, channel_max=cmax,
                                    seed=0, noise_strength=0.1, w_dim=wdim, mapping_layers=2)
        D = our_nets.make_discriminator(img_resolution=res, img_channels=2, channel_base=cbase, channel_max=cmax,
                                        seed=0)
        fnet = our_fnet.TinyFeatureNet(seed=5)
        g = torch.Generator().manual_seed(21)
        w0 = torch.randn([B, 1, wdim], generator=g)
        W = torch.randn([12, 1, wdim], generator=g).repeat(1, G.num_ws, 1)
        X = torch.rand([9, 2, res, res], generator=g) * 2 - 1
        fea = [torch.randn([9, fnet.out_features], generator=g) for _ in range(2)]
        cases = {
            'latent': dict(w_latent=0.5, w_pix=0.0, w_disc=0.0, w_lpips=0.0),
            'pix':    dict(w_latent=0.0, w_pix=2.0, w_disc=0.0, w_lpips=0.0),
            'disc':   dict(w_latent=0.0, w_pix=0.0, w_disc=1.0, w_lpips=0.0),
            'lpips':  dict(w_latent=0.0, w_pix=0.0, w_disc=0.0, w_lpips=3.0),
            'all':    dict(w_latent=0.3, w_pix=1.0, w_disc=0.5, w_lpips=2.0),
            'soft':   dict(w_latent=0.3, w_pix=1.0, w_disc=0.0, w_lpips=0.0, soft_aug=True, alpha=0.7),
        }
        out = dict(res=np.array(res), cbase=np.array(cbase), cmax=np.array(cmax), wdim=np.array(wdim),
                   w0=T(w0), W=T(W), X=T(X), fea0=T(fea[0]), fea1=T(fea[1]), epochs=np.array(5), lr=np.array(0.01),
                   crop=np.array(8))
        for name, kw in cases.items():
            m = build_ref_module(G, D, W, X, fea, fnet, res=res, batch=B, epochs=5, lr=0.01, crop=8, **kw)
            random.seed(6)
            torch.manual_seed(123)
            img, w_aug = m.forward(w0.clone(), ['a', 'b'])
            out[f'{name}_img'] = T(img)
            out[f'{name}_w_aug'] = T(w_aug)
            # crop position the reference drew (first draw after random.seed(6))
            random.seed(6)
            out[f'{name}_crop_pos'] = np.array(ref_ud.get_params(res, 8, 'center_random_crop')['crop_pos'])
            print(name, float(img.abs().mean()), float((w_aug[:, 0] - w0[:, 0]).abs().max()))
        # forward_ganrand (mapping + synthesis), truncation 0.7
        m = build_ref_module(G, D, W, X, None, None, res=res, batch=B, epochs=0, lr=0.01, crop=8,
                             w_latent=0, w_pix=0, w_disc=0, w_lpips=0)
        m.truncation_psi = 0.7
        with torch.no_grad():
            G.mapping.w_avg.copy_(torch.randn([wdim], generator=g))
        z = torch.randn([B, wdim], generator=g)
        torch.manual_seed(321)
        img, ws = m.forward_ganrand(z)
        out['ganrand_z'], out['ganrand_img'], out['ganrand_ws'] = T(z), T(img), T(ws)
        out['w_avg'] = T(G.mapping.w_avg)
    finally:
        our_nets.ops = our_ops
    np.savez_compressed(os.path.join(HERE, 'latent_loop.npz'), **out)
    print('latent_loop.npz', len(out), 'arrays')


if __name__ == '__main__':
    gen_l0()
    gen_criteria()
    gen_loop()
