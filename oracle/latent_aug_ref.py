"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the latent-optimisation loop and its criteria.

Restates augments/utils/util_latent_aug.py::LatentAug.forward (:207-310) and its helpers, with the
disk-dependent constructor replaced by explicit arguments (G, D, banks).  Pinned against the
reference's own LatentAug.forward run in the build container (tests/golden/latent_loop_*.npz).
"""
import math
import random

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# crops (augments/utils/util_dataset.py:284-332)

def center_crop_size(load_size):
    """util_dataset.py:317-323: CenterCrop(int(sqrt(res^2/2))) -> 181 / 362 / 724."""
    return int(math.sqrt((load_size * load_size) / 2))


def center_crop(img, size):
    """torchvision CenterCrop semantics: top = int(round((H - size) / 2.0)) (published formula;
    torchvision itself is not installed in the build container)."""
    h, w = img.shape[-2:]
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img[..., top:top + size, left:left + size]


def get_crop_params(load_size, crop_size, preprocess='center_random_crop', rng=random):
    """util_dataset.py:284-296: (x, y) drawn with python `random.randint` (inclusive bounds)."""
    assert preprocess in ('center_random_crop', 'random_crop')
    new = load_size
    if preprocess == 'center_random_crop':
        new = center_crop_size(load_size)
    x = rng.randint(0, max(0, new - crop_size))
    y = rng.randint(0, max(0, new - crop_size))
    return x, y


def apply_aug_transform(img, load_size, crop_size, preprocess, pos):
    """util_dataset.py:298-315 + crop :325-332."""
    if preprocess in ('center_crop', 'center_random_crop'):
        img = center_crop(img, center_crop_size(load_size))
    if preprocess in ('random_crop', 'center_random_crop'):
        x1, y1 = pos
        _, _, ow, oh = img.shape
        if ow > crop_size or oh > crop_size:
            img = img[:, :, y1:y1 + crop_size, x1:x1 + crop_size]
    return img


# ---------------------------------------------------------------------------------------------
# criteria

def l2_loss_vectorized(X, Y, compute_mean=True):
    """Pairwise squared L2 in GEMM form; util_latent_aug.py:315-361.

    D[m,n] = |Y_m|^2 + |X_n|^2 - 2 <Y_m, X_n>;  mean: sum(D)/(m*n)/prod(feature dims)."""
    assert X.ndim == Y.ndim and X.ndim in (2, 3, 4)
    n = X.shape[0]
    m = Y.shape[0]
    feat = 1
    for s in Y.shape[1:]:
        feat *= s
    Xf = X.reshape(n, -1)
    Yf = Y.reshape(m, -1)
    YY = Yf.square().sum(1)
    XX = Xf.square().sum(1)
    YX = Yf @ Xf.t()                       # einsum('nchw,mchw->nm', [Y, X]) : [m, n]
    D = (YY.unsqueeze(-1) + XX) - 2 * YX
    if compute_mean:
        D = D.sum() / (m * n)
        D = D / feat
    return D


def loss_latent(ws, W, w_latent):
    """util_latent_aug.py:427-433."""
    return l2_loss_vectorized(ws, W) * w_latent


def loss_pix(x, x_tr, w_pix):
    """util_latent_aug.py:373-385 (x, x_tr already centre-cropped)."""
    n_modes = x.shape[1]
    loss = 0.0
    for i in range(n_modes):
        loss = loss + l2_loss_vectorized(x[:, i:i + 1], x_tr[:, i:i + 1]) * w_pix
    return loss / n_modes


def loss_disc(D, x, w_disc):
    """util_latent_aug.py:363-371."""
    return F.softplus(-D(x, None)).mean() * w_disc


def loss_lpips(feature_net, x_crop, fea_banks, w_lpips):
    """util_latent_aug.py:387-409 with `feature_net(x[b,3,h,w]) -> [b,F]` standing in for the
    (unobtainable offline) vgg16.pt(return_lpips=True)."""
    n_modes = x_crop.shape[1]
    loss = 0.0
    for i in range(n_modes):
        x = x_crop[:, i:i + 1].repeat([1, 3, 1, 1])
        fs = feature_net(x)
        d = l2_loss_vectorized(fs, fea_banks[i], compute_mean=False)
        loss = loss + d.sum() / (fs.shape[0] * fea_banks[i].shape[0]) * w_lpips
    return loss / n_modes


# ---------------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults used at util_latent_aug.py:213: betas (0.9,0.999), eps 1e-8,
# no weight decay, no amsgrad).  Restated explicitly; pinned against torch.optim.Adam in tests.

class AdamState:
    def __init__(self, p, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.t = 0

    def step(self, p, g):
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * g
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = self.v.sqrt() / math.sqrt(bc2) + self.eps
        return p - (self.lr / bc1) * self.m / denom


# ---------------------------------------------------------------------------------------------
# the loop

class LatentAugRef:
    """Restates LatentAug.forward (util_latent_aug.py:207-310) for injected G/D and banks."""

    def __init__(self, G, D=None, W=None, X=None, fea=None, feature_net=None, res=256, num_epochs=5,
                 opt_lr=0.01, w_latent=0.0, w_pix=0.0, w_disc=0.0, w_lpips=0.0, crop_size=64,
                 preprocess='center_random_crop', soft_aug=False, alpha=1.0, final_noise_mode='random',
                 fused_modconv=True, dtype=torch.float32):
        self.G, self.D, self.W, self.X, self.fea, self.feature_net = G, D, W, X, fea, feature_net
        self.res, self.num_epochs, self.opt_lr = res, num_epochs, opt_lr
        self.w_latent, self.w_pix, self.w_disc, self.w_lpips = w_latent, w_pix, w_disc, w_lpips
        self.crop_size, self.preprocess = crop_size, preprocess
        self.soft_aug, self.alpha = soft_aug, alpha
        self.final_noise_mode = final_noise_mode
        self.fused_modconv = fused_modconv
        self.dtype = dtype          # float64: tolerance anchor (the caller also moves G/D/banks and sets sg2_networks.COMPUTE_DTYPE)
        self.num_ws = G.num_ws
        self.trace = None

    def broadcasting(self, w):
        return w.repeat([1, self.num_ws, 1])            # util_latent_aug.py:493-494

    def forward(self, w, crop_pos=None, record=False):
        """w [b,1,512] -> (img [b,C,r,r], w_aug [b,num_ws,512]).  crop_pos: (x,y) or None (draw)."""
        w = w.detach().to(self.dtype)
        w_opt = w.clone()
        adam = AdamState(w_opt, self.opt_lr)
        if crop_pos is None:
            crop_pos = get_crop_params(self.res, self.crop_size, self.preprocess)
        cc = center_crop_size(self.res)
        trace = {'w': [], 'loss': [], 'loss_latent': [], 'loss_pix': [], 'loss_disc': [], 'loss_lpips': [],
                 'grad': []}
        Xc = center_crop(self.X, cc) if (self.w_pix > 0 and self.X is not None) else None
        for _ in range(self.num_epochs):
            wv = w_opt.clone().requires_grad_(True)
            ws = self.broadcasting(wv)
            x = self.G.synthesis(ws, noise_mode='const', fused_modconv=self.fused_modconv)   # :227
            ll = lp = ld = lf = torch.zeros([])
            if self.w_latent > 0:
                ll = loss_latent(ws, self.W, self.w_latent)
            if self.w_disc > 0:
                ld = loss_disc(self.D, x, self.w_disc)
            if self.w_pix > 0:
                lp = loss_pix(center_crop(x, cc), Xc, self.w_pix)
            if self.w_lpips > 0:
                xa = apply_aug_transform(x, self.res, self.crop_size, self.preprocess, crop_pos)
                lf = loss_lpips(self.feature_net, xa, self.fea, self.w_lpips)
            loss = -ll - lp - lf + ld                                                     # :270
            (g,) = torch.autograd.grad(loss, wv)
            w_opt = adam.step(w_opt, g)
            if record:
                trace['w'].append(w_opt.clone())
                trace['grad'].append(g.clone())
                trace['loss'].append(float(loss))
                trace['loss_latent'].append(float(ll))
                trace['loss_pix'].append(float(lp))
                trace['loss_disc'].append(float(ld))
                trace['loss_lpips'].append(float(lf))
        if self.soft_aug:                                                                 # :303-306
            w_aug = self.broadcasting(self.alpha * w_opt + (1 - self.alpha) * w)
        else:
            w_aug = self.broadcasting(w_opt)
        with torch.no_grad():
            img = self.G.synthesis(w_aug, noise_mode=self.final_noise_mode,
                                   fused_modconv=self.fused_modconv)                      # :308,:488
        self.trace = trace if record else None
        return img, w_aug
