"""Oracle (TEST INFRASTRUCTURE): StyleGAN2 generator / discriminator in torch-CPU fp32.

The reference tree does NOT contain the SG2 network source (models/stylegan3/legacy.py:120-121
imports `training.networks_stylegan2`, which is absent; the class source travels inside network
pickles via torch_utils/persistence.py:118-126).  This file restates the public NVlabs definition
recorded in SURVEY.md Appendix A.  **Parity unpinned** for this layer: it is anchored only by
  * parameter / buffer names and layouts: legacy.py:171-203 (G), :271-288 (D)
  * constructor defaults: legacy.py:122-144 (G), :220-247 (D)
  * the L0 ops it is composed from (oracle/sg2_ops.py, pinned to reference goldens)
  * call sites: augments/utils/util_latent_aug.py:203,227,367,460,488.
`state_dict()` keys equal the reference names so real checkpoints' tensors map one-to-one.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import sg2_ops as ops

# The reference's networks force fp32 off-GPU; tests that anchor a tolerance against a float64 run of
# the same definition set this to torch.float64 (and move the module with .double()) for the duration of that run.
COMPUTE_DTYPE = torch.float32


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    log2 = int(math.log2(img_resolution))
    return {2 ** i: min(channel_base // (2 ** i), channel_max) for i in range(2, log2 + 1)}


class FullyConnectedLayer(nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1.0,
                 bias_init=0.0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([out_features], float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight * self.weight_gain
        b = self.bias
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        x = x.matmul(w.t())
        return ops.bias_act(x, b, act=self.activation)


class MappingNetwork(nn.Module):
    """z -> normalise -> 8 x FC(lrelu, lr_mul 0.01) -> broadcast -> truncation lerp (legacy.py:137-142,172-176)."""

    def __init__(self, z_dim, w_dim, num_ws, num_layers=8, lr_multiplier=0.01):
        super().__init__()
        self.z_dim, self.w_dim, self.num_ws, self.num_layers = z_dim, w_dim, num_ws, num_layers
        feats = [z_dim] + [w_dim] * num_layers
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(feats[i], feats[i + 1], activation='lrelu',
                                                        lr_multiplier=lr_multiplier))
        self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c=None, truncation_psi=1.0, truncation_cutoff=None):
        x = z.to(COMPUTE_DTYPE)
        x = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.resolution, self.up = in_channels, out_channels, resolution, up
        self.conv_clamp = conv_clamp
        self.padding = kernel_size // 2
        self.register_buffer('resample_filter', ops.setup_filter(resample_filter))
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.register_buffer('noise_const', torch.randn([resolution, resolution]))
        self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1.0, noise=None):
        styles = self.affine(w)
        nz = None
        if noise is not None:                       # explicit noise tensor [B,1,r,r] (already unit-variance)
            nz = noise * self.noise_strength
        elif noise_mode == 'random':
            nz = torch.randn([x.shape[0], 1, self.resolution, self.resolution]) * self.noise_strength
        elif noise_mode == 'const':
            nz = self.noise_const * self.noise_strength
        x = ops.modulated_conv2d(x, self.weight, styles, noise=nz, up=self.up, padding=self.padding,
                                 resample_filter=self.resample_filter, flip_weight=(self.up == 1),
                                 fused_modconv=fused_modconv)
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return ops.bias_act(x, self.bias, act='lrelu', gain=ops.SQRT2 * gain, clamp=clamp)


class ToRGBLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1.0 / math.sqrt(in_channels * kernel_size ** 2)

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        x = ops.modulated_conv2d(x, self.weight, styles, demodulate=False, fused_modconv=fused_modconv)
        return ops.bias_act(x, self.bias, clamp=self.conv_clamp)


class SynthesisBlock(nn.Module):
    """architecture='skip' (legacy.py:132)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.in_channels, self.resolution, self.is_last = in_channels, resolution, is_last
        self.register_buffer('resample_filter', ops.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim, resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim, resolution, conv_clamp=conv_clamp)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

    def forward(self, x, img, ws, noise_mode='random', fused_modconv=True, noises=None):
        w_iter = iter(ws.unbind(dim=1))
        noises = list(noises) if noises is not None else None
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = self.conv1(x, next(w_iter), noise_mode=noise_mode, fused_modconv=fused_modconv,
                           noise=noises.pop(0) if noises else None)
        else:
            x = self.conv0(x, next(w_iter), noise_mode=noise_mode, fused_modconv=fused_modconv,
                           noise=noises.pop(0) if noises else None)
            x = self.conv1(x, next(w_iter), noise_mode=noise_mode, fused_modconv=fused_modconv,
                           noise=noises.pop(0) if noises else None)
        if img is not None:
            img = ops.upsample2d(img, self.resample_filter)
        y = self.torgb(x, next(w_iter), fused_modconv=fused_modconv)
        img = img + y if img is not None else y
        return x, img


class SynthesisNetwork(nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512,
                 conv_clamp=256, resample_filter=(1, 3, 3, 1)):
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.block_resolutions = [2 ** i for i in range(2, int(math.log2(img_resolution)) + 1)]
        ch = channels_dict(img_resolution, channel_base, channel_max)
        self.channels = ch
        self.num_ws = 0
        for res in self.block_resolutions:
            cin = ch[res // 2] if res > 4 else 0
            block = SynthesisBlock(cin, ch[res], w_dim, res, img_channels, is_last=(res == img_resolution),
                                   resample_filter=resample_filter, conv_clamp=conv_clamp)
            self.num_ws += block.num_conv
            if res == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, noise_mode='random', fused_modconv=True, noises=None, return_features=False):
        """noises: optional list (one per SynthesisLayer, in execution order) of [B,1,r,r] unit noise."""
        ws = ws.to(COMPUTE_DTYPE)
        x = img = None
        w_idx = 0
        n_idx = 0
        feats = []
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            bw = ws.narrow(1, w_idx, block.num_conv + block.num_torgb)
            w_idx += block.num_conv
            bn = None
            if noises is not None:
                bn = noises[n_idx:n_idx + block.num_conv]
                n_idx += block.num_conv
            x, img = block(x, img, bw, noise_mode=noise_mode, fused_modconv=fused_modconv, noises=bn)
            feats.append(x)
        if return_features:
            return img, feats
        return img


class Generator(nn.Module):
    """Contract used by the hot path: .mapping(z,c,truncation_psi), .synthesis(ws,noise_mode=),
    .z_dim/.w_dim/.num_ws  (util_latent_aug.py:119-121,203,227,460,488)."""

    def __init__(self, z_dim=512, w_dim=512, img_resolution=256, img_channels=2, channel_base=32768,
                 channel_max=512, conv_clamp=256, mapping_layers=8):
        super().__init__()
        self.z_dim, self.w_dim, self.c_dim = z_dim, w_dim, 0
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim, img_resolution, img_channels, channel_base, channel_max,
                                          conv_clamp)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim, w_dim, self.num_ws, num_layers=mapping_layers)

    def forward(self, z, c=None, truncation_psi=1.0, **kw):
        return self.synthesis(self.mapping(z, c, truncation_psi=truncation_psi), **kw)


# ----------------------------------------------------------------------------------------------
# Discriminator (architecture 'resnet', legacy.py:224,271-288)

class Conv2dLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer('resample_filter', ops.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1.0 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = ops.act_defaults(activation)[1]
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels])) if bias else None

    def forward(self, x, gain=1.0):
        w = self.weight * self.weight_gain
        x = ops.conv2d_resample(x, w, f=self.resample_filter, up=self.up, down=self.down,
                                padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return ops.bias_act(x, self.bias, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first,
                 conv_clamp=None):
        super().__init__()
        self.in_channels, self.resolution, self.first = in_channels, resolution, first
        if first:
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, 1, activation='lrelu', conv_clamp=conv_clamp)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, 3, activation='lrelu', conv_clamp=conv_clamp)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, 3, activation='lrelu', down=2, conv_clamp=conv_clamp)
        self.skip = Conv2dLayer(tmp_channels, out_channels, 1, bias=False, down=2)

    def forward(self, x, img):
        if self.first:
            x = self.fromrgb(img)
        y = self.skip(x, gain=math.sqrt(0.5))
        x = self.conv0(x)
        x = self.conv1(x, gain=math.sqrt(0.5))
        return y + x


class MinibatchStdLayer(nn.Module):
    def __init__(self, group_size=4, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = min(self.group_size, N) if self.group_size is not None else N
        Fc = self.num_channels
        c = C // Fc
        y = x.reshape(G, -1, Fc, c, H, W)
        y = y - y.mean(dim=0)
        y = y.square().mean(dim=0)
        y = (y + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4])
        y = y.reshape(-1, Fc, 1, 1).repeat(G, 1, H, W)
        return torch.cat([x, y], dim=1)


class DiscriminatorEpilogue(nn.Module):
    def __init__(self, in_channels, resolution, mbstd_group_size=4, mbstd_num_channels=1, conv_clamp=None):
        super().__init__()
        self.mbstd = MinibatchStdLayer(mbstd_group_size, mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, 3, activation='lrelu',
                                conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * resolution ** 2, in_channels, activation='lrelu')
        self.out = FullyConnectedLayer(in_channels, 1)

    def forward(self, x):
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = self.fc(x.flatten(1))
        return self.out(x)


class Discriminator(nn.Module):
    """D(img, c=None) -> logits [N,1]  (call: util_latent_aug.py:367)."""

    def __init__(self, img_resolution=256, img_channels=2, channel_base=32768, channel_max=512, conv_clamp=256,
                 mbstd_group_size=4):
        super().__init__()
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.block_resolutions = [2 ** i for i in range(int(math.log2(img_resolution)), 2, -1)]
        ch = channels_dict(img_resolution, channel_base, channel_max)
        for res in self.block_resolutions:
            first = res == img_resolution
            setattr(self, f'b{res}', DiscriminatorBlock(0 if first else ch[res], ch[res], ch[res // 2], res,
                                                        img_channels, first, conv_clamp=conv_clamp))
        self.b4 = DiscriminatorEpilogue(ch[4], 4, mbstd_group_size=mbstd_group_size, conv_clamp=conv_clamp)

    def forward(self, img, c=None):
        x = None
        img = img.to(COMPUTE_DTYPE)
        for res in self.block_resolutions:
            x = getattr(self, f'b{res}')(x, img)
        return self.b4(x)


def make_generator(img_resolution=256, img_channels=2, channel_base=32768, channel_max=512, seed=0,
                   noise_strength=0.0, conv_clamp=256, w_dim=512, mapping_layers=8):
    """Random-init G exactly as BASELINE.md "Synthetic inputs": SG2 default init under manual_seed(seed)."""
    g = torch.Generator().manual_seed(seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        G = Generator(z_dim=w_dim, w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                      channel_base=channel_base, channel_max=channel_max, conv_clamp=conv_clamp,
                      mapping_layers=mapping_layers)
    finally:
        torch.random.set_rng_state(state)
    del g
    if noise_strength != 0.0:
        with torch.no_grad():
            for m in G.synthesis.modules():
                if isinstance(m, SynthesisLayer):
                    m.noise_strength.fill_(noise_strength)
    return G.eval().requires_grad_(False)


def make_discriminator(img_resolution=256, img_channels=2, channel_base=32768, channel_max=512, seed=0,
                       conv_clamp=256):
    state = torch.random.get_rng_state()
    torch.manual_seed(seed + 1000)
    try:
        D = Discriminator(img_resolution, img_channels, channel_base, channel_max, conv_clamp)
    finally:
        torch.random.set_rng_state(state)
    return D.eval().requires_grad_(False)
