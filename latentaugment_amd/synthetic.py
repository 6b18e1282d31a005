"""Synthetic stand-ins for the artefacts the reference loads from disk (BASELINE.md "Synthetic inputs").

No dataset, network pickle or VGG weights are reachable offline, so benchmarks and smoke tests use:
  * a random-init SG2 generator state_dict with the reference's parameter names (legacy.py:171-203) and SG2's
    default initialisation (weights randn, biases 0, affine.bias 1, const / noise_const randn, noise_strength 0);
  * latents w0 = randn([B,1,512]) (seed 1), batch dict A,B = randn.clamp(-1,1) (seed 2),
    banks W (seed 3) and X (seed 4), python-`random` crop seed 6.
Pure host-side tensor construction (plumbing); nothing here computes the hot path.
"""
import math

import torch


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    """C[r] = min(channel_base // r, channel_max)  (config-e: 16384, config-f: 32768; legacy.py:128-129)."""
    return {2 ** i: min(channel_base // (2 ** i), channel_max) for i in range(2, int(math.log2(img_resolution)) + 1)}


def make_generator_state_dict(img_resolution=256, img_channels=2, channel_base=32768, channel_max=512, w_dim=512,
                              seed=0, noise_strength=0.0, mapping_layers=8, lr_multiplier=0.01):
    """state_dict of a random-init SG2 generator (architecture 'skip'), keys as in the reference's G_ema."""
    g = torch.Generator().manual_seed(seed)
    ch = channels_dict(img_resolution, channel_base, channel_max)
    sd = {}
    num_ws = 2 * int(math.log2(img_resolution)) - 2
    f1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    fir = torch.outer(f1, f1) / 64.0

    def conv(prefix, cin, cout, res, up):
        sd[f'{prefix}.affine.weight'] = torch.randn([cin, w_dim], generator=g)
        sd[f'{prefix}.affine.bias'] = torch.ones([cin])
        sd[f'{prefix}.weight'] = torch.randn([cout, cin, 3, 3], generator=g)
        sd[f'{prefix}.noise_const'] = torch.randn([res, res], generator=g)
        sd[f'{prefix}.noise_strength'] = torch.tensor(float(noise_strength))
        sd[f'{prefix}.bias'] = torch.zeros([cout])
        sd[f'{prefix}.resample_filter'] = fir.clone()

    for res in sorted(ch):
        p = f'synthesis.b{res}'
        if res == 4:
            sd[f'{p}.const'] = torch.randn([ch[4], 4, 4], generator=g)
        else:
            conv(f'{p}.conv0', ch[res // 2], ch[res], res, 2)
        conv(f'{p}.conv1', ch[res], ch[res], res, 1)
        sd[f'{p}.torgb.affine.weight'] = torch.randn([ch[res], w_dim], generator=g)
        sd[f'{p}.torgb.affine.bias'] = torch.ones([ch[res]])
        sd[f'{p}.torgb.weight'] = torch.randn([img_channels, ch[res], 1, 1], generator=g)
        sd[f'{p}.torgb.bias'] = torch.zeros([img_channels])
        sd[f'{p}.resample_filter'] = fir.clone()
    for i in range(mapping_layers):
        sd[f'mapping.fc{i}.weight'] = torch.randn([w_dim, w_dim], generator=g) / lr_multiplier
        sd[f'mapping.fc{i}.bias'] = torch.zeros([w_dim])
    sd['mapping.w_avg'] = torch.zeros([w_dim])
    meta = dict(img_resolution=img_resolution, img_channels=img_channels, w_dim=w_dim, z_dim=w_dim, num_ws=num_ws,
                channels=ch)
    return sd, meta


def make_latents(batch, w_dim=512, seed=1):
    return torch.randn([batch, 1, w_dim], generator=torch.Generator().manual_seed(seed))


def make_batch(batch, res=256, seed=2):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn([batch, 1, res, res], generator=g).clamp(-1, 1)
    B = torch.randn([batch, 1, res, res], generator=g).clamp(-1, 1)
    paths = [f'train/p{i:03d}/s_{10 + 5 * (i % 23):05d}.pickle' for i in range(batch)]
    return {'A': A, 'B': B, 'A_paths': paths, 'B_paths': list(paths)}


def make_banks(num_ws, res=256, img_channels=2, w_dim=512, M_w=1024, M_x=256, seed_w=3, seed_x=4):
    W = torch.randn([M_w, 1, w_dim], generator=torch.Generator().manual_seed(seed_w)).repeat(1, num_ws, 1)
    X = torch.rand([M_x, img_channels, res, res], generator=torch.Generator().manual_seed(seed_x)) * 2 - 1
    return W, X


def make_feature_banks(M_x, F, img_channels=2, seed=5):
    """fea_<mode> = randn([M_x, F]) / sqrt(F) per modality (seed 5): stand-in for the real-image LPIPS feature banks
    (util_latent_aug.py:160-171); unit-norm rows, like LPIPS feature vectors."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn([M_x, F], generator=g) * (1.0 / F) ** 0.5 for _ in range(img_channels)]


def make_discriminator_state_dict(img_resolution=256, img_channels=2, channel_base=32768, channel_max=512, seed=1000):
    """state_dict of a random-init SG2 discriminator (architecture 'resnet'), keys as in the reference's D
    (legacy.py:271-288): weights randn, biases 0."""
    g = torch.Generator().manual_seed(seed)
    ch = channels_dict(img_resolution, channel_base, channel_max)
    sd = {}
    for res in sorted(ch, reverse=True):
        if res == 4:
            break
        p = f'b{res}'
        if res == img_resolution:
            sd[f'{p}.fromrgb.weight'] = torch.randn([ch[res], img_channels, 1, 1], generator=g)
            sd[f'{p}.fromrgb.bias'] = torch.zeros([ch[res]])
        sd[f'{p}.conv0.weight'] = torch.randn([ch[res], ch[res], 3, 3], generator=g)
        sd[f'{p}.conv0.bias'] = torch.zeros([ch[res]])
        sd[f'{p}.conv1.weight'] = torch.randn([ch[res // 2], ch[res], 3, 3], generator=g)
        sd[f'{p}.conv1.bias'] = torch.zeros([ch[res // 2]])
        sd[f'{p}.skip.weight'] = torch.randn([ch[res // 2], ch[res], 1, 1], generator=g)
    sd['b4.conv.weight'] = torch.randn([ch[4], ch[4] + 1, 3, 3], generator=g)
    sd['b4.conv.bias'] = torch.zeros([ch[4]])
    sd['b4.fc.weight'] = torch.randn([ch[4], ch[4] * 16], generator=g)
    sd['b4.fc.bias'] = torch.zeros([ch[4]])
    sd['b4.out.weight'] = torch.randn([1, ch[4]], generator=g)
    sd['b4.out.bias'] = torch.zeros([1])
    return sd


def make_vgg16_lpips_ops(seed=7, width=64, in_ch=3):
    """Op list (for synthesis.FeatureEngine) of a VGG16-shaped LPIPS net with random He-initialised weights: 13 x
    conv3x3+ReLU, 4 x max-pool, taps after relu1_2 / 2_2 / 3_3 / 4_3 / 5_3 with random positive lin weights.  Stand-in for
    NVIDIA's vgg16.pt (util_latent_aug.py:36), which cannot be downloaded offline."""
    g = torch.Generator().manual_seed(seed)
    ops = []
    c = in_ch
    cfg = [(1, 2), (2, 2), (4, 3), (8, 3), (8, 3)]
    for si, (mult, n) in enumerate(cfg):
        for _ in range(n):
            co = mult * width
            ops.append(('conv', torch.randn([co, c, 3, 3], generator=g) * (2.0 / (c * 9)) ** 0.5,
                        torch.randn([co], generator=g) * 0.05))
            c = co
        ops.append(('tap', torch.rand([c], generator=g) + 0.1))
        if si + 1 < len(cfg):
            ops.append(('maxpool',))
    return ops


def lpips_num_features(crop=64, width=64):
    return sum(m * width * (crop >> i) ** 2 for i, (m, _) in enumerate([(1, 2), (2, 2), (4, 3), (8, 3), (8, 3)]))
