"""Oracle (TEST INFRASTRUCTURE): LPIPS-shaped feature extractors with synthetic weights.

The reference's perceptual criterion calls NVIDIA's TorchScript `vgg16.pt(x, resize_images=False,
return_lpips=True)` (augments/utils/util_latent_aug.py:36,395), a network download that is not
available offline (**parity unpinned** for the weights).  What the criterion needs from it is a map
[b,3,h,w] -> [b,F] whose squared L2 distance is the LPIPS distance: per tapped layer, unit-normalise
over channels, scale by sqrt(lin weight), divide by sqrt(H*W), flatten and concatenate.  These
stand-ins keep exactly that structure with random weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _lpips_pack(feats, lins):
    outs = []
    for f, lin in zip(feats, lins):
        n = f * torch.rsqrt(f.square().sum(dim=1, keepdim=True) + 1e-10)
        n = n * lin.sqrt().reshape(1, -1, 1, 1) / float(f.shape[2] * f.shape[3]) ** 0.5
        outs.append(n.flatten(1))
    return torch.cat(outs, dim=1)


class TinyFeatureNet(nn.Module):
    """Two conv taps; used by the golden loop cases (crop 8x8 -> F = 8*64 + 16*16 = 768)."""

    def __init__(self, seed=5, crop=8):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn([8, 3, 3, 3], generator=g) * 0.3, requires_grad=False)
        self.b1 = nn.Parameter(torch.randn([8], generator=g) * 0.1, requires_grad=False)
        self.w2 = nn.Parameter(torch.randn([16, 8, 3, 3], generator=g) * 0.2, requires_grad=False)
        self.b2 = nn.Parameter(torch.randn([16], generator=g) * 0.1, requires_grad=False)
        self.lin1 = nn.Parameter(torch.rand([8], generator=g), requires_grad=False)
        self.lin2 = nn.Parameter(torch.rand([16], generator=g), requires_grad=False)
        self.out_features = 8 * crop * crop + 16 * (crop // 2) ** 2

    def forward(self, x):
        f1 = F.relu(F.conv2d(x, self.w1, self.b1, padding=1))
        f2 = F.relu(F.conv2d(F.avg_pool2d(f1, 2), self.w2, self.b2, padding=1))
        return _lpips_pack([f1, f2], [self.lin1, self.lin2])


class VGG16Features(nn.Module):
    """VGG16-shaped LPIPS feature net (13 conv3x3+ReLU, 4 max-pools, taps after relu1_2/2_2/3_3/4_3/5_3) with random
    He-initialised weights and random positive lin weights.  `width` scales the channel counts (64 = the real VGG16)."""
    CFG = [(1, 2), (2, 2), (4, 3), (8, 3), (8, 3)]      # (channel multiple, convs) per stage

    def __init__(self, seed=7, width=64, in_ch=3):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.stages = []
        c = in_ch
        k = 0
        for si, (mult, n) in enumerate(self.CFG):
            convs = []
            for _ in range(n):
                co = mult * width
                w = nn.Parameter(torch.randn([co, c, 3, 3], generator=g) * (2.0 / (c * 9)) ** 0.5, requires_grad=False)
                b = nn.Parameter(torch.randn([co], generator=g) * 0.05, requires_grad=False)
                self.register_parameter(f'w{k}', w)
                self.register_parameter(f'b{k}', b)
                convs.append((w, b))
                c = co
                k += 1
            lin = nn.Parameter(torch.rand([c], generator=g) + 0.1, requires_grad=False)
            self.register_parameter(f'lin{si}', lin)
            self.stages.append((convs, lin))

    def ops(self):
        """The op list latentaugment_amd.synthesis.FeatureEngine consumes."""
        out = []
        for si, (convs, lin) in enumerate(self.stages):
            for w, b in convs:
                out.append(('conv', w, b))
            out.append(('tap', lin))
            if si + 1 < len(self.stages):
                out.append(('maxpool',))
        return out

    def forward(self, x):
        feats, lins = [], []
        for si, (convs, lin) in enumerate(self.stages):
            for w, b in convs:
                x = F.relu(F.conv2d(x, w, b, padding=1))
            feats.append(x)
            lins.append(lin)
            if si + 1 < len(self.stages):
                x = F.max_pool2d(x, 2)
        return _lpips_pack(feats, lins)


def tiny_ops(net):
    """Op list of TinyFeatureNet for latentaugment_amd.synthesis.FeatureEngine."""
    return [('conv', net.w1, net.b1), ('tap', net.lin1), ('avgpool',), ('conv', net.w2, net.b2), ('tap', net.lin2)]
