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
