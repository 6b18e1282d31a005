"""A synthetic stand-in for NVIDIA's TorchScript `vgg16.pt` (the reference loads the real one from a URL,
util_latent_aug.py:35-43): VGG16 topology at reduced width, random weights, the call signature the reference uses
(`module(x, resize_images=False, return_lpips=True)`, :395), an ImageNet-style input layer held in buffers and the five LPIPS
channel weights held as [1,C,1,1] buffers.  Scripted with torch.jit.script and saved, so that loading goes through
torch.jit.load exactly as in the reference.  Test data only -- no reference source is involved."""
import torch
import torch.nn as nn
import torch.nn.functional as F

_CFG = [(1, 2), (2, 2), (4, 3), (8, 3), (8, 3)]


class _ScriptVGG(nn.Module):
    def __init__(self, width=8, seed=3, lin_sqrt=False, with_norm=True):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        convs = []
        c = 3
        chans = []
        for mult, n in _CFG:
            for _ in range(n):
                co = mult * width
                m = nn.Conv2d(c, co, 3, padding=1)
                with torch.no_grad():
                    m.weight.copy_(torch.randn([co, c, 3, 3], generator=g) * (2.0 / (c * 9)) ** 0.5)
                    m.bias.copy_(torch.randn([co], generator=g) * 0.05)
                convs.append(m)
                c = co
            chans.append(c)
        self.layers = nn.ModuleList(convs)
        self.lin_sqrt = lin_sqrt
        self.with_norm = with_norm
        for i, ch in enumerate(chans):
            lin = torch.rand([1, ch, 1, 1], generator=g) + 0.1
            self.register_buffer(f'lpips{i}', lin.sqrt() if lin_sqrt else lin)
        if with_norm:
            self.register_buffer('mean', torch.tensor([123.675, 116.28, 103.53]).reshape(1, 3, 1, 1))
            self.register_buffer('std', torch.tensor([58.395, 57.12, 57.375]).reshape(1, 3, 1, 1))
        else:
            self.register_buffer('mean', torch.zeros(1, 3, 1, 1))
            self.register_buffer('std', torch.ones(1, 3, 1, 1))

    def _pack(self, f, lin):
        n = f * torch.rsqrt(f.square().sum(dim=1, keepdim=True) + 1e-10)
        w = lin if self.lin_sqrt else lin.sqrt()
        n = n * w / float(f.shape[2] * f.shape[3]) ** 0.5
        return n.flatten(1)

    def forward(self, img, resize_images: bool = True, return_features: bool = False, return_lpips: bool = False):
        x = (img.to(torch.float32) - self.mean) / self.std
        if resize_images:
            x = F.interpolate(x, size=(224, 224), mode='bilinear', align_corners=False)
        outs = []
        k = 0
        for conv in self.layers:
            x = F.relu(conv(x))
            if k == 1:
                outs.append(self._pack(x, self.lpips0))
                x = F.max_pool2d(x, 2)
            elif k == 3:
                outs.append(self._pack(x, self.lpips1))
                x = F.max_pool2d(x, 2)
            elif k == 6:
                outs.append(self._pack(x, self.lpips2))
                x = F.max_pool2d(x, 2)
            elif k == 9:
                outs.append(self._pack(x, self.lpips3))
                x = F.max_pool2d(x, 2)
            elif k == 12:
                outs.append(self._pack(x, self.lpips4))
            k += 1
        if return_lpips:
            return torch.cat(outs, dim=1)
        return x.flatten(1)


def save_scripted_vgg(path, **kw):
    m = torch.jit.script(_ScriptVGG(**kw).eval())
    m.save(str(path))
    return m


def eval_ops_cpu(ops, x):
    """Plain-torch evaluation of a FeatureEngine op list (checker for the loader's mapping; CPU)."""
    outs = []
    for op in ops:
        if op[0] == 'conv':
            x = F.relu(F.conv2d(x, op[1], op[2], padding=1))
        elif op[0] == 'tap':
            n = x * torch.rsqrt(x.square().sum(dim=1, keepdim=True) + 1e-10)
            outs.append((n * op[1].sqrt().reshape(1, -1, 1, 1) / float(x.shape[2] * x.shape[3]) ** 0.5).flatten(1))
        elif op[0] == 'maxpool':
            x = F.max_pool2d(x, 2)
        elif op[0] == 'avgpool':
            x = F.avg_pool2d(x, 2)
    return torch.cat(outs, dim=1)
