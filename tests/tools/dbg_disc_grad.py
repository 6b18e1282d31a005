"""one-off: per-sample error of the discriminator's image gradient against float64, over precision / batch / resolution"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sg2_networks as nets
from latentaugment_amd import _lib
if os.environ.get('LA_DEV_BUILD'):
    _lib.select_dev_build()
from latentaugment_amd.synthesis import DiscriminatorEngine
dev = torch.device('cuda:0')
bias_mode = os.environ.get('BIAS', 'zero')
for res, cb, cm in ((32, 2048, 128), (64, 4096, 128), (128, 8192, 128)):
    for B in (1, 2, 4, 8):
        D = nets.make_discriminator(img_resolution=res, img_channels=2, channel_base=cb, channel_max=cm, seed=2)
        with torch.no_grad():
            for n, p in D.named_parameters():
                if n.endswith('bias'):
                    p.zero_() if bias_mode == 'zero' else p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))) * 0.1)
        gen = torch.Generator().manual_seed(9)
        x = torch.randn([B, 2, res, res], generator=gen)
        dl = torch.ones([B, 1])
        nets.COMPUTE_DTYPE = torch.float64
        Dd = D.double()
        xr = x.double().clone().requires_grad_(True)
        (g64,) = torch.autograd.grad(Dd(xr, None), [xr], dl.double())
        D.float(); nets.COMPUTE_DTYPE = torch.float32
        row = []
        for prec in ('f32', 'f16x2', 'bf16x3'):
            eng = DiscriminatorEngine(D, dev, max_batch=B, precision=prec)
            eng.forward(x.to(dev))
            gx = eng.backward(dl.to(dev)).cpu().double()
            row.append(prec + ' ' + ' '.join(f'{float((gx[n] - g64[n]).pow(2).mean().sqrt() / g64[n].abs().max()):.1e}' for n in range(B)))
        print(f'res {res} B {B} bias {bias_mode}: rms err / max per sample | ' + ' | '.join(row), flush=True)
