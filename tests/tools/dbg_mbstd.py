"""one-off: per-sample error of the discriminator gradient with one tiny sample in the MinibatchStd group (tests/... one_tiny_sample)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sg2_networks as nets
from latentaugment_amd import _lib
if os.environ.get('LA_LIB'):
    import ctypes
    _lib.LIB_PATH = os.environ['LA_LIB']
    import torch as _t  # noqa
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    for _k in list(_lib.SIGNATURES):
        if not hasattr(_probe, _k):
            del _lib.SIGNATURES[_k]
from latentaugment_amd.synthesis import DiscriminatorEngine
dev = torch.device('cuda:0')
MODE = os.environ.get('DL_MODE', 'rand')
for shrink in (1.0, 2.0 ** -13):
    D = nets.make_discriminator(img_resolution=64, img_channels=2, channel_base=4096, channel_max=128, seed=2)
    with torch.no_grad():
        for n, p in D.named_parameters():
            if n.endswith('bias'):
                p.zero_()
    eng = DiscriminatorEngine(D, dev, max_batch=4, precision='f16x2')
    gen = torch.Generator().manual_seed(9)
    x = torch.randn([4, 2, 64, 64], generator=gen)
    x[2] *= shrink
    dl = torch.randn([4, 1], generator=gen)
    if MODE == 'ones':
        dl = torch.ones([4, 1])
    elif MODE == 'flip':
        dl = dl.flip(0).contiguous()
    print('dl', dl.flatten().tolist())
    res = {}
    for dt in (torch.float32, torch.float64):
        Dd = D.to(dt)
        nets.COMPUTE_DTYPE = dt
        xr = x.to(dt).clone().requires_grad_(True)
        lg = Dd(xr, None)
        (g,) = torch.autograd.grad(lg, [xr], dl.to(dt))
        res[dt] = (lg.detach().double(), g.double())
    D.float(); nets.COMPUTE_DTYPE = torch.float32
    logits = eng.forward(x.to(dev)).cpu().double()
    gx = eng.backward(dl.to(dev)).cpu().double()
    l64, g64 = res[torch.float64]; l32, g32 = res[torch.float32]
    print(f'shrink {shrink:.1e}: logits err HIP {float((logits - l64).abs().max()):.2e} ref32 {float((l32 - l64).abs().max()):.2e}')
    for n in range(4):
        m = float(g64[n].abs().max())
        print(f'   sample {n}: max|g| {m:.2e}  HIP err max {float((gx[n] - g64[n]).abs().max()) / m:.2e} rms {float((gx[n] - g64[n]).pow(2).mean().sqrt()) / m:.2e} | '
              f'ref32 err max {float((g32[n] - g64[n]).abs().max()) / m:.2e} rms {float((g32[n] - g64[n]).pow(2).mean().sqrt()) / m:.2e}')
