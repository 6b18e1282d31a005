"""Diagnosis (GPU box): style gradients of a backward pass after a row-windowed forward against the same after a whole-frame forward,
per style segment (which layer's reduction sees rows that the windowed forward did not compute)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latentaugment_amd import _lib                                          # noqa: E402
from latentaugment_amd.synthesis import SynthesisEngine                     # noqa: E402
from oracle import sg2_networks as nets                                     # noqa: E402

dev = torch.device('cuda', 0)
res, lo, hi = 128, 27, 101
G = nets.make_generator(img_resolution=res, img_channels=2, channel_base=4096, channel_max=64, seed=3, noise_strength=0.1, w_dim=64, mapping_layers=1)
eng = SynthesisEngine.from_generator(G, dev, max_batch=2, precision='f16x2')
gen = torch.Generator().manual_seed(9)
ws_old = torch.randn([2, G.num_ws, 64], generator=gen) * 3
ws = torch.randn([2, G.num_ws, 64], generator=gen)
g_img = torch.zeros([2, 2, res, res])
g_img[:, :, lo:hi] = torch.randn([2, 2, hi - lo, res], generator=gen)
lib = _lib.load()
eng.forward(ws.to(dev), noise_mode='const')
dws_full = eng.backward(g_img.to(dev)).clone()
ds_full = eng.style_grads(2).clone()
ys_full = [eng.layer_output(k, 2).clone() for k in range(len(eng.layer_resolutions))]
eng.forward(ws_old.to(dev), noise_mode='const')
_lib.check(lib.la_synth_set_row_window(eng.handle, lo, hi), 'w')
eng.forward(ws.to(dev), noise_mode='const')
ys_win = [eng.layer_output(k, 2).clone() for k in range(len(eng.layer_resolutions))]
dws_win = eng.backward(g_img.to(dev)).clone()
ds_win = eng.style_grads(2).clone()
print('dws max diff', float((dws_full - dws_win).abs().max()), 'of', float(dws_full.abs().max()))
d = (ds_full - ds_win).abs().cpu().numpy()
print('style-gradient rows:', d.shape)
nz = np.where(d.max(axis=0) > 1e-5 * np.abs(ds_full.cpu().numpy()).max())[0]
print('columns that differ:', nz.min() if len(nz) else None, nz.max() if len(nz) else None, len(nz))
off = 0
for k, r in enumerate(eng.layer_resolutions):
    a, b = ys_full[k].cpu().numpy(), ys_win[k].cpu().numpy()
    rows = np.where(np.abs(a - b).max(axis=(0, 1, 3)) > 1e-4 * np.abs(a).max())[0]
    print(f'layer {k} res {r}: rows that differ from the whole-frame pass: {rows.min() if len(rows) else None}..{rows.max() if len(rows) else None} ({len(rows)}); equal rows '
          f'{sorted(set(range(r)) - set(rows.tolist()))[:1]}..{sorted(set(range(r)) - set(rows.tolist()))[-1:]}')

# ---- the engine's backward against autograd through the oracle network, whole frames, per precision and for a dense / row-sparse gradient
_lib.check(lib.la_synth_set_row_window(eng.handle, 0, 0), 'w')
wsr = ws.clone().requires_grad_(True)
img_r = G.synthesis(wsr, noise_mode='const')
g_dense = torch.randn(img_r.shape, generator=gen)
for name, gi in (('rows lo..hi only', g_img), ('dense', g_dense)):
    (dws_r,) = torch.autograd.grad(img_r, [wsr], gi, retain_graph=True)
    (dws_r64,) = torch.autograd.grad(G.double().synthesis(ws.double().requires_grad_(True), noise_mode='const'), [], gi.double(), allow_unused=True) if False else (None,)
    for prec in ('f32', 'f16x2', 'bf16x3'):
        e2 = SynthesisEngine.from_generator(G.float(), dev, max_batch=2, precision=prec)
        e2.forward(ws.to(dev), noise_mode='const')
        dw = e2.backward(gi.to(dev)).cpu()
        err = (dw - dws_r).abs()
        print(f'{name:18s} {prec:7s}: max |dws| {float(dws_r.abs().max()):.3f}  max err {float(err.max()):.3e}  rms err {float((err ** 2).mean().sqrt()):.3e}')
