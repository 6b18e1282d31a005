"""Two-stream non-reproducibility, forward pass only: which layer output differs first when two engines' forward passes overlap?
Engine A's stored layer outputs after an overlapped pass against the same after a solo pass."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='f16x2')
ap.add_argument('--res', type=int, default=256)
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--passes', type=int, default=6, help='forward passes per overlapped run (the last one is compared)')
a = ap.parse_args()
from latentaugment_amd import _lib                                          # noqa: E402
_lib.select_dev_build()
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.synthesis import SynthesisEngine                     # noqa: E402

dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=a.res, img_channels=2, channel_base=32768, seed=0)
w0 = synthetic.make_latents(8, seed=1).to(dev)
third = SynthesisEngine.from_generator(sd, dev, 8, precision=a.precision)
third.forward(w0.repeat(1, third.num_ws, 1), noise_mode='const'); torch.cuda.synchronize()
ea = SynthesisEngine.from_generator(sd, dev, 4, precision=a.precision)
eb = SynthesisEngine.from_generator(sd, dev, 4, precision=a.precision)
wa, wb = w0[:4].repeat(1, ea.num_ws, 1).contiguous(), w0[4:].repeat(1, eb.num_ws, 1).contiguous()
nl = len(ea.layer_resolutions)


def snap(e):
    return [e.layer_output(k, 4).clone() for k in range(nl)]


ea.forward(wa, noise_mode='const'); torch.cuda.synchronize()
solo = snap(ea)
img_solo = ea.forward(wa, noise_mode='const').clone(); torch.cuda.synchronize()
again = snap(ea)
print('solo repeat: layers identical:', all(torch.equal(x, y) for x, y in zip(solo, again)), flush=True)
eb.forward(wb, noise_mode='const'); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for rep in range(a.reps):
    with torch.cuda.stream(s2):
        for _ in range(a.passes):
            eb.forward(wb, noise_mode='const')
    with torch.cuda.stream(s1):
        for _ in range(a.passes):
            img = ea.forward(wa, noise_mode='const')
    torch.cuda.synchronize()
    cur = snap(ea)
    first = next((k for k in range(nl) if not torch.equal(cur[k], solo[k])), None)
    if first is None:
        print(f'rep {rep}: all {nl} layer outputs identical; image identical: {torch.equal(img, img_solo)}', flush=True)
        continue
    d = (cur[first] - solo[first])
    nz = (d != 0)
    per_b = nz.flatten(1).sum(1).tolist()
    per_c = nz.sum(dim=(0, 2, 3))
    ys = nz.any(dim=(0, 1, 3)).nonzero().flatten().tolist()
    xs = nz.any(dim=(0, 1, 2)).nonzero().flatten().tolist()
    print(f'rep {rep}: first differing layer {first} (res {ea.layer_resolutions[first]}), {int(nz.sum())} of {nz.numel()} elements, max |d| {float(d.abs().max()):.3e} '
          f'(max |y| {float(solo[first].abs().max()):.2e}); per sample {per_b}; channels hit {int((per_c > 0).sum())} of {per_c.numel()} '
          f'(first {per_c.nonzero().flatten()[:6].tolist()}); rows {ys[:4]}..{ys[-2:]} ({len(ys)}), cols {xs[:4]}..{xs[-2:]} ({len(xs)}); '
          f'later layers differing: {[k for k in range(first + 1, nl) if not torch.equal(cur[k], solo[k])]}', flush=True)
    idx = nz.nonzero()[:6].tolist()
    for (b_, c_, y_, x_) in idx:
        print(f'    [b {b_} c {c_} y {y_} x {x_}] solo {float(solo[first][b_, c_, y_, x_]):+.6f} overlapped {float(cur[first][b_, c_, y_, x_]):+.6f}; '
              f'solo neighbours x-1 {float(solo[first][b_, c_, y_, max(x_ - 1, 0)]):+.6f} x+1 {float(solo[first][b_, c_, y_, min(x_ + 1, solo[first].shape[3] - 1)]):+.6f}', flush=True)
