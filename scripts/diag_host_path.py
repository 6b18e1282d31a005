"""Host-side cost per batch of the plugin protocol at the global batch of an N-rank run (every rank handles the whole gathered batch on
the host): get_output's device-to-host copy (page-locked staging against pageable), set_input, the latent lookup, the final-noise draw."""
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda', 0)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0) / n


for gb in (8, 16, 32, 64):
    x = torch.randn([gb, 2, 256, 256], device=dev)

    def pageable():
        return x.cpu()

    def pinned():
        h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
        h.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
        return h

    a, b = torch.randn([gb, 1, 256, 256]), torch.randn([gb, 1, 256, 256])
    print(f'global batch {gb}: D2H pageable {t(pageable):.2f} ms, page-locked {t(pinned):.2f} ms, host cat of the input pair {t(lambda: torch.cat([a, b], 1)):.2f} ms',
          flush=True)

from latentaugment_amd import synthetic
from latentaugment_amd.synthesis import SynthesisEngine
sd, meta = synthetic.make_generator_state_dict(img_resolution=256, img_channels=2, channel_base=32768, seed=0)
for k in list(sd):
    if k.endswith('noise_strength'):
        sd[k] = torch.ones_like(sd[k])
eng = SynthesisEngine.from_generator(sd, dev, 8, precision='f16x2')
for gb in (8, 64):
    print(f'final noise, global batch {gb}, rows 0..8: {t(lambda: eng.make_noises(gb, batch_seed=5, rows=(0, 8))):.2f} ms', flush=True)
