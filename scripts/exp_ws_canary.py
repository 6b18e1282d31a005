"""Does any kernel write past the end (or before the start) of a handle's workspace?  Every uint8 workspace tensor gets 1 MB of a known
pattern on both sides; one B = 4 loop runs; the guards are checked."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                # noqa: E402
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.latent_aug import LatentAug                          # noqa: E402

GUARD = 1 << 20
guards = []
_empty = torch.empty


def guarded_empty(*a, **k):
    if k.get('dtype') is torch.uint8 and 'device' in k and torch.device(k['device']).type == 'cuda':
        n = int(a[0][0])
        t = _empty([n + 2 * GUARD], **k)
        t.fill_(0xAB)
        guards.append((t, n))
        return t[GUARD:GUARD + n]
    return _empty(*a, **k)


torch.empty = guarded_empty
dev = torch.device('cuda', 0)
preset = sys.argv[1] if len(sys.argv) > 1 else 'B'
sys.argv = ['bench.py'] + (['--preset', preset] if preset != 'B' else [])
args = bench.apply_preset(bench.parse())
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
w0 = synthetic.make_latents(8, seed=1).to(dev)
for batch in (8, 4, 3):
    opt = bench.make_opt(args, 0, batch)
    opt.final_noise_mode = 'const'
    inject = dict(generator=sd, banks={'W': W, 'X': X})
    if args.w_disc > 0:
        inject['discriminator'] = synthetic.make_discriminator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base)
    la = LatentAug('train', opt, '/tmp', [0], **inject)
    la.run_local(w0[:batch], crop_pos=(0, 0))
    torch.cuda.synchronize()
    for t, n in guards:
        lo = int((t[:GUARD] != 0xAB).sum()); hi = int((t[GUARD + n:] != 0xAB).sum())
        print(f'batch {batch}: workspace of {n} bytes: {lo} bytes changed before it, {hi} after it' + ('' if lo + hi == 0 else '   <-- OUT OF BOUNDS WRITE'),
              flush=True)
    guards.clear()
    del la
