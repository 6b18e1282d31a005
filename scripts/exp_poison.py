"""Do results depend on memory the loop never wrote?  Before any handle exists, most of the free device memory is filled with a byte
pattern and handed back to torch's caching allocator, so that every later allocation (workspaces, outputs) is carved out of it: memory
a kernel reads without having written it -- padding inside a workspace, bytes past the end of a tensor -- then holds that pattern
instead of the zeros of a fresh hipMalloc.  Patterns: 0x00 (the fresh-process case), 0xFF (NaN as fp32 and as fp16), 0x4B (a large finite
number).  The final latent of a seeded B = 4 loop must not depend on it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                # noqa: E402
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.latent_aug import LatentAug                          # noqa: E402

dev = torch.device('cuda', 0)
pattern = int(sys.argv[1], 0)
preset = sys.argv[2] if len(sys.argv) > 2 else 'B'
out = sys.argv[3]
gb = int(os.environ.get('LA_POISON_GB', '24'))
junk = [torch.full([1 << 30], pattern, dtype=torch.uint8, device=dev) for _ in range(gb)]
torch.cuda.synchronize()
del junk                      # back to the caching allocator, contents intact
sys.argv = ['bench.py'] + (['--preset', preset] if preset != 'B' else [])
args = bench.apply_preset(bench.parse())
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
w0 = synthetic.make_latents(8, seed=1).to(dev)
opt = bench.make_opt(args, 0, 4)
opt.final_noise_mode = 'const'
inject = dict(generator=sd, banks={'W': W, 'X': X})
if args.w_disc > 0:
    inject['discriminator'] = synthetic.make_discriminator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base)
if args.w_lpips > 0:
    inject['feature_net'] = synthetic.make_vgg16_lpips_ops(seed=7)
    F = synthetic.lpips_num_features(64)
    gen = torch.Generator(device=dev).manual_seed(5)
    inject['banks']['fea'] = [torch.randn([args.M_x, F], device=dev, generator=gen) * (1.0 / F) ** 0.5 for _ in range(2)]
la = LatentAug('train', opt, '/tmp', [0], **inject)
img, w, _ = la.run_local(w0[:4], crop_pos=(0, 0))
torch.cuda.synchronize()
print(f'pattern {pattern:#04x}: reserved {torch.cuda.memory_reserved() >> 30} GiB, finite: {bool(torch.isfinite(w).all() and torch.isfinite(img).all())}', flush=True)
torch.save({'w': w.cpu(), 'img': img.cpu()}, out)
