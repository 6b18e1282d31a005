import sys, os, io, contextlib, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.getcwd())
import bench
from latentaugment_amd import synthetic
from latentaugment_amd.latent_aug import LatentAug
sys.argv = ['bench.py', '--preset', 'E']
args = bench.apply_preset(bench.parse())
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
F = synthetic.lpips_num_features(64)
gen = torch.Generator(device=dev).manual_seed(5)
banks = {'W': W, 'X': X, 'fea': [torch.randn([args.M_x, F], device=dev, generator=gen) * (1.0 / F) ** 0.5 for _ in range(2)]}
opt = bench.make_opt(args, 0, 8); opt.verbose_log = True; opt.final_noise_mode = 'const'
la = LatentAug('train', opt, '/tmp/e_times', [0], generator=sd, banks=banks, feature_net=synthetic.make_vgg16_lpips_ops(seed=7),
               discriminator=synthetic.make_discriminator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base))
w0 = synthetic.make_latents(8, seed=1).to(dev)
with contextlib.redirect_stdout(io.StringIO()):
    la.forward(w0, [f'f{i}' for i in range(8)])
t = la.stats_time
keys = list(next(iter(t.values())).keys())
print({k: round(1e3 * float(np.mean([v[k] for v in t.values()])), 3) for k in keys}, 'ms per epoch (device, serial brackets incl. loss scalars)')
