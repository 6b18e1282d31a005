"""Bisecting the two-stream non-reproducibility: two independent B = 4 loops overlapped on two streams, each against its own solo
result, under a configuration given on the command line (precision, resolution, steps, criteria, graph / eager, third handle alive, dev-build
environment switches such as LA_NO_XS_HANDOFF / LA_NO_SEAM_FUSE).  Prints one line: the largest latent difference of either loop over the reps."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='f16x2')
ap.add_argument('--res', type=int, default=256)
ap.add_argument('--channel-base', type=int, default=32768)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--reps', type=int, default=4)
ap.add_argument('--no-graph', action='store_true')
ap.add_argument('--no-third', action='store_true')
ap.add_argument('--w-pix', type=float, default=0.1)
ap.add_argument('--fwd-only', action='store_true', help='compare final images of plain synthesis passes instead of loops')
ap.add_argument('--tag', default='')
a = ap.parse_args()
from latentaugment_amd import _lib                                          # noqa: E402
_lib.select_dev_build()
import bench                                                                # noqa: E402
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.latent_aug import LatentAug                          # noqa: E402

dev = torch.device('cuda', 0)
sys.argv = ['bench.py', '--precision', a.precision, '--res', str(a.res), '--channel-base', str(a.channel_base)]
args = bench.apply_preset(bench.parse())
args.w_pix = a.w_pix
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
w0 = synthetic.make_latents(8, seed=1).to(dev)


def make(batch):
    opt = bench.make_opt(args, 0, batch)
    opt.final_noise_mode = 'const'
    opt.hip_graph = not a.no_graph
    opt.opt_num_epochs = a.steps
    return LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X})


def run(la, w):
    if a.fwd_only:
        ws = w.repeat(1, la.num_ws, 1)
        out = None
        for _ in range(max(a.steps, 1)):
            out = la.engine.forward(ws, noise_mode='const').clone()
        return out
    return la.run_local(w, crop_pos=(0, 0))[1]


s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
if not a.no_third:
    lf = make(8); run(lf, w0); torch.cuda.synchronize()
la, lb = make(4), make(4)
ra = run(la, w0[:4]).clone(); torch.cuda.synchronize()
rb = run(lb, w0[4:]).clone(); torch.cuda.synchronize()
solo = float((run(la, w0[:4]) - ra).abs().max()); torch.cuda.synchronize()
worst = 0.0
for rep in range(a.reps):
    with torch.cuda.stream(s1):
        oa = run(la, w0[:4])
    with torch.cuda.stream(s2):
        ob = run(lb, w0[4:])
    torch.cuda.synchronize()
    worst = max(worst, float((ra - oa).abs().max()), float((rb - ob).abs().max()))
env = ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('LA_'))
print(f'[{a.tag or "cfg"}] prec={a.precision} res={a.res} steps={a.steps} graph={not a.no_graph} third={not a.no_third} fwd_only={a.fwd_only} w_pix={a.w_pix} {env}: '
      f'solo repeat {solo:.1e}, overlapped worst {worst:.3e}', flush=True)
