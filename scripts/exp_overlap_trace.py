"""Where does a loop first differ when a second, independent loop overlaps it (third handle alive: scripts/exp_overlap_diag.py)?
Per-step traces (latent, image, gradient) of the overlapped run against the solo run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                # noqa: E402
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.latent_aug import LatentAug                          # noqa: E402

dev = torch.device('cuda', 0)
sys.argv = ['bench.py']
args = bench.apply_preset(bench.parse())
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
w0 = synthetic.make_latents(8, seed=1).to(dev)


def make(batch):
    opt = bench.make_opt(args, 0, batch)
    opt.final_noise_mode = 'const'
    opt.opt_num_epochs = 20
    return LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X})


lf = make(8); lf.run_local(w0, crop_pos=(0, 0)); torch.cuda.synchronize()
la, lb = make(4), make(4)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
want = {'want': ('w', 'grad')}
t0 = dict(want); la.run_local(w0[:4], crop_pos=(0, 0), trace=t0); torch.cuda.synchronize()
lb.run_local(w0[4:], crop_pos=(0, 0)); torch.cuda.synchronize()
for rep in range(3):
    t1 = dict(want)
    with torch.cuda.stream(s2):
        lb.run_local(w0[4:], crop_pos=(0, 0))
    with torch.cuda.stream(s1):
        la.run_local(w0[:4], crop_pos=(0, 0), trace=t1)
    torch.cuda.synchronize()
    for s in range(20):
        dg = float((t0['grad'][s] - t1['grad'][s]).abs().max()); dw = float((t0['w'][s] - t1['w'][s]).abs().max())
        if dg > 0 or dw > 0 or s == 19:
            gm = float(t0['grad'][s].abs().max())
            nb = (t0['grad'][s] != t1['grad'][s]).flatten(1).sum(1).tolist()
            print(f'rep {rep} step {s + 1}: grad max diff {dg:.2e} (max |g| {gm:.2e}; differing entries per sample {nb}); latent after the step {dw:.2e}', flush=True)
            if dg > 0:
                break
