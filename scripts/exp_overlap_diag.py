"""Diagnosis of the two-stream result corruption: one B = 4 loop on stream 1, with (a) nothing, (b) unrelated torch work (matmuls / copies)
on stream 2, (c) a second independent loop on stream 2; captured-step replay and eager launches."""
import os
import random
import sys
import time

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


def make(batch, graph):
    opt = bench.make_opt(args, 0, batch)
    opt.final_noise_mode = 'const'
    opt.hip_graph = graph
    opt.opt_num_epochs = int(os.environ.get('LA_STEPS', '20'))
    return LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X})


s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
A = torch.randn([4096, 4096], device=dev); Bm = torch.randn([4096, 4096], device=dev)
big = torch.randn([64 << 20], device=dev)
for graph in (True, False):
    if os.environ.get('LA_THIRD'):
        lf = make(8, graph); lf.run_local(w0, crop_pos=(0, 0)); torch.cuda.synchronize()
    la, lb = make(4, graph), make(4, graph)
    ref = la.run_local(w0[:4], crop_pos=(0, 0))[1].clone(); torch.cuda.synchronize()
    again = la.run_local(w0[:4], crop_pos=(0, 0))[1].clone(); torch.cuda.synchronize()
    print(f'graph={graph}: alone twice: {float((ref - again).abs().max()):.3e}', flush=True)
    lb.run_local(w0[4:], crop_pos=(0, 0)); torch.cuda.synchronize()
    for name, other in (('matmuls', lambda: [torch.mm(A, Bm) for _ in range(40)]), ('copies', lambda: [big.clone() for _ in range(40)]),
                        ('second loop', lambda: lb.run_local(w0[4:], crop_pos=(0, 0)))):
        with torch.cuda.stream(s2):
            other()
        with torch.cuda.stream(s1):
            out = la.run_local(w0[:4], crop_pos=(0, 0))[1]
        torch.cuda.synchronize()
        print(f'graph={graph}: with {name} on the other stream: max |d w| {float((ref - out).abs().max()):.3e}', flush=True)
    # both directions at once, as scripts/exp_two_streams.py does: a on s1, b on s2, each against its own solo result
    refb = lb.run_local(w0[4:], crop_pos=(0, 0))[1].clone(); torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(s1):
            if os.environ.get('LA_WAIT'):
                s1.wait_stream(torch.cuda.default_stream(dev))
            oa = la.run_local(w0[:4], crop_pos=(0, 0))[1]
        with torch.cuda.stream(s2):
            if os.environ.get('LA_WAIT'):
                s2.wait_stream(torch.cuda.default_stream(dev))
            ob = lb.run_local(w0[4:], crop_pos=(0, 0))[1]
        torch.cuda.synchronize()
        print(f'graph={graph}: two loops, rep {rep}: max |d w| a {float((ref - oa).abs().max()):.3e}  b {float((refb - ob).abs().max()):.3e}', flush=True)
