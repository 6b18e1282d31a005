"""Dev tool: graph replay vs eager, per criterion and per run index (which runs differ, by how much)."""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from test_hip_synthesis import _opt, tiny_G, tiny_D
from oracle import feature_net as fnets
from oracle import latent_aug_ref as lar
from latentaugment_amd.latent_aug import LatentAug
gl = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'latent_loop.npz'))
dev = torch.device('cuda', 0)
G, D = tiny_G(gl), tiny_D(gl)
fnet = fnets.TinyFeatureNet(seed=5, crop=8)
banks = {'W': torch.tensor(gl['W']), 'X': torch.tensor(gl['X']), 'fea': [torch.tensor(gl['fea0']), torch.tensor(gl['fea1'])]}
w = torch.tensor(gl['w0']).to(dev)
kw = dict(w_lpips=3.0)
extra = dict(discriminator=D, feature_net=fnets.tiny_ops(fnet))
pos = tuple(int(v) for v in gl['lpips_crop_pos'])
res = {}
for tag, mode, wl in (('graph', True, False), ('eager', False, False), ('eager2', False, False), ('eager+losses', False, True), ('graph+losses', True, True)):
    la = LatentAug('train', _opt(batch_size=2, hip_graph=mode, precision='f32', **kw), '/tmp', [0], generator=G, banks=banks, **extra)
    outs = []
    for _ in range(3):
        img, w_aug, _ = la.run_local(w, crop_pos=pos, want_losses=wl)
        torch.cuda.synchronize()
        outs.append(w_aug.clone())
    res[tag] = outs
    print(tag, 'vs golden', ['%.3g' % float((o.cpu() - torch.tensor(gl['lpips_w_aug'])).abs().max()) for o in outs], 'err:', la._lib.la_last_error(), flush=True)
