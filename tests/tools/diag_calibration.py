"""Operand-scale calibration on the full-size seeded workloads: the calibration statistic and the step-1 gradient error of both
scale modes against the float64 fixture (tests/golden/fullsize_<cfg>.npz)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fullsize_common import CONFIGS, CROP, CROP_SEED, build_tensors      # noqa: E402
from test_hip_synthesis import _opt                                     # noqa: E402
from latentaugment_amd.latent_aug import LatentAug, get_params          # noqa: E402

dev = torch.device('cuda', 0)
for name in sys.argv[1:] or ['B']:
    c = CONFIGS[name]
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', f'fullsize_{name}.npz'))
    sd, meta, dsd, W, X, fea, w0 = build_tensors(c)
    for mode in ('bound', 'data', 'auto'):
        opt = _opt(img_resolution=c['res'], batch_size=c['batch'], opt_num_epochs=1, opt_lr=0.01, crop_size_aug=CROP,
                   w_latent=c['w_latent'], w_pix=c['w_pix'], w_disc=0.0, w_lpips=0.0, final_noise_mode='const',
                   criterion_mode='gemm', precision='f16x2', operand_scale=mode)
        la = LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X})
        random.seed(CROP_SEED)
        pos = get_params(c['res'], CROP)['crop_pos']
        trace = {'want': ('w', 'grad')}
        la.run_local(w0.to(dev), want_losses=True, crop_pos=pos, trace=trace)
        g = trace['grad'].cpu().numpy()[0].astype(np.float64)
        line = f'{name} {mode}: chosen={la.engine.operand_scale} calibration={getattr(la.engine, "calibration", None)}'
        if c['w_disc'] == 0 and c['w_lpips'] == 0 and 'o64_grad1' in fx.files:
            g64 = fx['o64_grad1']; gr = fx['ref32_grad1'].astype(np.float64)
            line += f' rms_err={np.sqrt(((g - g64) ** 2).mean()):.3e} ref_rms_err={np.sqrt(((gr - g64) ** 2).mean()):.3e} gmax={np.abs(g64).max():.3e}'
        print(line, flush=True)
        del la
