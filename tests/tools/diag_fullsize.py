"""Dev tool: latent / image error of a full-size parity case against the float64 fixture, per contraction precision."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_hip_fullsize as T
dev = torch.device('cuda', 0)
name = sys.argv[1]
for prec in sys.argv[2:]:
    c, fx, w0, w, isub, img, losses = T._run_case(name, dev, precision=prec)
    hm, hr = T._err(w, fx['o64_w']); rm, rr = T._err(fx['ref32_w'], fx['o64_w'])
    im, ir = T._err(isub, fx['o64_img_sub']); jm, jr = T._err(fx['ref32_img_sub'], fx['o64_img_sub'])
    print(f'{name} {prec}: latent HIP max {hm:.3e} rms {hr:.3e} | ref max {rm:.3e} rms {rr:.3e} | ratio rms {hr / rr:.2f};  image HIP rms {ir:.3e} ref rms {jr:.3e} ratio {ir / jr:.2f}', flush=True)
    for k, col in (('loss_latent', 0), ('loss_pix', 1), ('loss_disc', 2), ('loss_lpips', 3)):
        ref = fx['o64_' + k]
        if np.abs(ref).max() > 0:
            print('   ', k, 'max rel err', float(np.abs(losses[:, col] / ref - 1).max()))
