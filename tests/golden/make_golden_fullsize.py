#!/usr/bin/env python3
"""Full-size golden vectors for BASELINE.json configs B, C, D, E (and F = B with the discriminator criterion) -- made by RUNNING THE
REFERENCE in the build container.

Run from the repo root:   python tests/golden/make_golden_fullsize.py [B C D E F]
Needs /root/reference (read-only).  Never runs on the GPU box; only tests/golden/fullsize_<cfg>.npz travels.

Per config two CPU runs on identical seeded inputs (latentaugment_amd.synthetic, BASELINE.md "Synthetic inputs"):
  * ref32 -- the REFERENCE's own `LatentAug.forward` (augments/utils/util_latent_aug.py:207-310) on the reference's op
    layer, float32: the reference CPU path the north-star names.  Same harness as tests/golden/make_golden.py (our SG2
    network definition re-wired onto the reference ops, because the reference tree holds no network source).
  * o64 -- the oracle restatement of the same loop in float64: the anchor that says how far float32 itself is from the
    exact result, so that the GPU tests can state "HIP is no further from float64 than 1.5x the reference's float32 is".
Stored: the optimised latent of every sample (both runs), the latent after EVERY step of both runs (`o64_w_steps`,
`ref32_w_steps`: the drift-versus-step curve of float32 against float64), the gradient dL/dw of the first step of both runs
(`o64_grad1`, `ref32_grad1`: every criterion + G (+ D, feature net) backward at full size, before Adam's sign-like first
step hides magnitudes), the final image sub-sampled on a regular grid plus per-plane float64 moments of the whole image,
and checksums of the inputs.

How the per-step state of the REFERENCE's loop is observed without touching it: `torch.optim.Adam.step` is wrapped for
the duration of the ref32 run (harness code, this file) -- the wrapper reads the parameter's `.grad` before and the
parameter after the reference's own `optimizer.step()` (util_latent_aug.py:276).
"""
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (installs the absent-module stand-ins, imports the reference)

from latentaugment_amd import synthetic          # noqa: E402
from oracle import feature_net as fnets          # noqa: E402
from oracle import latent_aug_ref as lar         # noqa: E402
from oracle import sg2_networks as nets          # noqa: E402
from oracle import sg2_ops as our_ops            # noqa: E402

torch.set_num_threads(8)

sys.path.insert(0, os.path.dirname(HERE))
from fullsize_common import CONFIGS, CROP, CROP_SEED, LPIPS_WIDTH, build_tensors, subsample      # noqa: E402


def build_inputs(c):
    """The seeded tensors plus the oracle's network objects around them."""
    sd, meta, dsd, W, X, fea, w0 = build_tensors(c)
    G = nets.Generator(img_resolution=c['res'], img_channels=2, channel_base=c['channel_base'])
    G.load_state_dict(sd, strict=False)
    G = G.eval().requires_grad_(False)
    D = fnet = None
    if dsd is not None:
        D = nets.Discriminator(img_resolution=c['res'], img_channels=2, channel_base=c['channel_base'])
        D.load_state_dict(dsd, strict=False)
        D = D.eval().requires_grad_(False)
    if fea is not None:
        fnet = fnets.VGG16Features(seed=7, width=LPIPS_WIDTH)     # VGG16 topology at full width, random weights (vgg16.pt is a download)
    return sd, meta, G, D, fnet, W, X, fea, w0


def moments(img):
    d = img.double()
    return np.stack([d.sum(dim=(2, 3)).numpy(), d.square().sum(dim=(2, 3)).numpy()])


def run(name):
    c = dict(CONFIGS[name])
    if os.environ.get('FULLSIZE_STEPS'):      # a longer / shorter loop than fullsize_common states (the file name then says so)
        c['steps'] = int(os.environ['FULLSIZE_STEPS'])
    t0 = time.time()
    sd, meta, G, D, fnet, W, X, fea, w0 = build_inputs(c)
    random.seed(CROP_SEED)
    crop_pos = mg.ref_ud.get_params(c['res'], CROP, 'center_random_crop')['crop_pos']
    out = dict(cfg=np.array(repr(c)), crop_pos=np.array(crop_pos), w0_sum=np.array(float(w0.double().sum())),
               W_sum=np.array(float(W.double().sum())), X_sum=np.array(float(X.double().sum())))
    if fea is not None:
        out['fea_sum'] = np.array([float(f.double().sum()) for f in fea])
    kw = dict(w_latent=c['w_latent'], w_pix=c['w_pix'], w_disc=c['w_disc'], w_lpips=c['w_lpips'])

    # ---- ref32: the reference's loop on the reference's ops
    nets.ops = mg._RefOpsAdapter
    try:
        m = mg.build_ref_module(G, D, W, X, fea, fnet, res=c['res'], batch=c['batch'], epochs=c['steps'], lr=0.01, crop=CROP, **kw)
        random.seed(CROP_SEED)
        torch.manual_seed(123)
        steps32, grads32 = [], []
        adam_step = torch.optim.Adam.step

        def spy(self, *a, **k):
            p = self.param_groups[0]['params'][0]
            grads32.append(p.grad.detach().clone())
            r = adam_step(self, *a, **k)
            steps32.append(p.detach().clone())
            return r
        torch.optim.Adam.step = spy
        try:
            img, w_aug = m.forward(w0.clone(), [f'f{i}' for i in range(c['batch'])])
        finally:
            torch.optim.Adam.step = adam_step
    finally:
        nets.ops = our_ops
    assert len(steps32) == c['steps'] and steps32[0].shape == w0.shape
    out['ref32_w_steps'] = torch.stack([t[:, 0] for t in steps32]).numpy()
    out['ref32_grad1'] = grads32[0][:, 0].numpy()
    assert float((steps32[-1][:, 0] - w_aug[:, 0]).abs().max()) == 0.0
    out['ref32_w'] = mg.T(w_aug[:, 0])
    assert float((w_aug - w_aug[:, :1]).abs().max()) == 0.0
    out['ref32_img_sub'] = mg.T(subsample(img, c['res']))
    out['ref32_img_mom'] = moments(img.detach())
    print(name, 'ref32 done', f'{time.time() - t0:.0f}s', 'max|dw|', float((w_aug[:, 0] - w0[:, 0]).abs().max()), flush=True)
    del m

    # ---- o64: the oracle restatement in float64
    nets.COMPUTE_DTYPE = torch.float64
    try:
        G.double()
        if D is not None:
            D.double()
        if fnet is not None:
            fnet.double()
        ref = lar.LatentAugRef(G, D, W=W.double(), X=X.double(), fea=[f.double() for f in fea] if fea is not None else None,
                               feature_net=fnet, res=c['res'], num_epochs=c['steps'], opt_lr=0.01, crop_size=CROP,
                               final_noise_mode='const', dtype=torch.float64, fused_modconv=False, **kw)
        # (non-fused modulated conv: equal to the fused form in exact arithmetic, and the float64 CPU convolution only
        #  parallelises over the batch dimension, which the fused form collapses to 1)
        img64, w64 = ref.forward(w0, crop_pos=tuple(crop_pos), record=True)
    finally:
        nets.COMPUTE_DTYPE = torch.float32
        G.float()
    out['o64_w'] = w64[:, 0].numpy()
    out['o64_w_steps'] = torch.stack([t[:, 0] for t in ref.trace['w']]).to(torch.float32).numpy()
    out['o64_grad1'] = ref.trace['grad'][0][:, 0].numpy()
    for k in ('loss_latent', 'loss_pix', 'loss_disc', 'loss_lpips'):
        out['o64_' + k] = np.array(ref.trace[k])
    out['o64_img_sub'] = subsample(img64, c['res']).numpy()
    out['o64_img_mom'] = moments(img64)
    e32 = np.abs(out['ref32_w'].astype(np.float64) - out['o64_w'])
    print(name, 'o64 done', f'{time.time() - t0:.0f}s', 'ref32-vs-o64 latent err max', e32.max(), 'rms', np.sqrt((e32 ** 2).mean()), flush=True)
    fname = f'fullsize_{name}.npz' if not os.environ.get('FULLSIZE_STEPS') else f'fullsize_{name}_{c["steps"]}steps.npz'
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(name, 'saved', fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, 'KB', flush=True)


if __name__ == '__main__':
    for n in (sys.argv[1:] or ['B', 'C', 'D', 'E', 'F']):
        run(n)
