"""Seeded inputs of the full-size parity cases (BASELINE.json configs B-E), shared by the fixture generator
(tests/golden/make_golden_fullsize.py, build container, runs the reference) and the GPU tests (tests/test_hip_fullsize.py).
Host-side tensor construction only; nothing here imports the oracle or the reference."""
from latentaugment_amd import synthetic

# criterion weights of config E: the authors' values (backbone_latentaug.py:46-54); banks at Pelvis scale (latent_aug.py:78-79)
CONFIGS = {
    'B': dict(res=256, channel_base=32768, batch=8, steps=20, M_w=1024, M_x=256, w_latent=0.001, w_pix=0.1, w_disc=0.0, w_lpips=0.0),
    'C': dict(res=512, channel_base=32768, batch=4, steps=20, M_w=1024, M_x=256, w_latent=0.001, w_pix=0.1, w_disc=0.0, w_lpips=0.0),
    'D': dict(res=1024, channel_base=32768, batch=2, steps=50, M_w=1024, M_x=256, w_latent=0.001, w_pix=0.1, w_disc=0.0, w_lpips=0.0),
    'E': dict(res=256, channel_base=16384, batch=8, steps=20, M_w=6026, M_x=1572, w_latent=0.001, w_pix=0.1, w_disc=0.01, w_lpips=10.0),
    # SURVEY 8(d)'s second run: the bench workload (config B) with the discriminator criterion on -- D at config-f width, 256^2
    'F': dict(res=256, channel_base=32768, batch=8, steps=20, M_w=1024, M_x=256, w_latent=0.001, w_pix=0.1, w_disc=0.01, w_lpips=0.0),
}
CROP = 64          # crop_size_aug
CROP_SEED = 6      # python `random` seed of the crop position
LPIPS_WIDTH = 64   # VGG16 at full width


def build_tensors(c):
    """(G state_dict, meta, D state_dict | None, W, X, fea | None, w0) from seeds only."""
    sd, meta = synthetic.make_generator_state_dict(img_resolution=c['res'], img_channels=2, channel_base=c['channel_base'], seed=0)
    W, X = synthetic.make_banks(meta['num_ws'], res=c['res'], M_w=c['M_w'], M_x=c['M_x'])
    w0 = synthetic.make_latents(c['batch'])
    dsd = fea = None
    if c['w_disc'] > 0:
        dsd = synthetic.make_discriminator_state_dict(img_resolution=c['res'], img_channels=2, channel_base=c['channel_base'])
    if c['w_lpips'] > 0:
        fea = synthetic.make_feature_banks(c['M_x'], synthetic.lpips_num_features(CROP, LPIPS_WIDTH))
    return sd, meta, dsd, W, X, fea, w0


def subsample(img, res):
    """64 x 64 regular grid of an image batch [B,C,res,res] (what the fixtures store of the final image)."""
    st = res // 64
    return img[:, :, st // 2::st, st // 2::st]
