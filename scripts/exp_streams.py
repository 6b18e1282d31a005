"""Experiment: split the per-GPU batch over S HIP streams (independent samples) to overlap the small-resolution layers of one
half with the large layers of the other.  Prints ms per batch for S = 1, 2, 4."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import synthetic
from latentaugment_amd.latent_aug import LatentAug
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=256, img_channels=2, channel_base=32768, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=256, M_w=1024, M_x=256)
B = 8
w0 = synthetic.make_latents(B).to(dev)
for S in (1, 2, 4):
    b = B // S
    opt = types.SimpleNamespace(img_resolution=256, batch_size=b, modalities_aug='A,B', opt_num_epochs=20, opt_lr=0.01, truncation_psi=1.0,
                                w_pix=0.1, w_lpips=0.0, w_latent=0.001, w_disc=0.0, crop_size_aug=64, preprocess_aug='center_random_crop',
                                soft_aug=False, alpha=1.0, verbose_log=False, criterion_mode='gemm', final_noise_mode='const',
                                precision='bf16x3')
    las = [LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X}) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run():
        outs = []
        for i, (la, st) in enumerate(zip(las, streams)):
            with torch.cuda.stream(st):
                outs.append(la.run_local(w0[i * b:(i + 1) * b]))
        torch.cuda.synchronize()
        return outs
    run()
    t0 = time.time()
    for _ in range(3):
        run()
    print('streams', S, 'ms/batch', round((time.time() - t0) / 3 * 1e3, 1), flush=True)
    del las
    torch.cuda.empty_cache()
