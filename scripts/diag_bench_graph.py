"""Dev tool: per-batch wall time of the bench workload in graph mode (hang diagnosis: faulthandler dumps after 50 s)."""
import faulthandler, os, sys, time, types
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentaugment_amd import synthetic
from latentaugment_amd.latent_aug import LatentAug
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda', 0)
sd, meta = synthetic.make_generator_state_dict(img_resolution=res, img_channels=2, channel_base=32768, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=res, M_w=1024, M_x=256)
opt = types.SimpleNamespace(img_resolution=res, batch_size=B, modalities_aug='A,B', opt_num_epochs=steps, opt_lr=0.01, truncation_psi=1.0,
                            w_pix=0.1, w_lpips=0.0, w_latent=0.001, w_disc=0.0, crop_size_aug=64, preprocess_aug='center_random_crop',
                            soft_aug=False, alpha=1.0, verbose_log=False, criterion_mode='gemm', final_noise_mode='random', precision='f16x2',
                            hip_graph=True)
la = LatentAug('train', opt, '/tmp', [0], generator=sd, banks={'W': W, 'X': X})
w0 = synthetic.make_latents(B).to(dev)
print('built', flush=True)
for i in range(4):
    t0 = time.time()
    img, w_aug, _ = la.run_local(w0)
    print('enqueued', i, f'{time.time() - t0:.3f}s', flush=True)
    torch.cuda.synchronize()
    print('batch', i, f'{time.time() - t0:.3f}s', 'err:', la._lib.la_last_error(), float(w_aug.abs().max()), flush=True)
