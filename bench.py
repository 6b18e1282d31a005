#!/usr/bin/env python3
"""bench.py -- augmented images/s of the latent-optimisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one batch through the plugin API (set_input -> forward -> get_output): B images, `--latent-steps` Adam steps
of StyleGAN2 synthesis forward + backward-to-w each, then the final synthesis (reference loop: backbone_latentaug.py:99-106).
Workload (BASELINE.json configs[1]): SG2 config-f 256x256, 2 channels, random-init G, B = 8 per GPU, 20 latent steps,
criteria w_latent=0.001 / w_pix=0.1 (banks M_w=1024, M_x=256, scanned every step), synthetic inputs.
Inputs (latents, banks, weights) are resident in HBM when the timed region starts; the image D2H copy of get_output()
is inside the timed region, as it is in the reference's loop.
Weak scaling: every rank optimises its own B images; one RCCL all_gather of the augmented batch per step.
"""
import argparse
import json
import os
import random
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
# fp32-equivalent peak of each contraction mode: one fp32 product = 1 fp32 MFMA, or 6 / 3 bf16 MFMAs of the split scheme
CONTRACTION = {
    'f32': dict(peak=MFMA_F32_PEAK_TFLOPS, kernel='la_conv_igemm_kernel (fp32 MFMA 32x32x2, exact fp32)', mfma_per_product=1),
    'bf16x3': dict(peak=MFMA_BF16_PEAK_TFLOPS / 6, kernel='la_conv_bf16_halo_kernel / la_conv_bf16_kernel <FMT_BF16X3> (fp32 split into 3 bf16 terms, 6 bf16 '
                   'MFMA 32x32x16 per product, fp32 accumulate; fp32-class error)', mfma_per_product=6),
    'f16x2': dict(peak=MFMA_BF16_PEAK_TFLOPS / 3, kernel='la_conv_bf16_halo_kernel / la_conv_bf16_kernel <FMT_F16X2> (fp32 scaled by powers of two and split into '
                  '2 fp16 terms, 3 fp16 MFMA 32x32x16 per product, fp32 accumulate; fp32-class error)', mfma_per_product=3),
    'bf16x2': dict(peak=MFMA_BF16_PEAK_TFLOPS / 3, kernel='la_conv_bf16_kernel<NTERM=2> (2 bf16 terms, 3 bf16 MFMA per product; '
                   'approximate mode)', mfma_per_product=3),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=4)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    p.add_argument('--latent-steps', type=int, default=20)
    p.add_argument('--res', type=int, default=256)
    p.add_argument('--channel-base', type=int, default=32768, help='32768 = config-f, 16384 = config-e')
    p.add_argument('--criterion-mode', default='gemm', choices=['gemm', 'collapsed'])
    p.add_argument('--precision', default='f16x2', choices=['f32', 'f16x2', 'bf16x3', 'bf16x2'],
                   help='contraction arithmetic: exact fp32 MFMA, or fp32 operands split into 2 scaled fp16 / 3 bf16 / 2 bf16 terms '
                        'on the 16-bit MFMA (fp32 accumulate); f16x2 and bf16x3 have fp32-class error')
    p.add_argument('--w-disc', type=float, default=0.0, help='discriminator criterion weight (BASELINE.md second run: 0.01)')
    p.add_argument('--preset', default='B', choices=['B', 'E'],
                   help="B: BASELINE.json configs[1] (the metric's config). E: configs[4] per-GPU shape -- config-e 256^2, all four "
                        "criteria at the authors' weights (w_lpips 10, w_pix 0.1, w_latent 0.001, w_disc 0.01), Pelvis-scale banks")
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true', help='skip the HIP-event leg (used under rocprofv3)')
    p.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                   help="'gloo' + --force-device rehearses the multi-rank path on a 1-GPU box (collective staged through host)")
    p.add_argument('--force-device', type=int, default=-1, help='rehearsal only: every rank uses this device index')
    return p.parse_args()


def apply_preset(args):
    args.w_pix, args.w_latent, args.w_lpips, args.M_w, args.M_x = 0.1, 0.001, 0.0, 1024, 256
    if args.preset == 'E':
        args.channel_base = 16384
        args.w_lpips, args.w_disc, args.M_w, args.M_x = 10.0, 0.01, 6026, 1572     # backbone_latentaug.py:46-54, SURVEY 8d
    return args


def make_opt(args, local_rank):
    return types.SimpleNamespace(
        aug='latent', gpu_ids=[local_rank], gpu_ids_aug=str(local_rank), checkpoints_dir='/tmp', name='bench', phase='train',
        img_resolution=args.res, batch_size=args.batch, modalities_aug='A,B', opt_num_epochs=args.latent_steps, opt_lr=0.01,
        truncation_psi=1.0, w_pix=args.w_pix, w_lpips=args.w_lpips, w_latent=args.w_latent, w_disc=args.w_disc, crop_size_aug=64,
        preprocess_aug='center_random_crop', soft_aug=False, alpha=1.0, verbose_log=False, rand_aug=False,
        lower_bound_clip=False, p_thres=0.0, init_w='inv', criterion_mode=args.criterion_mode, final_noise_mode='random',
        precision=args.precision)


def cpu_baseline(sd, meta, args):
    """The oracle (a CPU port pinned to reference goldens) on the host cores, bounded sample, scaled to images/s."""
    import torch
    from oracle import latent_aug_ref as lar
    from oracle import sg2_networks as nets
    from latentaugment_amd import synthetic
    # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribes
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    G = nets.Generator(z_dim=meta['w_dim'], w_dim=meta['w_dim'], img_resolution=args.res, img_channels=2,
                       channel_base=args.channel_base)
    G.load_state_dict(sd, strict=False)
    G = G.eval().requires_grad_(False)
    b = 4
    W, X = synthetic.make_banks(G.num_ws, res=args.res, M_w=1024, M_x=256)
    nstep = 4
    ref = lar.LatentAugRef(G, None, W=W, X=X, res=args.res, num_epochs=nstep, opt_lr=0.01, crop_size=64, w_latent=0.001,
                           w_pix=0.1)
    w0 = synthetic.make_latents(b)
    with torch.no_grad():
        G.synthesis(w0.repeat(1, G.num_ws, 1), noise_mode='const')      # warm the allocator / thread pool
        t0 = time.time()
        G.synthesis(w0.repeat(1, G.num_ws, 1), noise_mode='const')
        t_fwd = time.time() - t0
    t0 = time.time()
    ref.forward(w0, crop_pos=(0, 0))          # nstep optimisation steps (fwd + bwd + Adam) + the final synthesis
    t_loop = time.time() - t0
    t_step = max(t_loop - t_fwd, 1e-9) / nstep
    per_batch = args.latent_steps * t_step + t_fwd
    return {'value': b / per_batch, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle (CPU restatement pinned to reference goldens), same G/banks, B={b}: {nstep} latent steps + final '
                      f'synthesis timed ({t_loop:.1f}s, of which final {t_fwd:.1f}s), extrapolated to {args.latent_steps} steps'}


def main():
    args = apply_preset(parse())
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run')
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group('gloo')

    from latentaugment_amd import _lib, synthetic
    from latentaugment_amd.augments import create_augment
    from latentaugment_amd.latent_aug import InMemoryLatentCodes

    sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base,
                                                   seed=0)
    W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
    data = synthetic.make_batch(args.batch, res=args.res, seed=2 + rank)
    w0 = synthetic.make_latents(args.batch, seed=1 + rank)
    codes = InMemoryLatentCodes({p: w0[i, 0].numpy() for i, p in enumerate(data['A_paths'])})
    opt = make_opt(args, local_rank)
    # each rank owns its own B images (weak scaling); the plugin itself is run un-sharded per rank, the gather of the
    # whole job's output is done below with the same single collective the sharded plugin path uses
    opt.inject = dict(generator=sd, banks={'W': W, 'X': X}, latent_codes=codes, group=None)
    if args.w_lpips > 0:
        # VGG16-shaped feature net with random weights (the real vgg16.pt is a download); feature banks drawn on the device
        opt.inject['feature_net'] = synthetic.make_vgg16_lpips_ops(seed=7)
        F = synthetic.lpips_num_features(64)
        gen = torch.Generator(device=dev).manual_seed(5)
        opt.inject['banks']['fea'] = [torch.randn([args.M_x, F], device=dev, generator=gen) * (1.0 / F) ** 0.5 for _ in range(2)]
    if args.w_disc > 0:
        opt.inject['discriminator'] = synthetic.make_discriminator_state_dict(img_resolution=args.res, img_channels=2,
                                                                              channel_base=args.channel_base)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        aug = create_augment(opt)
    la = aug.latent_aug
    random.seed(6)

    def one_step():
        aug.set_input(data)
        # plugin forward without the process-group sharding (each rank has distinct samples)
        aug.w_AB = aug.sample_from_inversion(aug.fname).to(dev)
        img, w_aug, _ = la.run_local(aug.w_AB)
        if world > 1:
            flat = torch.cat([img.reshape(args.batch, -1), w_aug.reshape(args.batch, -1)], dim=1)
            if args.dist_backend == 'nccl':
                out = torch.empty([world * args.batch, flat.shape[1]], device=dev)
                dist.all_gather_into_tensor(out, flat)      # ONE RCCL collective per batch over xGMI
            else:                                           # rehearsal path (gloo): same collective, staged through host
                hflat = flat.cpu()
                out = torch.empty([world * args.batch, hflat.shape[1]])
                dist.all_gather_into_tensor(out, hflat)
        aug.real_AB_aug, aug.w_AB_aug = img, w_aug
        return aug.get_output()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    lib = _lib.load()
    prof = (not args.no_roofline)
    barrier()
    if prof:
        # HIP events around a hashed 1-in-8 sample of the contraction launches (bracketing all ~900 per batch costs ~3 % of the
        # timed region); the averages below are over the sampled launches, 'launches' is the total
        _lib.check(lib.la_prof_set_stride(8), 'la_prof_set_stride')
        _lib.check(lib.la_prof_begin(), 'la_prof_begin')
    t0 = time.time()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    elapsed = time.time() - t0
    roof = None
    if prof:
        import ctypes as C
        ms, n, fl, by = C.c_double(), C.c_long(), C.c_double(), C.c_double()
        rc = lib.la_prof_end(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        total_launches = max(int(lib.la_prof_total_launches()), 1)
        if rc == 0 and ms.value > 0:
            tf = fl.value / (ms.value * 1e-3) / 1e12
            cm = CONTRACTION[args.precision]
            roof = {'bound': 'mfma', 'kernel': cm['kernel'] + ', all contraction launches',
                    'achieved': tf, 'peak': cm['peak'], 'unit': 'TFLOP/s', 'frac': tf / cm['peak'],
                    'peak_note': 'fp32-equivalent: dense MFMA peak of the instruction used / MFMAs issued per fp32 product '
                                 f"({cm['mfma_per_product']}); executed MFMA rate = {tf * cm['mfma_per_product']:.0f} TFLOP/s",
                    'traffic': None, 'launches': int(total_launches), 'sampled_launches': n.value,
                    'avg_launch_ms': ms.value / max(n.value, 1),
                    'kernel_time_frac_of_wall': ms.value * 1e-3 * (total_launches / max(n.value, 1)) / elapsed,
                    'algorithmic_gbs': by.value / (ms.value * 1e-3) / 1e9, 'hbm_peak_gbs': HBM_PEAK_GBS}
            # HBM bytes per launch from the committed PMC passes of this exact workload (scripts/make_profiles.sh)
            for pmc in (os.path.join(ROOT, 'profiles', f'r01_d_pmc_traffic_{args.precision}.json'),
                        os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')):
                if not os.path.isfile(pmc):
                    continue
                pj = json.load(open(pmc))
                if pj.get('precision') == args.precision and args.w_disc == 0 and args.preset == 'B':
                    roof['traffic'] = pj.get('bytes_per_launch')
                    roof['traffic_note'] = pj.get('note')
                    break
            roof['algorithmic_bytes_per_launch'] = by.value / max(n.value, 1)
    assert out['A'].shape == (args.batch, 1, args.res, args.res)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    images = args.steps * args.batch * world
    line = {
        'metric': 'augmented images/sec (256^2, 20 latent steps)', 'value': images / elapsed, 'unit': 'images/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f16x2': 'f32 (scaled split-fp16x2 on fp16 MFMA, fp32 accumulate)',
                  'bf16x3': 'f32 (split-bf16x3 on bf16 MFMA, fp32 accumulate)',
                  'bf16x2': 'f32 (split-bf16x2 on bf16 MFMA, approximate)'}[args.precision], 'data': 'synthetic',
        'config': {'workload': f'SG2 config-{"f" if args.channel_base == 32768 else "e"} {args.res}x{args.res} 2-ch, '
                               f'random-init G, batch={args.batch}/GPU, {args.latent_steps} latent steps, '
                               f'w_latent={args.w_latent:g} w_pix={args.w_pix:g} w_disc={args.w_disc:g} w_lpips={args.w_lpips:g} '
                               f'(M_w={args.M_w}, M_x={args.M_x}, criterion_mode={args.criterion_mode}), '
                               f'contraction={args.precision}',
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}'},
    }
    if roof is not None:
        line['roofline'] = roof
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.preset == 'B' and args.w_disc == 0:
        line['cpu_baseline'] = cpu_baseline(sd, meta, args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
