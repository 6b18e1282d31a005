"""Experiment: the bench batch as two half-batches (B = 4 each) on two HIP streams of ONE process, each half with its own loop /
synthesis handles (no shared scratch), against the single B = 8 loop -- the half-batches are independent samples, so while one is
in its latency-bound low-resolution layers the other can fill the chip.  Also checks that overlapping changes no bit of either
half (round 2 saw wrong results when two streams of ONE synthesis handle overlapped)."""
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
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 2
OFFSET_US = float(os.environ.get('LA_EXP_LANE_OFFSET_US', '0'))
sys.argv = ['bench.py'] + sys.argv[2:]                                     # bench arguments after the stream count (--preset E, --w-disc 0.01)
args = bench.apply_preset(bench.parse())
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
NB = args.batch
w0 = synthetic.make_latents(NB, seed=1).to(dev)


extra = {}
banks = {'W': W, 'X': X}
if args.w_lpips > 0:
    extra['feature_net'] = synthetic.make_vgg16_lpips_ops(seed=7)
    F = synthetic.lpips_num_features(64)
    gen = torch.Generator(device=dev).manual_seed(5)
    banks['fea'] = [torch.randn([args.M_x, F], device=dev, generator=gen) * (1.0 / F) ** 0.5 for _ in range(2)]
if args.w_disc > 0:
    extra['discriminator'] = synthetic.make_discriminator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base)


def make(batch):
    opt = bench.make_opt(args, 0, batch)
    opt.final_noise_mode = 'const'
    return LatentAug('train', opt, '/tmp', [0], generator=sd, banks=banks, **extra)


full = make(NB)
parts = [make(NB // NG) for _ in range(NG)]
streams = [torch.cuda.Stream(device=dev, priority=(-1 if (k == 0 and os.environ.get('LA_EXP_LANE_PRIORITY')) else 0)) for k in range(NG)]      # LA_EXP_LANE_PRIORITY=1: lane 0 on a high-priority stream
CU_MASK = os.environ.get('LA_EXP_CU_MASK', '')      # 'half': lane k gets CUs [128 k, 128 k + 128); 'alt': even / odd 32-bit words of the mask
if CU_MASK:
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    streams = []
    for k in range(NG):
        words = [0] * 8
        for wd in range(8):
            if (CU_MASK == 'half' and (wd // (8 // NG)) == k) or (CU_MASK == 'alt' and wd % NG == k):
                words[wd] = 0xFFFFFFFF
        arr = (ctypes.c_uint32 * 8)(*words)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
        assert rc == 0, rc
        streams.append(torch.cuda.ExternalStream(st.value, device=dev))
    print('CU-masked streams:', CU_MASK, flush=True)
random.seed(6)


def run_full():
    return full.run_local(w0, crop_pos=(0, 0))[:2]


def run_parts(concurrent):
    outs = []
    per = NB // NG
    for k, (la, st) in enumerate(zip(parts, streams)):
        with torch.cuda.stream(st if concurrent else torch.cuda.current_stream()):
            if concurrent:
                st.wait_stream(torch.cuda.default_stream(dev))
                if k > 0 and OFFSET_US > 0:      # de-phasing experiment: lane k starts k * OFFSET_US later (LA_EXP_LANE_OFFSET_US)
                    torch.cuda._sleep(int(k * OFFSET_US * 2100))
            outs.append(la.run_local(w0[k::NG].contiguous(), crop_pos=(0, 0))[:2])      # interleaved: keeps D's MinibatchStd groups (n, n+2, ..)
        if not concurrent:
            torch.cuda.synchronize()
    if concurrent:
        for st in streams:
            torch.cuda.default_stream(dev).wait_stream(st)
    return outs


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


t_full = timed(run_full)
t_seq = timed(lambda: run_parts(False))
t_con = timed(lambda: run_parts(True))
print(f'one loop of {NB}: {1e3 * t_full:.1f} ms ({NB / t_full:.1f} images/s); {NG} loops of {NB // NG} one after the other: {1e3 * t_seq:.1f} ms '
      f'({NB / t_seq:.1f}); on {NG} streams: {1e3 * t_con:.1f} ms ({NB / t_con:.1f} images/s)', flush=True)
a = run_parts(False); torch.cuda.synchronize()
a2 = run_parts(False); torch.cuda.synchronize()
b = run_parts(True); torch.cuda.synchronize()
b2 = run_parts(True); torch.cuda.synchronize()
for k in range(NG):
    print(f'part {k}: alone twice {float((a[k][1] - a2[k][1]).abs().max()):.3e}; overlapped twice {float((b[k][1] - b2[k][1]).abs().max()):.3e}', flush=True)
for k in range(NG):
    di = float((a[k][0] - b[k][0]).abs().max()); dw = float((a[k][1] - b[k][1]).abs().max())
    print(f'part {k}: overlapped vs alone: max |d img| {di:.3e}, max |d w| {dw:.3e}', flush=True)

# ---- both half-batch loops (eager launches inside) captured as ONE graph: the halves are parallel branches of it, and each half's
# discriminator / perceptual fork is a branch of its branch.  (Two separately replayed graphs that fork inside do not overlap at all.)
if os.environ.get('LA_EXP_ONE_GRAPH', '1') != '0':
    lib = parts[0]._lib
    for la in parts:
        lib.la_latent_opt_set_graph(la._h, 0)
    wp = [w0[k::NG].contiguous() for k in range(NG)]
    for k, la in enumerate(parts):          # first-use effects outside the capture
        la.run_local(wp[k], crop_pos=(0, 0))
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    res = []
    with torch.cuda.graph(g, stream=cap, capture_error_mode='relaxed'):
        for k, (la, st) in enumerate(zip(parts, streams)):
            if k == 0:
                res.append(la.run_local(wp[k], crop_pos=(0, 0))[:2])
            else:
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    res.append(la.run_local(wp[k], crop_pos=(0, 0))[:2])
        for st in streams[1:]:
            cap.wait_stream(st)
    t_g = timed(g.replay)
    print(f'both halves as branches of ONE captured graph: {1e3 * t_g:.1f} ms ({NB / t_g:.1f} images/s)', flush=True)
    g.replay(); torch.cuda.synchronize()
    for k in range(NG):
        print(f'part {k}: one graph vs alone: max |d w| {float((a[k][1] - res[k][1]).abs().max()):.3e}', flush=True)
