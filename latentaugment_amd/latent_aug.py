"""Host-side mirror of augments/utils/util_latent_aug.py::LatentAug for the MI355X path.

Same constructor fields (read from `opt`), same `forward(w, fname) -> (img, w_aug)` contract, same attributes
(`num_ws, w_dim, z_dim, stats_dataset_w`).  The N-step optimisation itself is one call into the C ABI
(`la_latent_opt_run`); this file only moves tensors to the device, draws the crop position on the host and -- when a
process group is active -- shards independent samples over ranks and gathers the result with ONE collective.

Differences from the reference, all deliberate:
  * one process per GPU + torch.distributed (RCCL) instead of single-process nn.DataParallel (:28-33);
  * works with `gpu_ids=[k]` only on a ROCm device: there is no CPU path here (the reference's CPU path is the oracle);
  * G / banks can be injected (no pickle / zip needed for synthetic runs); on-disk formats are the next scope row;
  * the final synthesis uses explicit noise tensors drawn on the host RNG when noise_strength != 0 (defect g).
"""
import ctypes as C
import math
import os
import random

import numpy as np
import torch

from . import _lib
from .synthesis import DiscriminatorEngine, FeatureEngine, MappingEngine, SynthesisEngine


def center_crop_geometry(load_size):
    """(crop, offset) of util_dataset.get_center_crop: CenterCrop(int(sqrt(res^2/2))), torchvision offset."""
    crop = int(np.sqrt((load_size * load_size) / 2))
    return crop, int(round((load_size - crop) / 2.0))


def get_params(load_size, crop_size, preprocess='center_random_crop'):
    """Random-crop position, drawn ONCE per forward on the host (util_dataset.py:284-296)."""
    assert preprocess in ['center_random_crop', 'random_crop']
    new = load_size
    if preprocess == 'center_random_crop':
        new = int(np.sqrt((load_size * load_size) / 2))
    x = random.randint(0, max(0, new - crop_size))
    y = random.randint(0, max(0, new - crop_size))
    return {'crop_pos': (x, y)}


def write_png_gray(path, a):
    """8-bit greyscale PNG of a 2-D uint8 array (the reference writes its snapshots with cv2.imwrite, :655)."""
    import struct
    import zlib
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape
    raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 6)) +
                chunk(b'IEND', b''))


def _preproc3(p):
    """opt.lpips_preproc -> ((s0, s1, s2), (b0, b1, b2)): a scalar pair applies to all three repeated channels."""
    scale, shift = p
    scale = tuple(float(v) for v in scale) if hasattr(scale, '__len__') else (float(scale),) * 3
    shift = tuple(float(v) for v in shift) if hasattr(shift, '__len__') else (float(shift),) * 3
    assert len(scale) == 3 and len(shift) == 3
    return scale, shift


def write_curve_png(path, values, width=480, height=320, margin=24):
    """Line plot of a sequence as an 8-bit greyscale PNG: frame, polyline, nothing else (the reference draws its per-key curves
    with matplotlib, snapshot_stats :620-633; matplotlib is not a dependency here, so there are no tick labels or legend --
    the numbers are in the .jsonl next to the picture)."""
    v = np.asarray(list(values), dtype=np.float64)
    img = np.full([height, width], 255, dtype=np.uint8)
    x0, x1, y0, y1 = margin, width - margin - 1, margin, height - margin - 1
    img[y0, x0:x1 + 1] = img[y1, x0:x1 + 1] = 0
    img[y0:y1 + 1, x0] = img[y0:y1 + 1, x1] = 0
    if v.size and np.isfinite(v).any():      # (an all-NaN / all-inf curve: frame only)
        v = np.where(np.isfinite(v), v, np.nan)
        lo, hi = float(np.nanmin(v)), float(np.nanmax(v))
        span = hi - lo if hi > lo else 1.0
        xs = np.full(v.size, (x0 + x1) // 2) if v.size == 1 else np.round(x0 + 4 + (x1 - x0 - 8) * np.arange(v.size) / (v.size - 1)).astype(int)
        ys = np.round(y1 - 4 - (y1 - y0 - 8) * (np.nan_to_num(v, nan=lo) - lo) / span).astype(int)
        for i in range(v.size):
            img[max(ys[i] - 2, 0):ys[i] + 3, max(xs[i] - 2, 0):xs[i] + 3] = 0          # marker
            if i + 1 < v.size:
                n = int(max(abs(xs[i + 1] - xs[i]), abs(ys[i + 1] - ys[i]), 1))
                for t in range(n + 1):                                                 # segment, one pixel per step along the longer axis
                    img[int(round(ys[i] + (ys[i + 1] - ys[i]) * t / n)), int(round(xs[i] + (xs[i + 1] - xs[i]) * t / n))] = 64
    write_png_gray(path, img)


def shard_bounds(batch, world_size, rank):
    """Contiguous sample range of a rank: [rank*b, (rank+1)*b) with b = ceil(batch / world_size)."""
    per = (batch + world_size - 1) // world_size
    lo = min(rank * per, batch)
    hi = min(lo + per, batch)
    return lo, hi, per


def sharded(group=None):
    """True where forward() shards the batch over a process group: an initialised group of more than one rank.  (A group of ONE rank
    takes the same path -- broadcast, shard, all_gather -- when LATENTAUG_FORCE_SHARDED=1: the RCCL rehearsal of a 1-GPU box.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('LATENTAUG_FORCE_SHARDED') == '1'


def gather_shards(local, per, batch, group=None):
    """ONE all_gather of equally padded shards -> the full batch on every rank (the reference gathers to gpu_ids[0])."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local.device
    # RCCL ('nccl') gathers device buffers directly over xGMI; a gloo group (CPU rehearsal / tests) is staged through host
    stage = dev.type == 'cuda' and dist.get_backend(group) == 'gloo'
    pad = torch.zeros([per] + list(local.shape[1:]), dtype=local.dtype, device='cpu' if stage else dev)
    pad[:local.shape[0]] = local
    out = torch.empty([world * per] + list(local.shape[1:]), dtype=local.dtype, device=pad.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:batch].to(dev)


def broadcast_controls(crop_pos, group=None, device=None, host_group=None):
    """Rank 0's per-forward host draws -- (crop x, crop y, 48-bit seed of the final-synthesis noise) -- to every rank of the
    group.  Control plane only (three integers through the host); the data path keeps its single all_gather.  `host_group`: a gloo
    group over the same ranks.  Through an RCCL group the object broadcast reads its size back on the host, which waits for the
    RCCL stream, which waits for everything this rank has queued: the previous batch drains before the next one is enqueued.  Over
    the host group the ranks still meet here, but no GPU queue is involved and batches stay queued back to back."""
    import torch.distributed as dist
    ctl = [None]
    if dist.get_rank(group) == 0:
        ctl[0] = (int(crop_pos[0]), int(crop_pos[1]), random.getrandbits(48))
    src = dist.get_global_rank(group, 0) if group is not None else 0
    if host_group is not None:
        dist.broadcast_object_list(ctl, src=src, group=host_group, device=torch.device('cpu'))
        return ctl[0]
    on_host = dist.get_backend(group) == 'gloo' or device is None
    dist.broadcast_object_list(ctl, src=src, group=group, device=torch.device('cpu') if on_host else device)
    return ctl[0]


def host_control_group(group=None, device=None):
    """A gloo group over the ranks of the default (RCCL) group, for broadcast_controls; None where the data group is gloo already, is
    a sub-group (dist.new_group must be entered by EVERY process of the job, which only the default group guarantees here), or
    LATENTAUG_CTL_DEVICE=1 asks for the device path.  Collective: every rank of the default group calls it at the same point (the
    first sharded forward)."""
    import torch.distributed as dist
    if group is not None or dist.get_backend() == 'gloo' or os.environ.get('LATENTAUG_CTL_DEVICE') == '1':
        return None
    if not dist.is_gloo_available():
        return None
    try:
        g = dist.new_group(backend='gloo')
    except Exception as e:       # (e.g. no usable network interface for gloo on this host)
        print(f'[latentaugment_amd] no gloo control group ({type(e).__name__}: {e}); control broadcast stays on the device path')
        g = None
    # every rank must take the same path: agree over the data group that all of them got their host group
    ok = torch.tensor([1 if g is not None else 0], device=device if device is not None else torch.device('cuda', torch.cuda.current_device()), dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return g if int(ok.item()) == 1 else None


class InMemoryLatentCodes:
    """Minimal stand-in for util_dataset.LatentCodeDataset: fname -> inverted latent [num_ws, w_dim] (or [w_dim])."""

    def __init__(self, codes):
        self.codes = dict(codes)

    def lookup(self, fname):
        return np.asarray(self.codes[fname], dtype=np.float32)


class LatentAug:
    def __init__(self, phase, opt, save_dir, gpu_ids, generator=None, discriminator=None, banks=None, latent_codes=None,
                 feature_net=None, group=None, _shared=None):
        self.save_dir = save_dir
        self.phase = phase
        self.group = group
        self._ctl_group = None      # gloo companion of an RCCL group for the per-forward control broadcast (host_control_group)
        if not gpu_ids:
            raise _lib.LatentAugHipError(
                'LatentAug (MI355X path) needs a GPU id in gpu_ids_aug; the CPU restatement lives in oracle/ and is '
                'test infrastructure only')
        if len(gpu_ids) > 1:
            raise _lib.LatentAugHipError(
                'one process drives one GPU: launch one rank per GPU (torch.distributed / RCCL) instead of passing '
                'several gpu_ids to a single process')
        self.device = torch.device('cuda', gpu_ids[0])
        self.res = opt.img_resolution
        self.batch_size = opt.batch_size
        self.modalities = [m for m in str(opt.modalities_aug).split(',') if m]
        self.num_epochs = opt.opt_num_epochs
        self.opt_lr = opt.opt_lr
        self.truncation_psi = opt.truncation_psi
        self.w_pix, self.w_lpips, self.w_latent, self.w_disc = opt.w_pix, opt.w_lpips, opt.w_latent, opt.w_disc
        self.crop_size = opt.crop_size_aug
        self.preprocess = opt.preprocess_aug
        self.soft_aug, self.alpha = bool(opt.soft_aug), opt.alpha
        self.verbose_log = opt.verbose_log
        self.criterion_mode = getattr(opt, 'criterion_mode', 'gemm')
        self.final_noise_mode = getattr(opt, 'final_noise_mode', 'random')
        # contraction arithmetic: 'f16x2' = fp32 operands scaled by powers of two and split into 2 fp16 terms
        # (3 fp16 MFMAs per product, fp32 accumulate); 'bf16x3' = 3 bf16 terms (6 MFMAs, no range scaling needed).  Both pass
        # every fp32 parity test at unchanged tolerances.  'f32' = exact fp32 MFMA; 'bf16x2' approximate.
        self.precision = getattr(opt, 'precision', 'f16x2')
        self._script_path = None
        if self.w_lpips > 0 and feature_net is None:
            # the reference's default perceptual net (`lpips_script`): NVIDIA's TorchScript vgg16.pt, fetched from a URL by load_vgg()
            # (:35-43).  There is no network here: a LOCAL copy is accepted at opt.lpips_script_path or <model_dir>/vgg16.pt.
            cands = [getattr(opt, 'lpips_script_path', None)]
            if getattr(opt, 'lpips_script', 'lpips_script') == 'lpips_script' and getattr(opt, 'model_dir', None):
                cands.append(os.path.join(opt.model_dir, 'vgg16.pt'))
            self._script_path = next((c for c in cands if c and os.path.isfile(c)), None)
            if self._script_path is None:
                raise NotImplementedError(
                    'w_lpips > 0 needs `feature_net=` (op list for synthesis.FeatureEngine, e.g. vgg16_lpips_ops(...)) or a local copy '
                    "of NVIDIA's TorchScript vgg16.pt at opt.lpips_script_path / <model_dir>/vgg16.pt (the reference downloads it, "
                    "util_latent_aug.py:36; there is no network here), and banks['fea'] or the interim image zip to build them from")
        if generator is None:
            # load_stylegan (reference :466-484): <model_dir>/<dataset>/training-runs/<dataset_name>/<modalities>/<exp>/<pkl>
            from . import formats
            path = formats.find_network_pkl(opt.model_dir, opt.dataset_aug, opt.dataset_name_aug, self.modalities,
                                            opt.exp_stylegan, opt.network_pkl_stylegan)
            print(f'Loading stylegan from "{path}"...')
            nets_ = formats.load_network_pkl(path)
            generator = nets_['G_ema']
            if discriminator is None:
                discriminator = nets_.get('D')
        # per-process capacity: the whole batch, or -- with an active process group, where forward() hands this rank only its
        # shard -- ceil(batch / world) (`opt.max_local_batch` overrides)
        max_local = self.batch_size
        try:
            import torch.distributed as dist
            if sharded(group):
                max_local = (self.batch_size + dist.get_world_size(group) - 1) // dist.get_world_size(group)
        except (RuntimeError, ValueError):
            pass
        max_local = int(getattr(opt, 'max_local_batch', 0) or max_local)
        self.engine = SynthesisEngine.from_generator(generator, self.device, max_local, precision=self.precision)
        # (opt.operand_scale of rounds 2-3 -- 'auto' | 'bound' | 'data' -- is accepted and ignored: every fp16 operand scale is derived
        #  from the data of each pass by the producing kernels now, batch by batch; there is no calibration to freeze)
        assert self.engine.img_resolution == self.res, 'opt.img_resolution does not match the generator'
        assert self.engine.img_channels == len(self.modalities), 'one image channel per modality expected'
        self.num_ws, self.w_dim = self.engine.num_ws, self.engine.w_dim
        self.z_dim = getattr(generator, 'z_dim', self.w_dim)
        self._generator = generator
        self._mapping = None
        self.stats_dataset_w = latent_codes
        self.stats_loss = {}
        self.stats_time = {}
        self._verbose_flag = bool(opt.verbose_log)      # the reference logs the FIRST batch only (:189-191, :297-300)

        if banks is None and _shared is None and getattr(opt, 'interim_dir', None) and getattr(opt, 'dataset_aug', None):
            # real-data banks from the interim zips (reference :137-158), cached as DatasetStats pickles
            from . import formats
            root = os.path.join(opt.interim_dir, opt.dataset_aug)
            cache_dir = os.path.join(root, 'cache_dir')
            banks = {}
            wzip = os.path.join(root, str(getattr(opt, 'dataset_w_name', '')) + '.zip')
            if self.stats_dataset_w is None and os.path.isfile(wzip):
                self.stats_dataset_w = formats.LatentCodeDataset(os.path.join(root, opt.dataset_w_name + '.zip'),
                                                                 split=self.phase, w_dim=self.w_dim, num_ws=self.num_ws)
            if self.w_latent > 0:
                banks['W'] = formats.compute_stats(self.stats_dataset_w, 'latent', cache_dir, step=opt.step_w).get_all_torch()
            if self.w_pix > 0:
                ds = formats.ImgDataset(os.path.join(root, opt.dataset_name_aug + '.zip'), split=self.phase,
                                        modalities=self.modalities, resolution=self.res)
                banks['X'] = formats.compute_stats(ds, 'img', cache_dir, step=opt.step_img).get_all_torch()
        banks = banks or {}
        if _shared is not None:      # a stream lane of another LatentAug (build_lanes): the parent's device-resident banks, not copies
            banks = dict(_shared['banks'])
        lib = _lib.load()
        self._lib = lib
        crop, off = center_crop_geometry(self.res)
        self.center_crop, self.center_off = crop, off
        self.W = None
        self.Xc = None
        Mw = Mx = 0
        if self.w_latent > 0:
            W = banks['W'].to(device=self.device, dtype=torch.float32).contiguous()
            assert W.shape[1:] == (self.num_ws, self.w_dim)
            self.W, Mw = W, W.shape[0]
        if self.w_pix > 0 and _shared is not None:
            self.Xc = _shared['Xc']
            Mx = self.Xc.shape[1]
        elif self.w_pix > 0:
            X = banks['X'].to(device=self.device, dtype=torch.float32).contiguous()     # [M, C, R, R] in [-1, 1]
            assert X.shape[1:] == (len(self.modalities), self.res, self.res)
            Mx = X.shape[0]
            xc = torch.empty([Mx, X.shape[1], crop, crop], device=self.device, dtype=torch.float32)
            with torch.cuda.device(self.device):
                _lib.check(lib.la_center_crop_f32(_lib.ptr(X), _lib.ptr(xc), Mx * X.shape[1], self.res, crop, off,
                                                  _lib.stream_ptr()), 'la_center_crop')
            self.Xc = xc.permute(1, 0, 2, 3).contiguous()      # modality-major [C][M][crop*crop]
            del X, xc
        cfg = _lib.OptConfig(steps=int(self.num_epochs), lr=float(self.opt_lr), beta1=0.9, beta2=0.999, eps=1e-8,
                             w_latent=float(self.w_latent), w_pix=float(self.w_pix), w_disc=float(self.w_disc),
                             w_lpips=float(self.w_lpips),
                             criterion_mode={'gemm': 0, 'collapsed': 1}[self.criterion_mode],
                             soft_aug=int(self.soft_aug), alpha=float(self.alpha), loop_noise_mode=1,
                             final_noise_mode={'none': 0, 'const': 1, 'random': 2}[self.final_noise_mode],
                             norm_batch=int(getattr(opt, 'norm_batch', 0) or 0), crop=crop, crop_off=off)
        self._cfg = cfg
        nbytes = lib.la_latent_opt_workspace_bytes(self.res, self.engine.img_channels, self.w_dim, C.byref(cfg), Mw, Mx,
                                                   max_local)
        self._workspace = torch.empty([max(nbytes, 64)], dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        _lib.check(lib.la_latent_opt_create(self.engine.handle, self.res, self.engine.img_channels, self.w_dim,
                                            C.byref(cfg), _lib.ptr(self.W), Mw, _lib.ptr(self.Xc), Mx, max_local,
                                            _lib.ptr(self._workspace), self._workspace.numel(), C.byref(h)),
                   'la_latent_opt_create')
        self._h = h
        self._max_local = max_local
        self._opt, self._discriminator, self._feature_net = opt, discriminator, feature_net
        # stream lanes: a full local batch as two interleaved half-batch loops on two HIP streams, each with its own loop /
        # synthesis / discriminator handles.  `opt.stream_lanes`: 'auto' (default: two lanes for a full, even batch of >= 4 samples -- with
        # the discriminator a multiple of 8 -- when the perceptual criterion is off, or on TOGETHER with the discriminator: lanes_eligible),
        # 1 (never) or 2 (whenever the batch allows it).
        self.stream_lanes = getattr(opt, 'stream_lanes', 'auto')
        assert self.stream_lanes in ('auto', 1, 2), "opt.stream_lanes: 'auto', 1 or 2"
        self._lanes = None
        self._lane_stream = None
        self.lanes_active = False           # the last batch went through the lanes
        self.lanes_concurrent = True        # False: the lanes one after the other on one stream (bench.py's per-launch brackets)
        # First-batch self-check of the concurrent lanes (`opt.lanes_selfcheck`, default on): the first full batch runs through the lanes
        # one after the other AND side by side; the two must agree bit for bit (they are the same launches), otherwise this handle
        # warns and keeps the lanes serial.  Why: rounds 2-3 saw wrong results whenever two queues of this library overlapped; round 4
        # traced it to packed-FP32 instructions and removed them (DESIGN 8 'Two streams'), but the hardware mechanism is not pinned
        # down, so the product checks the property it relies on where it relies on it instead of trusting one toolchain's codegen.
        self._lanes_check = bool(getattr(opt, 'lanes_selfcheck', True))
        self.lanes_selfcheck = None         # None: not run yet | 'bit-identical' | 'differs: lanes serial from now on' | 'off'
        # launch mode of the step loop: one captured step replayed (default) or every launch eager (`opt.hip_graph = False`)
        self.hip_graph = bool(getattr(opt, 'hip_graph', True))
        _lib.check(lib.la_latent_opt_set_graph(h, int(self.hip_graph)), 'la_latent_opt_set_graph')
        # independent image criteria (discriminator, perceptual) side by side inside a step (default) or one after the other
        # (`opt.overlap_criteria = False`): bit-identical results, la_latent_opt_set_overlap
        # `opt.overlap_criteria`: True / 2 (default) = fork / join inside the captured step (two parallel branches of one graph), 1 = split
        # replay (round-5 experiment: the perceptual branch as a graph of its own on a side stream; measured equal: 145.5 against 145.1 ms
        # at preset E, 158.6 one after the other), False / 0 = one after the other
        oc = getattr(opt, 'overlap_criteria', True)
        self.overlap_criteria = int(oc) if not isinstance(oc, bool) else (2 if oc else 0)
        _lib.check(lib.la_latent_opt_set_overlap(h, self.overlap_criteria), 'la_latent_opt_set_overlap')
        # rows of the image that the loop's criteria read: the pixel criterion its centre crop (util_dataset.py:317-323), the perceptual
        # criterion a crop_size_aug window inside that crop when preprocess_aug is one of the centre modes -- the loop steps then
        # synthesise only what those rows depend on (la_latent_opt_set_row_window; `opt.loop_window = False`: whole frames in every step).
        # The discriminator reads whole frames: no window.
        self.loop_window = None
        if getattr(opt, 'loop_window', True) and self.w_disc <= 0 and (self.w_pix > 0 or self.w_lpips > 0) and \
                (self.w_lpips <= 0 or self.preprocess in ('center_crop', 'center_random_crop')):
            self.loop_window = (off, off + crop)
            _lib.check(lib.la_latent_opt_set_row_window(h, off, off + crop), 'la_latent_opt_set_row_window')
            if getattr(opt, 'loop_window_columns', True):      # (the crop is a square; the top block follows its columns too)
                _lib.check(lib.la_latent_opt_set_col_window(h, off, off + crop), 'la_latent_opt_set_col_window')
        self.disc = None
        if self.w_disc > 0:
            if discriminator is None:
                raise _lib.LatentAugHipError('w_disc > 0 needs the discriminator (pass `discriminator=` or a network pickle)')
            self.disc = DiscriminatorEngine(discriminator, self.device, max_local, precision=self.precision)
            assert self.disc.img_resolution == self.res and self.disc.img_channels == self.engine.img_channels
            _lib.check(lib.la_latent_opt_set_disc(h, self.disc.handle), 'la_latent_opt_set_disc')

        self.feat = None
        if self.w_lpips > 0:
            # perceptual criterion (reference :160-171, :387-409): features of crop_size_aug^2 crops, one bank per modality
            imgc = self.engine.img_channels
            if feature_net is not None:
                self.feat = FeatureEngine(feature_net, self.device, in_res=self.crop_size, max_batch=imgc * max_local,
                                          precision=self.precision)
            else:
                print(f'Loading VGG16 from: {self._script_path}')
                self.feat = FeatureEngine.from_torchscript(self._script_path, self.device, in_res=self.crop_size,
                                                           max_batch=imgc * max_local, precision=self.precision)
                if not hasattr(opt, 'lpips_preproc'):
                    # the script's own input layer, (x - mean_k) / std_k; the reference hands it the synthesised crop as it is (:394-395)
                    opt.lpips_preproc = (self.feat.pre_scale, self.feat.pre_shift)
            if 'fea' not in banks:
                if not (getattr(opt, 'interim_dir', None) and getattr(opt, 'dataset_aug', None)):
                    raise _lib.LatentAugHipError("w_lpips > 0 needs banks['fea'] or the interim image zip to build them from")
                banks['fea'] = self._build_feature_banks(opt)
            fea = banks['fea']
            assert len(fea) == imgc
            if _shared is not None and _shared.get('Fbank') is not None:
                self.Fbank = _shared['Fbank']
            else:
                self.Fbank = torch.stack([t.to(device=self.device, dtype=torch.float32) for t in fea]).contiguous()   # [C][Mf][F]
            assert self.Fbank.shape[2] == self.feat.num_features, 'feature bank does not match the feature net'
            Mf = self.Fbank.shape[1]
            scale, shift = _preproc3(getattr(opt, 'lpips_preproc', (1.0, 0.0)))
            nb = lib.la_latent_opt_lpips_workspace_bytes(imgc, self.feat.num_features, self.crop_size, Mf, max_local)
            self._lpips_ws = torch.empty([nb], dtype=torch.uint8, device=self.device)
            _lib.check(lib.la_latent_opt_set_lpips(h, self.feat.handle, _lib.ptr(self.Fbank), Mf, self.crop_size, float(scale[0]),
                                                   float(shift[0]), _lib.ptr(self._lpips_ws), nb), 'la_latent_opt_set_lpips')
            _lib.check(lib.la_latent_opt_set_lpips_preproc(h, (C.c_float * 3)(*scale), (C.c_float * 3)(*shift), 3),
                       'la_latent_opt_set_lpips_preproc')
        # the lanes' handles and workspaces exist from construction on (an allocation failure surfaces here, not inside the first batch)
        if _shared is None and self.lanes_eligible(self._max_local):
            self.build_lanes()

    def _build_feature_banks(self, opt):
        """fea_<mode> banks (reference :160-171 / extract_features_mode_torchscript :565-580): per real image and modality,
        a crop_size_aug^2 crop at a freshly drawn random position, repeated to 3 channels, through the feature net.
        `opt.lpips_bank_range`: 'unit' (default) feeds x/127.5-1, consistent with the synthesised images; 'raw' reproduces
        the reference's 0..255 input (SURVEY 3.4 defect f)."""
        from . import formats
        root = os.path.join(opt.interim_dir, opt.dataset_aug)
        ds = formats.ImgDataset(os.path.join(root, opt.dataset_name_aug + '.zip'), split=self.phase, modalities=self.modalities,
                                resolution=self.res)
        raw = getattr(opt, 'lpips_bank_range', 'unit') == 'raw'
        scale, shift = _preproc3(getattr(opt, 'lpips_preproc', (1.0, 0.0)))
        sc_t = torch.tensor(scale, device=self.device).reshape(1, 3, 1, 1)
        sh_t = torch.tensor(shift, device=self.device).reshape(1, 3, 1, 1)
        out = []
        for mode_id, mode in enumerate(self.modalities):
            def feature_fn(x, mode_id=mode_id):
                t = torch.from_numpy(x[:, mode_id:mode_id + 1]).to(self.device)
                if not raw:
                    t = t / 127.5 - 1
                ax, ay = self.crop_window(get_params(self.res, self.crop_size, self.preprocess)['crop_pos'])
                t = t[:, :, ay:ay + self.crop_size, ax:ax + self.crop_size].repeat(1, 3, 1, 1) * sc_t + sh_t   # plumbing
                return self.feat.forward(t.contiguous()).cpu().numpy()
            # The reference names the cache '<mode>-<crop>-features_jit-...': its contents also depend on the input range, the
            # preprocess and the feature-net weights, so those are folded into the tag -- a cache written by the reference (raw
            # 0..255 inputs, NVIDIA's weights) or under other settings is never picked up silently.
            tag = f'{mode}-{self.crop_size}'
            if not (raw and (scale, shift) == ((1.0,) * 3, (0.0,) * 3) and getattr(opt, 'lpips_cache_compat', False)):
                def fmt(v):
                    return f'{v[0]:g}' if v[0] == v[1] == v[2] else '_'.join(f'{t:g}' for t in v)
                tag += f"-{'raw' if raw else 'unit'}-s{fmt(scale)}-b{fmt(shift)}-w{self.feat.weights_digest}"
            st = formats.compute_stats(ds, 'features_jit', os.path.join(root, 'cache_dir'), cache_tag=tag,
                                       step=opt.step_img, feature_fn=feature_fn)
            out.append(st.get_all_torch())
        return out

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.la_latent_opt_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ helpers (reference :493-498)
    def broadcasting(self, latent):
        return latent.repeat([1, self.num_ws, 1])

    @staticmethod
    def reverse_broadcasting(latent):
        return latent[:, :1, :]

    # ------------------------------------------------------------------ the hot path
    def crop_window(self, crop_pos):
        """Absolute (x, y) of the crop_size_aug window for a position drawn by get_params (relative to the centre crop
        when preprocess is 'center_random_crop'; util_dataset.py:298-315)."""
        x1, y1 = crop_pos
        off = self.center_off if self.preprocess in ('center_crop', 'center_random_crop') else 0
        return off + int(x1), off + int(y1)

    def run_local(self, w, final_noises=None, want_losses=False, crop_pos=None, trace=None, out=None):
        """w [b,1,w_dim] on this device -> (img [b,C,R,R], w_aug [b,num_ws,w_dim], losses or None).
        out (optional): (img, w_aug) tensors to fill instead of allocating them on the current stream.
        trace (optional): dict that receives the per-step snapshots 'w' [steps,b,w_dim] (latent after the step) and 'img'
        [steps,b,C,R,R]; trace['want'] (default ('w', 'img')) selects them, and may name 'grad' [steps,b,w_dim] = dL/dw."""
        if self.feat is not None:
            if crop_pos is None:
                crop_pos = getattr(self, 'crop_params', None)
                crop_pos = crop_pos['crop_pos'] if crop_pos else get_params(self.res, self.crop_size, self.preprocess)['crop_pos']
            ax, ay = self.crop_window(crop_pos)
            _lib.check(self._lib.la_latent_opt_set_crop_pos(self._h, ax, ay), 'la_latent_opt_set_crop_pos')
        w = w.to(device=self.device, dtype=torch.float32).contiguous()
        b = w.shape[0]
        assert w.ndim == 3 and w.shape[1:] == (1, self.w_dim)
        if out is not None:
            img, w_aug = out
            assert img.shape == (b, self.engine.img_channels, self.res, self.res) and w_aug.shape == (b, self.num_ws, self.w_dim)
            assert img.is_contiguous() and w_aug.is_contiguous() and img.dtype == w_aug.dtype == torch.float32
        else:
            img = torch.empty([b, self.engine.img_channels, self.res, self.res], device=self.device, dtype=torch.float32)
            w_aug = torch.empty([b, self.num_ws, self.w_dim], device=self.device, dtype=torch.float32)
        losses = torch.zeros([max(self.num_epochs, 1), 4], device=self.device) if want_losses else None
        fn = None
        if self._cfg.final_noise_mode == 2:
            if final_noises is None:
                final_noises = self.engine.make_noises(b)
            fn = self.engine.noise_pointer_array(final_noises)
        tw = ti = tg = None
        want_img = trace is not None and 'img' in trace.get('want', ('w', 'img'))
        if trace is not None and self.num_epochs > 0:
            tw = torch.empty([self.num_epochs, b, self.w_dim], device=self.device, dtype=torch.float32)
            if want_img:
                ti = torch.empty([self.num_epochs, b, self.engine.img_channels, self.res, self.res], device=self.device, dtype=torch.float32)
            if 'grad' in trace.get('want', ()):
                tg = torch.empty([self.num_epochs, b, self.w_dim], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            if tw is not None:
                _lib.check(self._lib.la_latent_opt_set_trace(self._h, _lib.ptr(tw), _lib.ptr(ti)), 'la_latent_opt_set_trace')
                _lib.check(self._lib.la_latent_opt_set_grad_trace(self._h, _lib.ptr(tg)), 'la_latent_opt_set_grad_trace')
            try:
                _lib.check(self._lib.la_latent_opt_run(self._h, _lib.ptr(w), b, fn, _lib.ptr(img), _lib.ptr(w_aug),
                                                       _lib.ptr(losses), _lib.stream_ptr()), 'la_latent_opt_run')
            finally:
                if tw is not None:
                    _lib.check(self._lib.la_latent_opt_set_trace(self._h, None, None), 'la_latent_opt_set_trace')
                    _lib.check(self._lib.la_latent_opt_set_grad_trace(self._h, None), 'la_latent_opt_set_grad_trace')
        if tw is not None:
            trace['w'] = tw
            if ti is not None:
                trace['img'] = ti
            if tg is not None:
                trace['grad'] = tg
        self._keep = final_noises
        return img, w_aug, losses

    # ---- stream lanes: the local batch as two interleaved half-batch loops on two HIP streams
    def lanes_eligible(self, b):
        """Two lanes need the FULL local batch (the criteria's 1/(m*n) is fixed per handle), an even one of at least 4 (measured: +3 % at
        4 x 256^2, +5 % at 4 x 512^2, +3.5 % at 16 x 256^2; four lanes lose), and -- with the discriminator --
        a multiple of 8: MinibatchStd groups sample n with n + b/4, n + 2b/4, n + 3b/4 (MinibatchStdLayer: `x.reshape(G, -1, F, c, H, W)`, group size 4, as the pickled discriminators of the reference carry it), and the even /
        odd halves of the batch keep exactly those groups only then."""
        if self.stream_lanes == 1 or b != self._max_local or b < 4 or b % 2:
            return False
        if self.w_disc > 0 and b % 8:
            return False
        if self.stream_lanes == 2 or self.w_lpips <= 0:
            return True
        # perceptual criterion: lanes pay only beside the discriminator, with the criteria branches of a lane replayed on two streams
        # (overlap mode 1, build_lanes): preset E 144.1 -> 141.7 ms, three alternating pairs on one box (round 5); with the criteria one
        # after the other (mode 0) 148.9, with the fork inside each lane's captured step (mode 2) 185
        return self.w_disc > 0 and self.overlap_criteria >= 1

    def build_lanes(self):
        import copy
        half = self._max_local // 2
        opt = copy.copy(self._opt)
        opt.batch_size, opt.max_local_batch, opt.norm_batch = half, half, self._max_local
        opt.verbose_log, opt.stream_lanes = False, 1
        # two lanes that each fork inside their captured step lose badly (preset E: 185 against 142 ms): a lane's criteria branches are
        # replayed as graphs of their own on two streams instead (bit-identical; la_latent_opt_set_overlap mode 1)
        opt.overlap_criteria = 1 if self.overlap_criteria == 2 else self.overlap_criteria
        shared = {'banks': {'W': self.W, 'fea': list(self.Fbank) if self.feat is not None else None}, 'Xc': self.Xc,
                  'Fbank': self.Fbank if self.feat is not None else None}
        # (banks={}: a lane never builds banks of its own -- neither from `banks` nor from the interim zips -- it gets the parent's
        #  device-resident tensors through _shared)
        self._lanes = [LatentAug(self.phase, opt, self.save_dir, [self.device.index], generator=self._generator,
                                 discriminator=self._discriminator, banks={}, latent_codes=self.stats_dataset_w,
                                 feature_net=self._feature_net, _shared=shared) for _ in range(2)]
        self._lane_stream = torch.cuda.Stream(device=self.device)

    def run_lanes(self, w, final_noises=None, crop_pos=None, concurrent=True):
        """run_local for a full local batch through the two lanes: samples 0, 2, 4, .. on the current stream, samples 1, 3, 5, .. on the
        lane stream (forked from / joined into the current stream); every buffer either lane touches is allocated on the current
        stream before the fork.  `concurrent=False` runs the lanes one after the other on the current stream (same launches)."""
        if self._lanes is None:
            self.build_lanes()
        w = w.to(device=self.device, dtype=torch.float32).contiguous()
        b = w.shape[0]
        assert b == self._max_local and b % 2 == 0, 'stream lanes take the full (even) local batch'
        if crop_pos is None and self.feat is not None:
            crop_pos = getattr(self, 'crop_params', None)
            crop_pos = crop_pos['crop_pos'] if crop_pos else get_params(self.res, self.crop_size, self.preprocess)['crop_pos']
        if self._cfg.final_noise_mode == 2 and final_noises is None:
            final_noises = self.engine.make_noises(b)      # ONE draw for the batch, as the single loop makes it; a lane keeps its rows
        parts, outs, fns = [], [], []
        for k in (0, 1):
            parts.append(w[k::2].contiguous())
            outs.append((torch.empty([b // 2, self.engine.img_channels, self.res, self.res], device=self.device, dtype=torch.float32),
                         torch.empty([b // 2, self.num_ws, self.w_dim], device=self.device, dtype=torch.float32)))
            fns.append([t[k::2].contiguous() if t is not None else None for t in final_noises] if final_noises is not None else None)
        main = torch.cuda.current_stream(self.device)
        side = self._lane_stream if concurrent else main
        if concurrent:
            side.wait_stream(main)
        self._lanes[0].run_local(parts[0], fns[0], crop_pos=crop_pos, out=outs[0])
        with torch.cuda.stream(side):
            self._lanes[1].run_local(parts[1], fns[1], crop_pos=crop_pos, out=outs[1])
        if concurrent:
            main.wait_stream(side)
        img = torch.empty([b, self.engine.img_channels, self.res, self.res], device=self.device, dtype=torch.float32)
        w_aug = torch.empty([b, self.num_ws, self.w_dim], device=self.device, dtype=torch.float32)
        for k in (0, 1):
            img[k::2] = outs[k][0]
            w_aug[k::2] = outs[k][1]
        self._keep_lanes = (parts, outs, fns)
        return img, w_aug, None

    def run_batch(self, w, final_noises=None):
        """The local batch through two stream lanes when it qualifies (lanes_eligible), through the single loop otherwise."""
        self.lanes_active = self.lanes_eligible(w.shape[0])
        if not self.lanes_active:
            return self.run_local(w, final_noises)
        if self.lanes_selfcheck is None and self.lanes_concurrent:
            return self._lanes_first_batch(w, final_noises)
        return self.run_lanes(w, final_noises, concurrent=self.lanes_concurrent)

    def _lanes_first_batch(self, w, final_noises):
        """The first full batch of a handle: lanes one after the other, then side by side, on the same inputs; bit-identical or the
        handle stays serial (see __init__).  Costs one extra batch, once."""
        if not self._lanes_check:
            self.lanes_selfcheck = 'off'
            return self.run_lanes(w, final_noises, concurrent=True)
        if self._lanes is None:
            self.build_lanes()
        if self._cfg.final_noise_mode == 2 and final_noises is None:
            final_noises = self.engine.make_noises(w.shape[0])      # one draw for both runs
        crop_pos = None
        if self.feat is not None:
            cp = getattr(self, 'crop_params', None)
            crop_pos = cp['crop_pos'] if cp else get_params(self.res, self.crop_size, self.preprocess)['crop_pos']
        ref_img, ref_w, _ = self.run_lanes(w, final_noises, crop_pos=crop_pos, concurrent=False)
        img, w_aug, _ = self.run_lanes(w, final_noises, crop_pos=crop_pos, concurrent=True)
        if torch.equal(img, ref_img) and torch.equal(w_aug, ref_w):
            self.lanes_selfcheck = 'bit-identical'
            return img, w_aug, None
        import warnings
        dw = float((w_aug - ref_w).abs().max())
        self.lanes_selfcheck = 'differs: lanes serial from now on'
        self.lanes_concurrent = False
        warnings.warn(f'latentaugment_amd: the two stream lanes side by side did not reproduce the lanes one after the other on the first '
                      f'batch (max |dw| {dw:.3e}); this handle runs them one after the other from now on (opt.stream_lanes = 1 avoids the '
                      f'lanes altogether)')
        return ref_img, ref_w, None

    # ---- verbose_log artefacts of the first batch (reference :278-300, :620-655)
    def _log_first_batch(self, losses, elapsed, trace, fname, times=None):
        import json
        import pickle
        L = losses.cpu().numpy()
        active = [('loss_latent', 0, self._cfg.w_latent), ('loss_disc', 2, self._cfg.w_disc), ('loss_pix', 1, self._cfg.w_pix),
                  ('loss_lpips', 3, self._cfg.w_lpips)]
        for e in range(self.num_epochs):
            st = {name: float(L[e, col]) for name, col, wgt in active if wgt > 0}      # only the active criteria are logged (:233-268)
            st['loss'] = float(-L[e, 0] - L[e, 1] - L[e, 3] + L[e, 2])
            self.stats_loss[f'epoch_{e}'] = st
            # the loop runs on the device without a host round trip per epoch: the times are device times between HIP events around the
            # criteria of the epoch (seconds, the reference's keys; a criterion's bracket holds its loss scalar and its gradient launches --
            # the reference's the forward only, its backward sits in the untimed loss.backward()); without them the batch time spread evenly
            if times is not None:
                self.stats_time[f'epoch_{e}'] = {'time_latent': float(times[e, 0]), 'time_disc': float(times[e, 1]), 'time_pix': float(times[e, 2]),
                                                 'time_lpips': float(times[e, 3]), 'time_epoch': float(times[e, 4])}
            else:
                self.stats_time[f'epoch_{e}'] = {'time_epoch': elapsed / max(self.num_epochs, 1)}
            desc = ''.join(f'{k} {v:<4.2f} ' for k, v in st.items()) + '||| ' + ''.join(f'{k} {v:<4.3f} ' for k, v in self.stats_time[f'epoch_{e}'].items())
            print(f'epoch {e + 1:>4d}/{self.num_epochs}, {desc}')
        if self.save_dir and self.num_epochs > 0:
            os.makedirs(self.save_dir, exist_ok=True)
            for stats, title in ((self.stats_loss, 'losses'), (self.stats_time, 'times [s]')):
                ticks = list(stats.values())
                for key in ticks[0].keys():      # one curve per logged quantity, named as snapshot_stats names them (:620-633)
                    write_curve_png(os.path.join(self.save_dir, f'{title}_{key}.png'), [x[key] for x in ticks])
                with open(os.path.join(self.save_dir, f'{title}.jsonl'), 'w') as f:
                    f.write(json.dumps(stats, indent=2) + '\n')
            if trace and 'w' in trace and fname:      # snapshots only with a batch of one (:292-295)
                base = os.path.splitext(os.path.basename(str(fname[0])))[0]
                tw, ti = trace['w'].cpu().numpy(), trace['img'].cpu().numpy()
                w_in = trace.get('w_in')
                for e in range(self.num_epochs):
                    # w_<name>_<e>.pkl: the reference passes the loop's INPUT latent `w` to snap_w (:292, :636), so its file holds
                    # the same array at every epoch; mirrored.  The latent after this epoch's update goes to w_opt_<name>_<e>.pkl.
                    with open(os.path.join(self.save_dir, f'w_{base}_{e}.pkl'), 'wb') as f:
                        pickle.dump((w_in if w_in is not None else tw[e]).squeeze(), f, pickle.HIGHEST_PROTOCOL)
                    with open(os.path.join(self.save_dir, f'w_opt_{base}_{e}.pkl'), 'wb') as f:
                        pickle.dump(tw[e].squeeze(), f, pickle.HIGHEST_PROTOCOL)
                    if ti.shape[2] >= 2:
                        row = np.concatenate([ti[e, 0, 0], ti[e, 0, 1]], axis=1)      # modality A | modality B
                    else:
                        row = ti[e, 0, 0]
                    row = ((np.clip(row, -1.0, 1.0) + 1) / 2 * 255.0).astype(np.uint8)
                    write_png_gray(os.path.join(self.save_dir, f'{base}_{e}.png'), row)

    def forward(self, w, fname=None, final_noises=None):
        """LatentAug.forward(w, fname) (reference :207-310): returns (imgAB_aug, w_aug) for the FULL batch.

        With an initialised process group every rank passes the same full-batch `w`; rank k optimises samples
        [k*b, (k+1)*b) and a single all_gather returns the whole batch everywhere."""
        if w.ndim == 2:
            w = self.z_to_w(w)                      # reference :209-210
        # crop position: drawn once per forward on the host as the reference does (:216); only the (future) LPIPS
        # criterion consumes it, but the draw is kept so the python RNG stream matches the reference's.
        self.crop_params = get_params(self.res, self.crop_size, self.preprocess)
        import torch.distributed as dist
        if sharded(self.group):
            B = w.shape[0]
            rank = dist.get_rank(self.group)
            lo, hi, per = shard_bounds(B, dist.get_world_size(self.group), rank)
            # The reference draws ONE crop position per forward for the whole batch (:216) and one noise stream for the final
            # synthesis: rank 0's draws are the batch's (control plane: one tiny host-side broadcast of 3 integers; the data
            # path still has exactly one collective).  The final noise is a function of (seed, layer, global batch) only and a rank
            # takes its rows of it, so the gathered batch does not depend on how it was sharded.
            cx, cy, noise_seed = broadcast_controls(self.crop_params['crop_pos'], self.group, self.device, self._host_group())
            self._last_controls = (cx, cy, noise_seed)
            self.crop_params = {'crop_pos': (cx, cy)}
            timers = getattr(self, 'shard_timers', None)      # bench.py: [(start, loop done, gather done)] HIP events per forward
            if timers is not None:
                ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                ev[0].record()
            if hi > lo:
                if final_noises is not None:
                    fn = [t[lo:hi].contiguous() if t is not None else None for t in final_noises]
                elif self._cfg.final_noise_mode == 2:
                    fn = self.engine.make_noises(B, batch_seed=noise_seed, rows=(lo, hi))
                else:
                    fn = None
                if self._verbose_flag and rank == 0:
                    # first-batch log of the reference (:278-300): rank 0 reports its own shard, as replica 0 of a DataParallel run would
                    img, w_aug = self._run_verbose(w[lo:hi], fn, fname[lo:hi] if fname is not None else None)
                else:
                    img, w_aug, _ = self.run_batch(w[lo:hi], fn)
                self._verbose_flag = False
            else:
                img = torch.empty([0, self.engine.img_channels, self.res, self.res], device=self.device)
                w_aug = torch.empty([0, self.num_ws, self.w_dim], device=self.device)
            # one collective per batch: image and latent packed into a single buffer
            flat = torch.cat([img.reshape(img.shape[0], -1), w_aug.reshape(w_aug.shape[0], -1)], dim=1)
            if timers is not None:
                ev[1].record()
            full = gather_shards(flat, per, B, self.group)
            if timers is not None:
                ev[2].record()
                timers.append(ev)
            n_img = self.engine.img_channels * self.res * self.res
            img = full[:, :n_img].reshape(B, self.engine.img_channels, self.res, self.res)
            w_aug = full[:, n_img:].reshape(B, self.num_ws, self.w_dim)
            return img, w_aug
        if self._verbose_flag:
            img, w_aug = self._run_verbose(w, final_noises, fname)
            self._verbose_flag = False
            return img, w_aug
        img, w_aug, _ = self.run_batch(w, final_noises)
        return img, w_aug

    def _run_verbose(self, w, final_noises, fname):
        """One batch with the reference's first-batch artefacts (:278-300): loss scalars of every epoch, and with a batch of one the
        per-epoch snapshots."""
        import time
        trace = {} if w.shape[0] == 1 else None
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.la_latent_opt_set_time_trace(self._h, 1), 'la_latent_opt_set_time_trace')
        t0 = time.time()
        try:
            img, w_aug, losses = self.run_local(w, final_noises, want_losses=True, trace=trace)
            torch.cuda.synchronize(self.device)
            elapsed = time.time() - t0
            times = None
            if self.num_epochs > 0:      # per-criterion device times of every epoch (the reference's time_* keys, :221-272)
                buf = (C.c_float * (5 * self.num_epochs))()
                _lib.check(self._lib.la_latent_opt_get_times(self._h, buf), 'la_latent_opt_get_times')
                times = np.asarray(buf, dtype=np.float64).reshape(self.num_epochs, 5) * 1e-3
        finally:
            _lib.check(self._lib.la_latent_opt_set_time_trace(self._h, 0), 'la_latent_opt_set_time_trace')
        if trace is not None:
            trace['w_in'] = w.detach().cpu().numpy()
        self._log_first_batch(losses, elapsed, trace, fname, times)
        return img, w_aug

    def _host_group(self):
        if self._ctl_group is None:      # once; False = none
            self._ctl_group = host_control_group(self.group, self.device) or False
        return self._ctl_group or None

    @property
    def graph_state(self):
        """1: the optimisation step is replayed from a captured hipGraph; 0: eager launches (`opt.hip_graph = False`, or no batch has
        run yet); -1: eager because the runtime refused the capture.  (With stream lanes: of the lanes, which run the batches.)"""
        if self.lanes_active and self._lanes:
            return min(l.graph_state for l in self._lanes)
        return int(self._lib.la_latent_opt_graph_state(self._h))

    __call__ = forward

    @property
    def mapping(self):
        if self._mapping is None:
            self._mapping = MappingEngine(self._generator, self.device)
            self.z_dim = self._mapping.z_dim
        return self._mapping

    def z_to_w(self, z):
        """reference :459-464: w = reverse_broadcasting(G.mapping(z, None, truncation_psi))."""
        ws = self.mapping.forward(z.to(self.device), self.num_ws, self.truncation_psi)
        return self.reverse_broadcasting(ws)

    def forward_ganrand(self, z, noises=None):
        """reference :202-205: w_aug = G.mapping(z, c=None, truncation_psi); img = G.synthesis(w_aug)  (rand_aug mode)."""
        import torch.distributed as dist
        if sharded(self.group):
            # rand_aug with a process group: per-rank capacity is the shard (constructor), so the batch is sharded exactly as in
            # forward() -- rank 0's z is the batch's (the reference draws it once on the host, latent_aug.py:306-308), rank k maps and
            # synthesises samples [k*b, (k+1)*b), ONE all_gather returns image + latent to every rank
            B = z.shape[0]
            rank = dist.get_rank(self.group)
            lo, hi, per = shard_bounds(B, dist.get_world_size(self.group), rank)
            _, _, noise_seed = broadcast_controls((0, 0), self.group, self.device, self._host_group())
            gloo = dist.get_backend(self.group) == 'gloo'
            zb = z.detach().to('cpu' if gloo else self.device, torch.float32).contiguous()
            dist.broadcast(zb, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            n_img = self.engine.img_channels * self.res * self.res
            if hi > lo:
                ws = self.mapping.forward(zb[lo:hi].to(self.device), self.num_ws, self.truncation_psi)
                if noises is not None:
                    fn = [t[lo:hi].contiguous() if t is not None else None for t in noises]
                elif self.final_noise_mode == 'random':
                    fn = self.engine.make_noises(B, batch_seed=noise_seed, rows=(lo, hi))
                else:
                    fn = None
                img = self.engine.forward(ws, noise_mode=self.final_noise_mode, noises=fn)
                flat = torch.cat([img.reshape(hi - lo, -1), ws.reshape(hi - lo, -1)], dim=1)
            else:
                flat = torch.empty([0, n_img + self.num_ws * self.w_dim], device=self.device)
            full = gather_shards(flat, per, B, self.group)
            return (full[:, :n_img].reshape(B, self.engine.img_channels, self.res, self.res),
                    full[:, n_img:].reshape(B, self.num_ws, self.w_dim))
        ws = self.mapping.forward(z.to(self.device), self.num_ws, self.truncation_psi)
        img = self.engine.forward(ws, noise_mode=self.final_noise_mode, noises=noises)
        return img, ws


def define_latentaugment(module_name, phase, opt, save_dir, gpu_ids=[], **inject):
    """util_latent_aug.define_latentaugment (reference :45-64) -- returns the module itself (no DataParallel)."""
    if module_name == 'latent_aug':
        return LatentAug(phase, opt, save_dir, gpu_ids, **inject)
    raise NotImplementedError('Module name [%s] is not recognized' % module_name)
