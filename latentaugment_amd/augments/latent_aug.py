"""LatentAugment plugin: drop-in for the reference's augments/latent_aug.py::LatentAugment (:44-324).

Same command-line options (names, types, defaults: reference :57-96), same methods and attributes used by the drivers
(backbone_latentaug.py:86-124): set_input / forward / get_output / get_latent_input / get_latent_output / sanity_check /
stats_time.  The optimisation runs on the MI355X through latentaugment_amd.latent_aug.LatentAug.

Deliberate behavioural fixes: construction does not dereference `.module` (defect a); the batch is
sized by len(fname), so a last partial batch works (defect h).
"""
import os
import random
import time

import numpy as np
import torch

from .. import latent_aug as util_latent_aug
from ..latent_aug import write_png_gray
from .base_aug import BaseAugment


def reverse_broadcasting(latent):
    return latent[:, :1, :]


def set_gpu_ids(gpu_ids):
    out = []
    for s in str(gpu_ids).split(','):
        if s.strip() != '' and int(s) >= 0:
            out.append(int(s))
    return out


# The option surface is the contract with the reference's drivers (names, types, defaults: reference latent_aug.py:57-96); the help
# texts are this package's own.  (flag, kwargs)
_STR_OPTIONS = [
    ('--gpu_ids_aug', dict(type=str, default='0', help='device index of the augmenter; one id per process (one rank per GPU). Negative / empty: no device, which this backend refuses')),
    ('--dataset_aug', dict(metavar='DIR', default='Pelvis_2.1_repo_no_mask', help='dataset folder under model_dir / interim_dir')),
    ('--dataset_name_aug', dict(metavar='DIR', default='Pelvis_2.1_repo_no_mask-num-375_train-0.70_val-0.20_test-0.10', help='name of the image zip and of the training-run folder')),
    ('--modalities_aug', dict(metavar='DIR', default='MR_nonrigid_CT,MR_MR_T2', help='comma-separated modalities = image channels of the generator')),
    ('--exp_stylegan', dict(metavar='DIR', default='00003', help='training-run prefix of the StyleGAN2 experiment')),
    ('--network_pkl_stylegan', dict(metavar='DIR', default='network-snapshot-005320.pkl', help='network pickle inside that run (G_ema, D)')),
    ('--dataset_w_name', dict(metavar='DIR', default='Pelvis_2.1_repo_no_mask-num-375_train-0.70_val-0.20_test-0.10-expinv_00001', help='zip of the inverted latents, one pickle per slice')),
    ('--exp_inv', dict(metavar='DIR', default='00001', help='inversion experiment id (kept for command-line compatibility)')),
    ('--network_pkl_inv', dict(metavar='DIR', default='', help='inversion network pickle (kept for command-line compatibility)')),
]
_NUM_OPTIONS = [
    ('--img_resolution', int, 256, 'output resolution of the generator'),
    ('--truncation_psi', float, 1.0, 'truncation of mapped latents (rand_aug / z inputs)'),
    ('--step_img', int, 20, 'keep every n-th slice of a patient when the image bank is built'),
    ('--step_w', int, 5, 'keep every n-th slice of a patient when the latent bank is built'),
    ('--opt_num_epochs', int, 10, 'Adam steps on the latent per batch'),
    ('--opt_lr', float, 0.01, 'Adam learning rate'),
    ('--crop_size_aug', int, 64, 'side of the window the perceptual criterion looks at'),
    ('--w_pix', float, 1.0, 'weight of the pixel-space distance to the image bank'),
    ('--w_lpips', float, 1.0, 'weight of the perceptual distance to the feature bank'),
    ('--w_latent', float, 1.0, 'weight of the latent-space distance to the latent bank'),
    ('--w_disc', float, 1.0, 'weight of the discriminator (realism) term'),
    ('--p_thres', float, 1.0, 'a batch is augmented when a uniform draw exceeds this value'),
    ('--alpha', float, 1.0, 'soft_aug: weight of the moved latent in the blend with the inverted one'),
]


class LatentAugment(BaseAugment):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--model_dir', metavar='DIR', required=True, help='root of the pretrained networks (training-runs/...; a local vgg16.pt is looked up here)')
        parser.add_argument('--interim_dir', metavar='DIR', required=True, help='root of the interim zips and of the bank cache')
        for flag, kw in _STR_OPTIONS:
            parser.add_argument(flag, **kw)
        for flag, typ, default, text in _NUM_OPTIONS:
            parser.add_argument(flag, type=typ, default=default, help=text)
        parser.add_argument('--rand_aug', action='store_true', help='no optimisation: images of freshly mapped random z')
        parser.add_argument('--lower_bound_clip', action='store_true', help='clamp output pixels at -1 from below')
        parser.add_argument('--lpips_script', type=str, default='lpips_script', help="perceptual net: 'lpips_script' = the TorchScript vgg16.pt")
        parser.add_argument('--init_w', type=str, default='random', help="start latent of the optimisation: 'inv' (inverted latent of the slice)")
        parser.add_argument('--preprocess_aug', type=str, default='center_random_crop', help='window policy of the perceptual criterion: center_crop | random_crop | center_random_crop | original')
        parser.add_argument('--soft_aug', type=bool, default=False, help='return a blend of the inverted and the moved latent (see --alpha)')
        parser.add_argument('--verbose_log', type=bool, default=False, help='log losses / times of the first batch and write its snapshots')
        return parser

    def __init__(self, opt):
        BaseAugment.__init__(self, opt)
        self.gpu_ids_aug = set_gpu_ids(opt.gpu_ids_aug)
        self.device = torch.device('cuda:{}'.format(self.gpu_ids_aug[0])) if self.gpu_ids_aug else torch.device('cpu')
        self.phase = opt.phase
        self.batch_size = opt.batch_size
        self.rand_aug = opt.rand_aug
        self.lower_bound_clip = opt.lower_bound_clip
        self.p_thres = opt.p_thres
        self.init_w = opt.init_w
        self.verbose_log = opt.verbose_log
        self.stats_time = []

        if self.phase == 'train':
            if self.rand_aug:
                opt.w_pix = opt.w_lpips = opt.w_latent = opt.w_disc = 0.0
                opt.init_w = 'random'
                self.init_w = opt.init_w
                opt.opt_num_epochs = 0
                opt.soft_aug = False
            inject = getattr(opt, 'inject', None) or {}
            self.latent_aug = util_latent_aug.define_latentaugment(
                module_name='latent_aug', phase=opt.phase, opt=opt, save_dir=self.save_dir, gpu_ids=self.gpu_ids_aug,
                **inject)
            self.stats_dataset_w = self.latent_aug.stats_dataset_w
            self.num_ws = self.latent_aug.num_ws
            self.w_dim = self.latent_aug.w_dim
            self.z_dim = self.latent_aug.z_dim
        elif self.phase in ['val', 'test']:
            pass     # all augmentation disabled (reference :150-153)
        else:
            raise NotImplementedError

    @staticmethod
    def input_sanity_check(img):
        ok = isinstance(img, torch.Tensor) and img.dtype == torch.float32 and tuple(img.shape) == (1, 256, 256)
        assert ok, 'expected one float32 [1, 256, 256] slice per modality'

    output_sanity_check = input_sanity_check

    # ---- the dict protocol of the reference's drivers (reference :171-235)
    def set_input(self, data):
        """data = {'A', 'B': [B,1,r,r] float32 host tensors, 'A_paths', 'B_paths': per-sample file names (identical lists)}."""
        paths = data['A_paths']
        assert paths == data['B_paths'], 'paired modalities must come from the same slices'
        self.fname = paths
        self.real_A, self.real_B = data['A'], data['B']

    @property
    def real_AB(self):
        """The unaugmented pair as one [B,2,r,r] tensor (reference :189); joined on demand -- the augmenting branch never reads it."""
        return torch.cat([self.real_A, self.real_B], dim=1)

    def get_output(self):
        """The augmented pair on the host, one [B,1,r,r] tensor per modality, with the paths handed in."""
        both = self.real_AB_aug.detach()
        if both.is_cuda:
            # page-locked staging (torch's caching host allocator hands the block back once the caller drops the result): the
            # copy of a full gathered batch runs at the PCIe rate instead of the pageable one, which every rank of an N-rank run
            # pays N-fold -- each returns the WHOLE batch, as the reference's DataParallel gather does
            host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
            host.copy_(both, non_blocking=True)
            torch.cuda.synchronize(both.device)
            both = host
        out = {}
        for key, ch in (('A', 0), ('B', 1)):
            plane = both[:, ch:ch + 1]
            if self.lower_bound_clip and float(plane.min()) < -1:
                plane = plane.clamp(min=-1.0)
            out[key] = plane
        out['A_paths'] = out['B_paths'] = self.fname
        return out

    def _latent_dict(self, w):
        return {'w': w.detach().cpu().numpy().squeeze(), 'paths': '' if self.rand_aug else self.fname}

    def get_latent_output(self):
        return self._latent_dict(reverse_broadcasting(self.w_AB_aug))

    def get_latent_input(self):
        return self._latent_dict(self.w_AB)

    def forward(self):
        """One batch: with probability 1 - p_thres (training phase only) the latent optimisation, otherwise the inputs unchanged;
        the wall time of the call is appended to stats_time either way (the drivers average stats_time[1:])."""
        t0 = time.time()
        u = random.random()                      # drawn in every phase, as in the reference: the python RNG stream stays aligned
        augment = u > self.p_thres and self.phase == 'train'
        if not augment:
            self.real_AB_aug = self.real_AB
        elif self.rand_aug:
            z = self.sample_from_randn().to(self.device)
            self.real_AB_aug, self.w_AB_aug = self.latent_aug.forward_ganrand(z)
            self.w_AB = self.w_AB_aug
        else:
            if self.init_w != 'inv':
                raise NotImplementedError(f"init_w = {self.init_w!r}: only 'inv' (the inverted latent of each slice) is defined")
            self.w_AB = self.sample_from_inversion(self.fname).to(self.device)
            self.real_AB_aug, self.w_AB_aug = self.latent_aug(self.w_AB, self.fname)
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        dt = time.time() - t0
        if self.verbose_log:
            print('Augmentation completed in {:.0f}m {:.3f}s'.format(dt // 60, dt % 60))
        self.stats_time.append(dt)

    def sanity_check(self):
        """Shape / dtype checks around one forward, with the reference's two pictures (latent_aug.py:281-301): the first input pair as
        `<name>.png` and the first augmented pair as `<name>aug.png` in save_dir."""
        for t in (self.real_A[0], self.real_B[0]):
            self.input_sanity_check(t)
        visualize(self.real_A[0], self.real_B[0], _stem(self.fname[0]), self.save_dir)
        self.forward()
        out = self.get_output()
        for key in ('A', 'B'):
            self.output_sanity_check(out[key][0])
        visualize(out['A'][0], out['B'][0], _stem(out['A_paths'][0]) + 'aug', self.save_dir)

    def sample_from_randn(self):
        return torch.randn([self.batch_size, self.z_dim])

    def sample_from_inversion(self, fname):
        """Per-file inverted latent -> [len(fname), 1, w_dim] (reference :310-324; sized by the actual batch)."""
        rows = []
        for fn in fname:
            code = np.asarray(self.stats_dataset_w.lookup(fn), dtype=np.float32)
            rows.append(code.reshape(-1, self.w_dim)[:1])          # W space: row 0 of a [num_ws, w_dim] code, or the [w_dim] code itself
        w = torch.from_numpy(np.ascontiguousarray(np.stack(rows)))
        assert w.shape == (len(fname), 1, self.w_dim)
        return w



def _stem(path):
    return os.path.splitext(os.path.basename(str(path)))[0]


def visualize(imgA, imgB, img_name, save_dir):
    """The reference's `visualize` (latent_aug.py:327-341): modality A | modality B side by side as `<save_dir>/<img_name>.png`, grey
    levels stretched from the pair's minimum to its maximum as matplotlib's imshow does (8-bit PNG at the image's own resolution; the
    reference renders the same array through a matplotlib figure at 400 dpi)."""
    a = imgA.detach().cpu().numpy() if isinstance(imgA, torch.Tensor) else np.asarray(imgA)
    b = imgB.detach().cpu().numpy() if isinstance(imgB, torch.Tensor) else np.asarray(imgB)
    img = np.concatenate([a, b], axis=1) if a.ndim == 2 else np.concatenate([a[0], b[0]], axis=1)
    img = img.astype(np.float64)
    lo, hi = (float(np.nanmin(img)), float(np.nanmax(img))) if np.isfinite(img).any() else (0.0, 1.0)
    grey = np.clip((np.nan_to_num(img, nan=lo) - lo) / (hi - lo if hi > lo else 1.0), 0.0, 1.0)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        write_png_gray(os.path.join(save_dir, f'{img_name}.png'), np.round(grey * 255.0).astype(np.uint8))
