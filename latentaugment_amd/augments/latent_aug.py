"""LatentAugment plugin: drop-in for the reference's augments/latent_aug.py::LatentAugment (:44-324).

Same command-line options (names, types, defaults: reference :57-96), same methods and attributes used by the drivers
(backbone_latentaug.py:86-124): set_input / forward / get_output / get_latent_input / get_latent_output / sanity_check /
stats_time.  The optimisation runs on the MI355X through latentaugment_amd.latent_aug.LatentAug.

Deliberate behavioural fixes: construction does not dereference `.module` (defect a); the batch is
sized by len(fname), so a last partial batch works (defect h).
"""
import random
import time

import numpy as np
import torch

from .. import latent_aug as util_latent_aug
from .base_aug import BaseAugment


def reverse_broadcasting(latent):
    return latent[:, :1, :]


def set_gpu_ids(gpu_ids):
    out = []
    for s in str(gpu_ids).split(','):
        if s.strip() != '' and int(s) >= 0:
            out.append(int(s))
    return out


class LatentAugment(BaseAugment):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--model_dir', help='Where to load the StyleGAN/MappingNetwork pretrained model', metavar='DIR', required=True)
        parser.add_argument('--interim_dir', help='Where to save/load the data', metavar='DIR', required=True)
        parser.add_argument('--gpu_ids_aug', type=str, default='0', help='gpu ids: e.g. 0  0,1,2, 0,2. use -1 for CPU')
        parser.add_argument('--dataset_aug', help='', metavar='DIR', default="Pelvis_2.1_repo_no_mask")
        parser.add_argument('--dataset_name_aug', help='', metavar='DIR', default="Pelvis_2.1_repo_no_mask-num-375_train-0.70_val-0.20_test-0.10")
        parser.add_argument('--modalities_aug', help='', metavar='DIR', default="MR_nonrigid_CT,MR_MR_T2")
        parser.add_argument('--img_resolution', help='Image resolution.', type=int, default=256)
        parser.add_argument('--exp_stylegan', help='', metavar='DIR', default="00003")
        parser.add_argument('--network_pkl_stylegan', help='', metavar='DIR', default="network-snapshot-005320.pkl")
        parser.add_argument('--dataset_w_name', help='', metavar='DIR', default="Pelvis_2.1_repo_no_mask-num-375_train-0.70_val-0.20_test-0.10-expinv_00001")
        parser.add_argument('--exp_inv', help='', metavar='DIR', default="00001")
        parser.add_argument('--network_pkl_inv', help='', metavar='DIR', default="")
        parser.add_argument('--truncation_psi', help='Truncation value.', type=float, default=1.0)
        parser.add_argument('--rand_aug', action='store_true', help='Compute only random GAN augmentation.')
        parser.add_argument('--lower_bound_clip', action='store_true', help='Clip the pixels values under -1 to -1.')
        parser.add_argument('--step_img', help='Selection step to create the image dataset from which compute the distances.', type=int, default=20)
        parser.add_argument('--step_w', help='Selection step to create the latent dataset from which compute the distances.', type=int, default=5)
        parser.add_argument('--lpips_script', help='How to extract the features manifold.', type=str, default='lpips_script')
        parser.add_argument('--opt_num_epochs', help='Number of optimization steps', type=int, default=10)
        parser.add_argument('--opt_lr', help='Learning rate of optimization algorithm', type=float, default=0.01)
        parser.add_argument('--init_w', help='Initialization point for latent codes [inv | random]', type=str, default='random')
        parser.add_argument('--crop_size_aug', help='Size of the crop applied to images.', type=int, default=64)
        parser.add_argument('--preprocess_aug', help='Type of preprocessing applied for augmentation pipeline [center_crop | random_crop | center_random_crop | original ]', type=str, default='center_random_crop')
        parser.add_argument('--w_pix', help='Weight of recontruction loss', type=float, default=1.0)
        parser.add_argument('--w_lpips', help='Weight of lpips loss', type=float, default=1.0)
        parser.add_argument('--w_latent', help='Weight of latent loss', type=float, default=1.0)
        parser.add_argument('--w_disc', help='Weight of discriminator loss.', type=float, default=1.0)
        parser.add_argument('--p_thres', help='Augmentation probability.', type=float, default=1.0)
        parser.add_argument('--soft_aug', help='Activate smooth augmentation via interpolation.', type=bool, default=False)
        parser.add_argument('--alpha', help='Value for linear interpolation in soft_aug.', type=float, default=1.0)
        parser.add_argument('--verbose_log', help='Print losses and time during the optimization process.', type=bool, default=False)
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
        assert isinstance(img, torch.Tensor)
        assert img.dtype == torch.float32
        assert img.shape == (1, 256, 256)

    output_sanity_check = input_sanity_check

    def set_input(self, data):
        assert data['A_paths'] == data['B_paths']
        self.real_A = data['A']
        self.real_B = data['B']
        self.fname = data['A_paths']
        self.real_AB = torch.cat((self.real_A, self.real_B), dim=1)

    def get_output(self):
        real_AB_aug = self.real_AB_aug.detach().cpu()
        real_A_aug = real_AB_aug[:, 0, :, :].unsqueeze(dim=1)
        real_B_aug = real_AB_aug[:, 1, :, :].unsqueeze(dim=1)
        if self.lower_bound_clip:
            if real_A_aug.min().item() < -1:
                real_A_aug = torch.clamp(real_A_aug, min=-1.0, max=None)
            if real_B_aug.min().item() < -1:
                real_B_aug = torch.clamp(real_B_aug, min=-1.0, max=None)
        return {'A': real_A_aug, 'B': real_B_aug, 'A_paths': self.fname, 'B_paths': self.fname}

    def get_latent_output(self):
        w_aug = reverse_broadcasting(self.w_AB_aug).detach().cpu().numpy().squeeze()
        return {'w': w_aug, 'paths': self.fname if not self.rand_aug else ''}

    def get_latent_input(self):
        w = self.w_AB.detach().cpu().numpy().squeeze()
        return {'w': w, 'paths': self.fname if not self.rand_aug else ''}

    def forward(self):
        since = time.time()
        if random.random() > self.p_thres and self.phase == 'train':
            if self.rand_aug:
                w_AB = self.sample_from_randn().to(self.device)
                self.real_AB_aug, self.w_AB_aug = self.latent_aug.forward_ganrand(w_AB)
                self.w_AB = self.w_AB_aug
            else:
                if self.init_w == 'inv':
                    self.w_AB = self.sample_from_inversion(self.fname)
                else:
                    raise NotImplementedError
                self.w_AB = self.w_AB.to(self.device)
                self.real_AB_aug, self.w_AB_aug = self.latent_aug(self.w_AB, self.fname)
        else:
            self.real_AB_aug = torch.cat((self.real_A, self.real_B), dim=1)
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        time_elapsed = time.time() - since
        if self.verbose_log:
            print('Augmentation completed in {:.0f}m {:.3f}s'.format(time_elapsed // 60, time_elapsed % 60))
        self.stats_time.append(time_elapsed)

    def sanity_check(self):
        self.input_sanity_check(self.real_A[0])
        self.input_sanity_check(self.real_B[0])
        self.forward()
        data = self.get_output()
        self.output_sanity_check(data['A'][0])
        self.output_sanity_check(data['B'][0])

    def sample_from_randn(self):
        return torch.randn([self.batch_size, self.z_dim])

    def sample_from_inversion(self, fname):
        """Per-file inverted latent -> [len(fname), 1, w_dim] (reference :310-324; sized by the actual batch)."""
        w = torch.empty([len(fname), self.num_ws, self.w_dim], dtype=torch.float32)
        for i, fn in enumerate(fname):
            out_w = np.asarray(self.stats_dataset_w.lookup(fn), dtype=np.float32)
            if out_w.ndim == 1:
                out_w = np.broadcast_to(out_w[None], (self.num_ws, self.w_dim))
            w[i] = torch.from_numpy(np.ascontiguousarray(out_w))
        w = reverse_broadcasting(w)
        assert w.shape == (len(fname), 1, self.w_dim)
        return w
