"""Abstract base of augment plugins; same surface as the reference's augments/base_aug.py:7-64."""
import os
from abc import ABC, abstractmethod

import torch


class BaseAugment(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, data):
        """Unpack input data from the dataloader."""

    @abstractmethod
    def forward(self):
        pass

    def get_train_transform(self):
        pass

    def get_valid_transform(self):
        pass

    def sanity_check(self):
        pass
