"""latentaugment_amd -- MI355X-native latent-optimisation hot path of ltronchin/LatentAugment.

Layout:  csrc/  HIP kernels + C ABI (liblatentaug_hip.so)   |  _lib.py  ctypes binding
         ops.py  mirror of the reference op layer            |  synthesis.py  G.synthesis engine
         latent_aug.py  LatentAug.forward mirror             |  augments/  create_augment() plugin API
Importing the package does not need a GPU; calling any op does (there is no CPU fallback).
"""
__version__ = '0.1.0'
