"""`model` package surface of VPTR (reference: model/__init__.py:1-3) backed by MI355X HIP kernels."""
from .criterion import GDL, temporal_weight_func, MSELoss, BiPatchNCE, L1Loss, GANLoss
from .modules import VPTREnc, VPTRDec, VPTRDisc, VPTRFormerNAR, VPTRFormerFAR
from .autoencoder import init_weights

__all__ = ["GDL", "temporal_weight_func", "MSELoss", "BiPatchNCE", "L1Loss", "GANLoss", "VPTREnc", "VPTRDec", "VPTRDisc",
           "VPTRFormerNAR", "VPTRFormerFAR", "init_weights"]
