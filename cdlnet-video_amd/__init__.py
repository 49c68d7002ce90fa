"""cdlnet-video_amd: MI355X-native unrolled-ISTA hot path of RQLuo/CDLNet-video.

Import as `cdlnet_video_amd` (see the loader shim at the repository root; the directory name
carries a hyphen).  Public surface mirrors the reference's `model/net.py`.
"""
from . import _lib, nle, ops, parallel, train, utils
from ._lib import HipKernelError, HipLibraryMissing
from .gabor import ConvAdjoint2dGabor
from .net import ST, CDLNet, CDLNet_CSR, CDLNet_CSRf2, CDLNetVideo, GDLNet, ResidualBlock, prox_CSR, prox_CSR_f2
from .temporal import csr_inference_loop, csr_inference_v2
from .train import build_model, fit, init_model, load_ckpt, mcsure_loss, save_args, save_ckpt, train_step
from .utils import awgn, awgn3d, gen_bayer_mask, psnr

JDD_CDLNet = CDLNet      # BASELINE.json config 4: CDLNet(C=3) + Bayer mask

__all__ = ["CDLNet", "CDLNetVideo", "ResidualBlock", "GDLNet", "JDD_CDLNet", "CDLNet_CSR", "CDLNet_CSRf2", "prox_CSR",
           "prox_CSR_f2", "csr_inference_loop", "csr_inference_v2", "ConvAdjoint2dGabor", "ST",
           "build_model", "init_model", "load_ckpt", "save_ckpt", "train_step", "fit", "mcsure_loss", "save_args",
           "awgn", "awgn3d", "gen_bayer_mask", "psnr", "nle", "ops", "parallel", "train", "utils",
           "HipLibraryMissing", "HipKernelError"]
