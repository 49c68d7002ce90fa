"""Blind noise-level estimation (reference model/nle.py): `noise_level(y, method="MAD")` / `nle_mad(y)`.

The estimate runs in libcdlnet_hip.so (cdl_nle_mad): depthwise stride-2 correlation with the 'bior4.4'
diagonal (HH) analysis filter and an exact per-sample median.  `nle_pca` (the reference's other method, a
CPU eigen-decomposition translated from MATLAB) is not part of the hot path and is not provided.
"""
import ctypes

import torch

from . import _lib, ops


def nle_mad(y):
    """Median-absolute-deviation estimate of the AWGN standard deviation of y (N,C,H,W), in y's own
    scale; returns (N,1,1,1) like the reference (model/nle.py:17-27)."""
    y = ops._dev(y, "y")
    if y.dim() != 4:
        raise ValueError("nle_mad expects (N, C, H, W)")
    N, C, H, W = y.shape
    lib = _lib.lib()
    n = int(lib.cdl_nle_mad_scratch_floats(N, C, H, W))
    if n == 0:
        raise ValueError("image smaller than the 10 x 10 wavelet filter")
    scratch = ops._scratch(y.device, n)
    out = torch.empty(N, device=y.device, dtype=torch.float32)
    rc = lib.cdl_nle_mad(ops._ptr(y), ops._ptr(out), ops._ptr(scratch), n, N, C, H, W, ops._stream())
    _lib.check(rc, "cdl_nle_mad")
    return out.reshape(-1, 1, 1, 1)


def noise_level(y, method="MAD", **kwargs):
    """model/nle.py:9-15."""
    if method in (True, "MAD", "wvlt"):
        return nle_mad(y)
    raise NotImplementedError(f"noise_level method {method!r}: only the MAD / wavelet estimator runs on the device")
