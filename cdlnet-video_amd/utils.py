"""Input-side helpers the reference's trainers feed the hot path with (reference utils.py:13-55).

These generate data (noise, Bayer masks); they are tensor bookkeeping, not part of the loop,
and work on whatever device the batch lives on.
"""
import math

import torch


def gen_bayer_mask(x):
    """RGGB sampling mask for a 3-channel (N,3,H,W) batch (utils.py:13-19)."""
    m = torch.zeros_like(x)
    m[:, 0, 0::2, 0::2] = 1
    m[:, 1, 0::2, 1::2] = 1
    m[:, 1, 1::2, 0::2] = 1
    m[:, 2, 1::2, 1::2] = 1
    return m


def gen_bayer_mask3d(x):
    """utils.py:21-27: every spatial site is sampled (all-ones on the last two axes)."""
    return torch.ones_like(x)


def _awgn(x, noise_std, generator=None):
    # The draws happen on the generator's device and in the reference's order (rand for sigma, then randn): with a
    # CPU generator (e.g. torch.default_generator) a GPU run sees the very noise the reference's CPU path draws.
    dev = generator.device if generator is not None else x.device
    if isinstance(noise_std, (list, tuple)):
        shape = (len(x),) + (1,) * (x.dim() - 1)
        sigma = noise_std[0] + (noise_std[1] - noise_std[0]) * torch.rand(
            shape, device=dev, generator=generator).to(x.device)
    else:
        sigma = noise_std
    noise = torch.randn(x.shape, device=dev, dtype=x.dtype, generator=generator).to(x.device)
    return x + noise * (sigma / 255), sigma


def awgn(input, noise_std, generator=None):
    """y = x + n*sigma/255, sigma fixed or U(lo,hi) per sample shaped (N,1,1,1) (utils.py:29-41)."""
    return _awgn(input, noise_std, generator)


def awgn3d(input, noise_std, generator=None):
    """Same for (N,C,D,H,W) clips, sigma shaped (N,1,1,1,1) (utils.py:43-55)."""
    return _awgn(input, noise_std, generator)


def psnr(x, xhat):
    """-10 log10 MSE with peak 1.0 (analyze.py:104, analyze3d.py:131-133)."""
    return -10.0 * math.log10(torch.mean((x - xhat) ** 2).item())


def synthetic_clip(shape, seed=0, device="cpu", waves=6):
    """Smooth synthetic content in [0,1]: a sum of random 3-D sinusoids, min-max normalised --
    the idea of the reference's syn_data/gen.py:12-31 without its PNG round trip."""
    gen = torch.Generator().manual_seed(seed)
    N, C = shape[:2]
    sp = shape[2:]
    grids = torch.meshgrid(*[torch.linspace(0, 1, n) for n in sp], indexing="ij")
    out = torch.zeros(shape)
    for n in range(N):
        for c in range(C):
            acc = torch.zeros(sp)
            for _ in range(waves):
                freq = torch.rand(len(sp), generator=gen) * 12 + 1
                phase = torch.rand(1, generator=gen) * 2 * math.pi
                amp = torch.rand(1, generator=gen) + 0.2
                arg = sum(f * g for f, g in zip(freq, grids))
                acc += amp * torch.sin(2 * math.pi * arg + phase)
            out[n, c] = (acc - acc.min()) / (acc.max() - acc.min())
    return out.to(device)
