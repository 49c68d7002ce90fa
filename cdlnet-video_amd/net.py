"""Drop-in modules for the reference's `model/net.py` hot path.

Same constructor signatures, attribute names (`A`, `B`, `D`, `t`, `g`, `K`, `M`, `P`, `s`, `t0`,
`adaptive`), `state_dict` keys and `forward(y, sigma=None, mask=1) -> (xhat, z)` contract as
`CDLNet` (net.py:16-104), `CDLNetVideo` (net.py:121-227) and `GDLNet` (net.py:569-687); the K
iterations run in hand-written HIP kernels (libcdlnet_hip.so) instead of ATen convolutions.

The filter containers subclass `nn.Conv{2,3}d` / `nn.ConvTranspose{2,3}d` purely as parameter
holders: identical `weight` shapes and state_dict keys, and -- because their default
initialisers draw from the global RNG exactly like the reference's -- `torch.manual_seed(s);
CDLNet(...)` yields the reference's weights bit for bit.  Their `forward` is overridden to
call the HIP operators; no ATen convolution runs on the device.
"""
import sys

import numpy as np
import torch
import torch.nn as nn

from . import loop, ops
from .gabor import ConvAdjoint2dGabor, filter_banks, gabor_kernel_cpu
from .solvers import gram_operator, power_method


def ST(x, t):
    """Shrinkage-thresholding sign(x)*relu(|x|-t) (net.py:11-14) on the device.

    Stand-alone form for callers of the reference's helper (t broadcastable to
    (N, M, 1, ..)); inside the nets it is fused into the analysis kernel's epilogue.
    """
    N, M = x.shape[:2]
    tau = torch.as_tensor(t, dtype=torch.float32, device=x.device)
    while tau.dim() < x.dim():
        tau = tau.unsqueeze(0)
    tau = torch.broadcast_to(tau, (N, M) + (1,) * (x.dim() - 2)).reshape(N, M).contiguous()
    return ops.shrink(x, tau)


# ------------------------------------------------------------------------------------------ banks
class _Analysis2d(nn.Conv2d):
    def forward(self, x):
        g = ops.Geometry.make(x.shape[0], self.in_channels, self.out_channels, x.shape[2:],
                              self.kernel_size, self.padding, self.stride[0])
        return ops.analysis(g, x, self.weight.detach())


class _Synthesis2d(nn.ConvTranspose2d):
    def forward(self, z):
        sp = tuple(d * self.stride[0] for d in z.shape[2:])
        g = ops.Geometry.make(z.shape[0], self.out_channels, self.in_channels, sp,
                              self.kernel_size, self.padding, self.stride[0])
        return ops.synthesis(g, z, self.weight.detach())


class _Analysis3d(nn.Conv3d):
    def forward(self, x):
        g = ops.Geometry.make(x.shape[0], self.in_channels, self.out_channels, x.shape[2:],
                              self.kernel_size, self.padding, self.stride[0])
        return ops.analysis(g, x, self.weight.detach())


class _Synthesis3d(nn.ConvTranspose3d):
    def forward(self, z):
        sp = tuple(d * self.stride[0] for d in z.shape[2:])
        g = ops.Geometry.make(z.shape[0], self.out_channels, self.in_channels, sp,
                              self.kernel_size, self.padding, self.stride[0])
        return ops.synthesis(g, z, self.weight.detach())


def _noise_scale(sigma, adaptive, N, device):
    """c = sigma/255 per sample as an (N,) tensor, or None for c = 0 (net.py:82)."""
    if sigma is None or not adaptive:
        return None
    if torch.is_tensor(sigma):
        c = sigma.to(device=device, dtype=torch.float32).reshape(-1) / 255.0
        if c.numel() == 1:
            c = c.expand(N)
        if c.numel() != N:
            raise ValueError("sigma must be a scalar or hold one value per sample")
        return c.contiguous()
    return torch.full((N,), float(sigma) / 255.0, device=device, dtype=torch.float32)


def _mask_tensor(mask, y):
    if torch.is_tensor(mask):
        return torch.broadcast_to(mask.to(dtype=torch.float32), y.shape).contiguous()
    if mask is None or mask == 1:
        return None
    raise ValueError("mask must be 1 (no mask) or a tensor shaped like y")


class _ISTANet(nn.Module):
    """Shared forward / generator plumbing."""

    def _filters(self):
        A = [m.weight for m in self.A]
        B = [m.weight for m in self.B]
        return A, B

    def _run(self, y, sigma, mask, all_codes):
        if not y.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward: input is on {y.device}. This package has no CPU "
                "compute path; the iterations run in HIP kernels on a ROCm device.")
        y = y.to(torch.float32)
        A, B = self._filters()
        c = _noise_scale(sigma, self.adaptive, y.shape[0], y.device)
        return loop.run(y, _mask_tensor(mask, y), c, self.t, A, B, self.s, all_codes)

    def forward(self, y, sigma=None, mask=1):
        """LISTA + D with noise-adaptive thresholds: returns (xhat, z_K)."""
        xhat, z = self._run(y, sigma, mask, False)
        return xhat, z

    def forward_generator(self, y, sigma=None, mask=1):
        """Yields z_1..z_K, then xhat (net.py:94-104, 214-227)."""
        outs = self._run(y, sigma, mask, True)
        xhat, zK, earlier = outs[0], outs[1], outs[2:]
        for z in earlier:
            yield z
        yield zK
        yield xhat


# ------------------------------------------------------------------------------------------ 2-D
class CDLNet(_ISTANet):
    """Convolutional Dictionary Learning Network (2-D; JDD = C=3 + Bayer mask)."""

    def __init__(self, K=3, M=64, P=7, s=1, C=1, t0=0, adaptive=False, init=True):
        super().__init__()
        if P % 2 == 0:
            raise ValueError("P must be odd (the reference's conv / conv-transpose pair needs it)")
        self.A = nn.ModuleList([_Analysis2d(C, M, P, stride=s, padding=(P - 1) // 2, bias=False)
                                for _ in range(K)])
        self.B = nn.ModuleList([_Synthesis2d(M, C, P, stride=s, padding=(P - 1) // 2,
                                             output_padding=s - 1, bias=False) for _ in range(K)])
        self.D = self.B[0]
        self.t = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self.g = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))       # unused, kept for checkpoints
        W = torch.randn(M, C, P, P)
        for k in range(K):
            self.A[k].weight.data = W.clone()
            self.B[k].weight.data = W.clone()
        if init:
            print("Running power-method on initial dictionary...")
            with torch.no_grad():
                op = gram_operator(self.A[0].weight, self.D.weight, s, (P - 1) // 2)
                L = power_method(op, torch.rand(1, C, 128, 128), num_iter=200, verbose=False)[0]
            print(f"Done. L={L:.3e}.")
            if L < 0:
                print("STOP: something is very very wrong...")
                sys.exit()
            for k in range(K):
                self.A[k].weight.data /= np.sqrt(L)
                self.B[k].weight.data /= np.sqrt(L)
        self.K, self.M, self.P, self.s, self.t0, self.adaptive = K, M, P, s, t0, adaptive

    @torch.no_grad()
    def project(self):
        """l2-ball projection of every filter, R+ projection of the thresholds (net.py:66-74)."""
        self.t.clamp_(0.0)
        ops.project_filter_banks_([m.weight.data for m in self.A] + [m.weight.data for m in self.B])

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts upstream CDLNet-OJSP checkpoints that predate the unused `g` parameter."""
        if "g" not in state_dict and "t" in state_dict:
            state_dict = dict(state_dict)
            state_dict["g"] = self.g.detach().clone()
        return super().load_state_dict(state_dict, strict=strict, **kw)


# ------------------------------------------------------------------------------------------ CSR
def prox_CSR(u, z_prev, lambd, gamma):
    """The reference's helper (net.py:229-242) on the device; lambd, gamma broadcastable to (N, M, 1, 1)."""
    return _prox(u, z_prev, None, lambd, gamma, None)


def prox_CSR_f2(u, z_prev, z_after, lambd, gamma1, gamma2):
    """The reference's helper (net.py:244-262) on the device."""
    return _prox(u, z_prev, z_after, lambd, gamma1, gamma2)


def _prox(u, zp, za, lam, g1, g2):
    N, M = u.shape[:2]
    sp = tuple(u.shape[2:])
    g = ops.Geometry.make(N, 1, M, sp, (1,) * len(sp), (0,) * len(sp), 1)

    def rows(t):
        t = torch.as_tensor(t, dtype=torch.float32, device=u.device)
        while t.dim() < u.dim():
            t = t.unsqueeze(0)
        return torch.broadcast_to(t, (N, M) + (1,) * len(sp)).reshape(N, M).contiguous()

    return ops.prox_csr(g, u, zp, rows(lam), rows(g1), za, rows(g2) if za is not None else None)


class _CSRBase(_ISTANet):
    """Constructor plumbing shared by the two CSR nets (2-D only, like the reference's)."""

    def _banks(self, K, M, P, s, C):
        A = nn.ModuleList([_Analysis2d(C, M, P, stride=s, padding=(P - 1) // 2, bias=False) for _ in range(K)])
        B = nn.ModuleList([_Synthesis2d(M, C, P, stride=s, padding=(P - 1) // 2, output_padding=s - 1,
                                        bias=False) for _ in range(K)])
        return A, B

    def _spectral_init(self, K, M, P, s, C, init):
        W = torch.randn(M, C, P, P)
        for k in range(K):
            self.A[k].weight.data = W.clone()
            self.B[k].weight.data = W.clone()
        if init:
            print("Running power-method on initial dictionary...")
            with torch.no_grad():
                op = gram_operator(self.A[0].weight, self.D.weight, s, (P - 1) // 2)
                L = power_method(op, torch.rand(1, C, 128, 128), num_iter=200, verbose=False)[0]
            print(f"Done. L={L:.3e}.")
            if L < 0:
                print("STOP: something is very very wrong...")
                sys.exit()
            for k in range(K):
                self.A[k].weight.data /= np.sqrt(L)
                self.B[k].weight.data /= np.sqrt(L)

    @torch.no_grad()
    def project(self):
        """net.py:419-424 / 518-523: only `t`, `A`, `B` are projected (not t2, g*, A2, B2)."""
        self.t.clamp_(0.0)
        ops.project_filter_banks_([m.weight.data for m in self.A] + [m.weight.data for m in self.B])

    def _prep(self, y, sigma, mask):
        if not y.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward: input is on {y.device}. This package has no CPU "
                "compute path; the iterations run in HIP kernels on a ROCm device.")
        y = y.to(torch.float32)
        return y, _mask_tensor(mask, y), _noise_scale(sigma, self.adaptive, y.shape[0], y.device)


class CDLNet_CSR(_CSRBase):
    """CDLNet with the frame-recurrent CSR prior (net.py:363-463): without a neighbour code the plain
    loop on the second bank (A2, B2, t2; the final synthesis is still D = B[0]); with `z_prev` the
    first bank and prox_CSR(., z_prev, t, g)."""

    def __init__(self, K=3, M=64, P=7, s=1, C=1, t0=0, adaptive=False, init=True):
        super().__init__()
        if P % 2 == 0:
            raise ValueError("P must be odd (the reference's conv / conv-transpose pair needs it)")
        self.A, self.B = self._banks(K, M, P, s, C)
        self.A2, self.B2 = self._banks(K, M, P, s, C)
        self.D = self.B[0]
        self.t = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self.t2 = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self.g = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self._spectral_init(K, M, P, s, C, init)
        self.K, self.M, self.P, self.s, self.t0, self.adaptive = K, M, P, s, t0, adaptive

    def forward(self, y, z_prev=None, sigma=None, mask=1):
        y, mask_t, c = self._prep(y, sigma, mask)
        if z_prev is None:
            A = [m.weight for m in self.A2]
            B = [self.B[0].weight] + [m.weight for m in self.B2][1:]      # B2[0] is never applied
            xhat, z = loop.run(y, mask_t, c, self.t2, A, B, self.s)[:2]
            return xhat, z
        A, B = self._filters()
        return loop.run_csr(y, mask_t, c, z_prev, None, self.t, self.g, None, A, B, self.s)


class CDLNet_CSRf2(_CSRBase):
    """CDLNet with previous- and next-frame CSR priors (net.py:464-568): one bank, thresholds t, g1, g2;
    the branch is chosen by which neighbour codes are given."""

    def __init__(self, K=3, M=64, P=7, s=1, C=1, t0=0, adaptive=False, init=True):
        super().__init__()
        if P % 2 == 0:
            raise ValueError("P must be odd (the reference's conv / conv-transpose pair needs it)")
        self.A, self.B = self._banks(K, M, P, s, C)
        self.D = self.B[0]
        self.t = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self.g1 = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self.g2 = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        self._spectral_init(K, M, P, s, C, init)
        self.K, self.M, self.P, self.s, self.t0, self.adaptive = K, M, P, s, t0, adaptive

    def forward(self, y, z_prev=None, z_after=None, sigma=None, mask=1):
        y, mask_t, c = self._prep(y, sigma, mask)
        A, B = self._filters()
        if z_prev is None and z_after is None:
            xhat, z = loop.run(y, mask_t, c, self.t, A, B, self.s)[:2]
            return xhat, z
        if z_prev is not None and z_after is not None:
            return loop.run_csr(y, mask_t, c, z_prev, z_after, self.t, self.g1, self.g2, A, B, self.s)
        if z_prev is not None:
            return loop.run_csr(y, mask_t, c, z_prev, None, self.t, self.g1, None, A, B, self.s)
        return loop.run_csr(y, mask_t, c, z_after, None, self.t, self.g2, None, A, B, self.s)


# ------------------------------------------------------------------------------------------ 3-D
class _DenseConv3d(nn.Conv3d):
    """Parameter holder of a ResidualBlock convolution; stand-alone forward = the HIP analysis operator."""

    def forward(self, x):
        g = ops.residual_geometry(x, self.weight)
        return ops.analysis(g, x, self.weight.detach())


class ResidualBlock(nn.Module):
    """net.py:105-120: relu(conv2(relu(conv1 x)) + x) with two bias-free 3x3x3 convolutions, on the HIP
    operators (forward and backward: loop.ResidualBlockFn).  Same constructor and state_dict keys."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3, 3), stride=1, padding=1):
        super().__init__()
        ks = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        if in_channels != out_channels or stride != 1 or any(k % 2 == 0 for k in ks) or pd != tuple(k // 2 for k in ks):
            raise ValueError("ResidualBlock: the skip connection needs in == out channels, stride 1 and "
                             "'same' padding (the only form the reference builds, net.py:150)")
        self.conv1 = _DenseConv3d(in_channels, out_channels, ks, stride=1, padding=pd, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _DenseConv3d(out_channels, out_channels, ks, stride=1, padding=pd, bias=False)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("ResidualBlock.forward: this package has no CPU compute path")
        return loop.ResidualBlockFn.apply(x.to(torch.float32), self.conv1.weight, self.conv2.weight)


class CDLNetVideo(_ISTANet):
    """3-D (video / volume) twin; `P` is (kD, kH, kW) or an int (cube)."""

    def __init__(self, K=3, M=64, P=(7, 7, 5), s=1, C=1, t0=0, adaptive=False, depth=3, init=True,
                 residual=False):
        super().__init__()
        if isinstance(P, int):
            P = (P, P, P)
        P = tuple(int(p) for p in P)
        if any(p % 2 == 0 for p in P):
            raise ValueError("every filter extent must be odd")
        pad = tuple(p // 2 for p in P)
        self.A = nn.ModuleList([_Analysis3d(C, M, P, stride=s, padding=pad, bias=False)
                                for _ in range(K)])
        self.B = nn.ModuleList([_Synthesis3d(M, C, P, stride=s, padding=pad, output_padding=s - 1,
                                             bias=False) for _ in range(K)])
        self.D = self.B[0]
        self.t = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1, 1))
        self.residual = bool(residual)
        if self.residual:                       # created here, as the reference does: same RNG draws
            self.residual_blocks = nn.ModuleList([ResidualBlock(M, M) for _ in range(K)])
        W = torch.randn(M, C, *P)
        for k in range(K):
            self.A[k].weight.data = W.clone()
            self.B[k].weight.data = W.clone()
        if init:
            print("Running power-method on initial dictionary...")
            with torch.no_grad():
                op = gram_operator(self.A[0].weight, self.D.weight, s, pad)
                L = power_method(op, torch.rand(1, C, depth, 128, 128), num_iter=200, verbose=False)[0]
            print(f"Done. L={L:.3e}.")
            if L < 0:
                print("STOP: something is very very wrong...")
                sys.exit()
            for k in range(K):
                self.A[k].weight.data /= np.sqrt(L)
                self.B[k].weight.data /= np.sqrt(L)
        self.K, self.M, self.P, self.s, self.t0, self.adaptive = K, M, P, s, t0, adaptive

    def _run(self, y, sigma, mask, all_codes):
        if not self.residual:
            return super()._run(y, sigma, mask, all_codes)
        if not y.is_cuda:
            raise RuntimeError(f"CDLNetVideo.forward: input is on {y.device}; this package has no CPU compute path")
        y = y.to(torch.float32)
        A, B = self._filters()
        c = _noise_scale(sigma, self.adaptive, y.shape[0], y.device)
        blocks = [(b.conv1.weight, b.conv2.weight) for b in self.residual_blocks]
        return loop.run_residual(y, _mask_tensor(mask, y), c, self.t, A, B, self.s, blocks, all_codes)

    def forward_generator(self, y, sigma=None, mask=1):
        """net.py:214-227; with residual blocks the yielded codes are the ST outputs (before the block)."""
        if not self.residual:
            yield from super().forward_generator(y, sigma, mask)
            return
        outs = self._run(y, sigma, mask, True)
        yield from outs[2:]
        yield outs[0]

    @torch.no_grad()
    def project(self):
        """net.py:184-190.  (The reference's own call raises on torch >= 2: `torch.norm` with a
        3-axis `dim`; this does what it evidently means -- the l2 norm over the filter volume.)"""
        self.t.clamp_(0.0)
        ops.project_filter_banks_([m.weight.data for m in self.A] + [m.weight.data for m in self.B])


# ------------------------------------------------------------------------------------------ Gabor
class GDLNet(_ISTANet):
    """Gabor Dictionary Learning Network: same loop, filters synthesised from Gabor parameters."""

    def __init__(self, K=3, M=64, P=7, s=1, C=1, t0=0, order=1, adaptive=False, shared="", init=True):
        super().__init__()
        self.A = nn.ModuleList([ConvAdjoint2dGabor(M, C, P, stride=s, order=order) for _ in range(K)])
        self.B = nn.ModuleList([ConvAdjoint2dGabor(M, C, P, stride=s, order=order) for _ in range(K)])
        self.D = self.B[0]
        self.t = nn.Parameter(t0 * torch.ones(K, 2, M, 1, 1))
        alpha = torch.randn(order, M, C, 1, 1)
        a = torch.randn(order, M, C, 2)
        w0 = torch.randn(order, M, C, 2)
        psi = torch.randn(order, M, C)
        for k in range(K):
            for bank in (self.A[k], self.B[k]):
                bank.alpha.data = alpha.clone()
                bank.a.data = a.clone()
                bank.w0.data = w0.clone()
                bank.psi.data = psi.clone()
            if k > 0:                                   # parameter sharing by aliasing (net.py:607-622)
                if "alpha" in shared:
                    self.A[k].alpha = self.A[0].alpha
                    if k > 1:                           # never share the scale with D = B[0]
                        self.B[k].alpha = self.B[1].alpha
                if "a_" in shared:
                    self.A[k].a, self.B[k].a = self.A[0].a, self.B[0].a
                if "w0" in shared:
                    self.A[k].w0, self.B[k].w0 = self.A[0].w0, self.B[0].w0
                if "psi" in shared:
                    self.A[k].psi, self.B[k].psi = self.A[0].psi, self.B[0].psi
        if init:
            print("Running power-method on initial dictionary...")
            with torch.no_grad():
                wa = gabor_kernel_cpu(self.A[0].alpha, self.A[0].a, self.A[0].w0, self.A[0].psi, P, True)
                wd = gabor_kernel_cpu(self.D.alpha, self.D.a, self.D.w0, self.D.psi, P, False)
                L = power_method(gram_operator(wa, wd, s, (P - 1) // 2), torch.rand(1, C, 128, 128),
                                 num_iter=200, verbose=False)[0]
            print(f"Done. L={L:.3e}.")
            if L < 0:
                print("STOP: something is very very wrong...")
                sys.exit()
            for k in range(K):                          # net.py:637-642
                self.A[k].alpha.data /= np.sqrt(L)
                self.B[k].alpha.data /= np.sqrt(L)
                if "alpha" in shared:
                    self.B[1].alpha.data /= np.sqrt(L)
                    break
        self.K, self.M, self.P, self.s, self.t0 = K, M, P, s, t0
        self.order, self.adaptive = order, adaptive

    def _filters(self):
        """All 2K banks from one launch and one autograd node (gabor.filter_banks); same values as the per-module
        `get_filter` calls of the reference (net.py:665-671)."""
        K = len(self.A)
        banks = filter_banks(list(self.A) + list(self.B), [True] * K + [False] * K)
        return banks[:K], banks[K:]

    @torch.no_grad()
    def project(self):
        self.t.clamp_(0.0)
