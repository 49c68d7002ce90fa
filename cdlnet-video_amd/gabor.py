"""Gabor-parameterised filter bank (reference model/gabor.py).

`ConvAdjoint2dGabor` keeps the reference's parameter names and shapes
(alpha (order,M,C,1,1), a / w0 (order,M,C,2), psi (order,M,C)) so GDLNet state_dicts load
unchanged.  The filter bank is synthesised on the device by `cdl_gabor_filters` (one kernel
instead of the reference's ~10 ATen launches per call) with a hand-written adjoint.
"""
import torch
import torch.nn as nn

from . import ops


class _GaborBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, a, w0, psi, P, transpose):
        ctx.P, ctx.transpose = P, transpose
        ctx.save_for_backward(alpha, a, w0, psi)
        return ops.gabor_filters(alpha, a, w0, psi, P, transpose)

    @staticmethod
    def backward(ctx, dw):
        alpha, a, w0, psi = ctx.saved_tensors
        dal, da, dw0, dpsi = ops.gabor_filters_bwd(alpha, a, w0, psi, dw.contiguous(), ctx.P,
                                                   ctx.transpose)
        return dal, da, dw0, dpsi, None, None


class _GaborBanks(torch.autograd.Function):
    """(P, transposes, alpha_0, a_0, w0_0, psi_0, alpha_1, ...) -> every filter bank of a net, one launch each way.
    Aliased parameters (GDLNet's `shared`, net.py:607-622) simply appear several times among the inputs; autograd
    sums their gradients."""

    @staticmethod
    def forward(ctx, P, transposes, *flat):
        params = [tuple(flat[4 * k:4 * k + 4]) for k in range(len(flat) // 4)]
        ctx.P, ctx.transposes = P, tuple(transposes)
        ctx.save_for_backward(*flat)
        return tuple(ops.gabor_filter_banks(params, P, transposes))

    @staticmethod
    def backward(ctx, *dws):
        flat = ctx.saved_tensors
        params = [tuple(flat[4 * k:4 * k + 4]) for k in range(len(flat) // 4)]
        dws = [None if d is None else d.contiguous() for d in dws]
        grads = ops.gabor_filter_banks_bwd(params, dws, ctx.P, ctx.transposes)
        out = []
        for k, g in enumerate(grads):
            out.extend(g if dws[k] is not None else (None, None, None, None))
        return (None, None, *out)


def filter_banks(modules, transposes):
    """Filters of several ConvAdjoint2dGabor modules from one launch (and one autograd node)."""
    flat = []
    for m in modules:
        if not m.psi.is_cuda:
            raise RuntimeError("ConvAdjoint2dGabor: parameters are on the CPU; the filter synthesis kernel runs on "
                               "the ROCm device only (use .to('cuda'))")
        flat += [m.alpha, m.a, m.w0, m.psi]
    return list(_GaborBanks.apply(modules[0].ks, tuple(bool(t) for t in transposes), *flat))


def gabor_kernel_cpu(alpha, a, w0, psi, ks, transpose=False):
    """Init-time CPU evaluation of the same formula (gabor.py:7-28,46-51) for the constructor's
    power method; the device path is `cdl_gabor_filters`."""
    if transpose:
        w0, psi = -w0, -psi
    ax = torch.arange(ks, dtype=alpha.dtype) - (ks - 1) / 2.0
    gy, gx = torch.meshgrid(ax, ax, indexing="ij")
    env = torch.exp(-((a[..., 0, None, None] * gy) ** 2 + (a[..., 1, None, None] * gx) ** 2))
    car = torch.cos(w0[..., 0, None, None] * gy + w0[..., 1, None, None] * gx + psi[..., None, None])
    return (alpha * env * car).sum(dim=0)


class ConvAdjoint2dGabor(nn.Module):
    """Convolution pair with a mixture-of-Gabor kernel: `.T(x)` analyses, `forward(z)` synthesises."""

    def __init__(self, nic, noc, ks, stride=2, order=1):
        super().__init__()
        self.alpha = nn.Parameter(torch.randn((order, nic, noc, 1, 1)))
        self.a = nn.Parameter(torch.randn((order, nic, noc, 2)))
        self.w0 = nn.Parameter(torch.randn((order, nic, noc, 2)))
        self.psi = nn.Parameter(torch.randn((order, nic, noc)))
        self.order, self.stride, self.ks = order, stride, ks
        p = (ks - 1) // 2
        self._pad = (p, p, p, p)
        # The reference builds a throw-away ConvTranspose2d(1,1,ks) here (gabor.py:44); doing
        # the same keeps the global RNG stream -- and therefore seeded inits -- identical.
        nn.ConvTranspose2d(1, 1, ks, stride=stride)

    def get_filter(self, transpose=False):
        if self.psi.is_cuda:
            return _GaborBank.apply(self.alpha, self.a, self.w0, self.psi, self.ks, bool(transpose))
        raise RuntimeError("ConvAdjoint2dGabor.get_filter: parameters are on the CPU; the filter "
                           "synthesis kernel runs on the ROCm device only (use .to('cuda'))")

    def _geom(self, N, H, W):
        p = self._pad[0]
        M, C = self.psi.shape[1:]
        return ops.Geometry.make(N, C, M, (H, W), (self.ks, self.ks), (p, p), self.stride)

    def T(self, x):
        """Analysis: zero-pad + strided correlation with the (-w0,-psi) filter (gabor.py:53-55)."""
        g = self._geom(x.shape[0], x.shape[2], x.shape[3])
        return ops.analysis(g, x, self.get_filter(transpose=True).detach())

    def forward(self, x):
        """Synthesis: transposed correlation back to s*H x s*W (gabor.py:57-67)."""
        g = self._geom(x.shape[0], x.shape[2] * self.stride, x.shape[3] * self.stride)
        return ops.synthesis(g, x, self.get_filter().detach())
