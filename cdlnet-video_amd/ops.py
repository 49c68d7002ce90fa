"""Tensor-level wrappers over the C ABI (include/cdlnet_hip.h).

PyTorch supplies device memory and the stream; every operation here is a call into
libcdlnet_hip.so.  Inputs must be CUDA (ROCm) fp32 tensors -- there is no CPU path.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib


def _stream():
    """Stream of the CURRENT device: every public function of this module runs under `_on_tensor_device`,
    which makes the tensors' device current first, so this is the tensors' stream."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _first_cuda_tensor(args, kwargs=None):
    """First CUDA tensor among the arguments, searching nested lists / tuples to any depth (gabor_filter_banks takes
    a list of 4-tuples of tensors)."""
    for a in list(args) + (list(kwargs.values()) if kwargs else []):
        if torch.is_tensor(a):
            if a.is_cuda:
                return a
        elif isinstance(a, (list, tuple)):
            t = _first_cuda_tensor(a)
            if t is not None:
                return t
    return None


def _on_tensor_device(fn):
    """Run `fn` with its first CUDA tensor argument's device current (net.to('cuda:1') without a
    torch.cuda.set_device(1) must not hand cuda:1 pointers to a stream of cuda:0).  `_dev` then rejects
    any further tensor that lives elsewhere."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = _first_cuda_tensor(args, kwargs)
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapper


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: cdlnet-video_amd computes only through its HIP kernels "
            "(libcdlnet_hip.so) on a ROCm device; move the module and inputs to 'cuda'.")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{name} is on {t.device} but this call runs on cuda:{torch.cuda.current_device()}: "
                           "all tensors of one call must live on one device")
    return t.contiguous()


def _opt(t, name):
    return None if t is None else _dev(t, name)


def _three(v, fill):
    v = [int(x) for x in v]
    return [fill] * (3 - len(v)) + v


@dataclass(frozen=True)
class Geometry:
    """Operator geometry in the 3-D convention of `cdl_geom` (2-D: leading axis is 1)."""
    N: int
    C: int
    M: int
    dims: Tuple[int, int, int]        # padded image extents D,H,W
    P: Tuple[int, int, int]
    pad: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    ndim: int                         # 2 or 3: how tensors are shaped on the Python side

    @staticmethod
    def make(N, C, M, spatial, P, pad, stride):
        nd = len(spatial)
        stride = [stride] * nd if isinstance(stride, int) else list(stride)
        g = Geometry(int(N), int(C), int(M), tuple(_three(spatial, 1)), tuple(_three(P, 1)),
                     tuple(_three(pad, 0)), tuple(_three(stride, 1)), nd)
        for d, s in zip(g.dims, g.stride):
            if d % s:
                raise ValueError(f"image extent {d} is not a multiple of stride {s}")
        for p, q in zip(g.P, g.pad):
            if p % 2 == 0 or q != p // 2:
                raise ValueError("filters must have odd extent with padding P//2 "
                                 "(the reference's conv/conv-transpose pair only closes for odd P)")
        return g

    @property
    def code_spatial(self):
        return tuple(d // s for d, s in zip(self.dims, self.stride))[3 - self.ndim:]

    @property
    def image_spatial(self):
        return self.dims[3 - self.ndim:]

    def c_struct(self):
        return _lib.Geom(self.N, self.C, self.M, *self.dims, *self.P, *self.pad, *self.stride)

    def code_shape(self):
        return (self.N, self.M) + self.code_spatial

    def image_shape(self):
        return (self.N, self.C) + self.image_spatial

    def filter_shape(self):
        return (self.M, self.C) + self.P[3 - self.ndim:]


def _pads6(pads, ndim):
    """F.pad-ordered (last dim first) pads -> {d_lo,d_hi,h_lo,h_hi,w_lo,w_hi} int array."""
    per_axis = [(pads[2 * i], pads[2 * i + 1]) for i in range(ndim)][::-1]   # first spatial dim first
    flat = [0, 0] * (3 - ndim) + [v for lo_hi in per_axis for v in lo_hi]
    return (ctypes.c_int * 6)(*flat)


def split_pad(length, s):
    """(before, after) making `length` a multiple of s, floor/ceil split (model/utils.py:35-44)."""
    rem = length % s
    if rem == 0:
        return (0, 0)
    extra = s - rem
    return (extra // 2, extra - extra // 2)


def stride_pads(spatial, s):
    """F.pad-ordered tuple (model/utils.py:46-51, 103-111)."""
    out = []
    for length in reversed(tuple(spatial)):
        out.extend(split_pad(int(length), s))
    return tuple(out)


# ------------------------------------------------------------------------------------------
def preprocess(y, s, mask=None):
    """model/utils.py:5-22 / 70-87 on the device: returns (yp, mean(N,), pads, mask_p)."""
    y = _dev(y, "y")
    mask = _opt(mask, "mask")
    if mask is not None and mask.shape != y.shape:
        raise ValueError("mask must have the shape of y")
    nd = y.dim() - 2
    if nd not in (2, 3):
        raise ValueError("expected (N,C,H,W) or (N,C,D,H,W)")
    N, C = y.shape[:2]
    sp = _three(y.shape[2:], 1)
    pads = stride_pads(y.shape[2:], s)
    p6 = _pads6(pads, nd)
    padded = tuple(int(d) + p6[2 * i] + p6[2 * i + 1] for i, d in enumerate(sp))[3 - nd:]
    yp = torch.empty((N, C) + padded, device=y.device, dtype=torch.float32)
    mask_p = torch.empty_like(yp) if mask is not None else None
    mean = torch.empty(N, device=y.device, dtype=torch.float32)
    rc = _lib.lib().cdl_preprocess(_ptr(y), _ptr(mask), _ptr(yp), _ptr(mask_p), _ptr(mean),
                                   N, C, *sp, p6, _stream())
    _lib.check(rc, "cdl_preprocess")
    return yp, mean, pads, mask_p


def postprocess(xp, mean, pads):
    """model/utils.py:24-33 / 89-101: crop the stride padding and add the mean back."""
    xp = _dev(xp, "xp")
    nd = xp.dim() - 2
    N, C = xp.shape[:2]
    p6 = _pads6(pads, nd)
    spp = _three(xp.shape[2:], 1)
    sp = [spp[i] - p6[2 * i] - p6[2 * i + 1] for i in range(3)]
    out = torch.empty((N, C) + tuple(sp[3 - nd:]), device=xp.device, dtype=torch.float32)
    rc = _lib.lib().cdl_postprocess(_ptr(xp), _ptr(mean), _ptr(out), N, C, *sp, p6, _stream())
    _lib.check(rc, "cdl_postprocess")
    return out


def postprocess_bwd(gxhat, pads):
    gxhat = _dev(gxhat, "grad_xhat")
    nd = gxhat.dim() - 2
    N, C = gxhat.shape[:2]
    p6 = _pads6(pads, nd)
    sp = _three(gxhat.shape[2:], 1)
    spp = [sp[i] + p6[2 * i] + p6[2 * i + 1] for i in range(3)]
    out = torch.empty((N, C) + tuple(spp[3 - nd:]), device=gxhat.device, dtype=torch.float32)
    rc = _lib.lib().cdl_postprocess_bwd(_ptr(gxhat), _ptr(out), N, C, *sp, p6, _stream())
    _lib.check(rc, "cdl_postprocess_bwd")
    return out


def thresholds(t, c, N):
    """tau (K,N,M) = t[k,0] + c[n]*t[k,1]; t is the (K,2,M,1,1[,1]) parameter."""
    t = _dev(t, "t")
    K, two, M = t.shape[:3]
    c = _opt(c, "c")
    tau = torch.empty((K, N, M), device=t.device, dtype=torch.float32)
    rc = _lib.lib().cdl_thresholds(_ptr(t), _ptr(c), _ptr(tau), K, N, M, _stream())
    _lib.check(rc, "cdl_thresholds")
    return tau


def shrink(x, tau):
    """ST(x, tau) with tau (N,M) broadcast over the spatial axes (net.py:11-14)."""
    x, tau = _dev(x, "x"), _dev(tau, "tau")
    rows = x.shape[0] * x.shape[1]
    assert tau.numel() == rows
    out = torch.empty_like(x)
    rc = _lib.lib().cdl_shrink(_ptr(x), _ptr(tau), _ptr(out), rows, x.numel() // rows, _stream())
    _lib.check(rc, "cdl_shrink")
    return out


def analysis(g: Geometry, x, w, alpha=1.0, zin=None, gate=None, tau=None, out=None):
    x, w = _dev(x, "x"), _dev(w, "w")
    zin, gate, tau = _opt(zin, "zin"), _opt(gate, "gate"), _opt(tau, "tau")
    assert tuple(x.shape) == g.image_shape(), (x.shape, g.image_shape())
    assert tuple(w.shape) == g.filter_shape(), (w.shape, g.filter_shape())
    if out is None:
        out = torch.empty(g.code_shape(), device=x.device, dtype=torch.float32)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_analysis_workspace_floats(ctypes.byref(gs)))
    ws = _scratch(x.device, n) if n else None
    rc = _lib.lib().cdl_analysis_ws(ctypes.byref(gs), _ptr(x), _ptr(w), float(alpha), _ptr(zin),
                                    _ptr(gate), _ptr(tau), _ptr(out), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_analysis_ws")
    return out


def synthesis(g: Geometry, z, w, alpha=1.0, gate=None, mask=None, sub=None, out=None):
    z, w = _dev(z, "z"), _dev(w, "w")
    gate, mask, sub = _opt(gate, "gate"), _opt(mask, "mask"), _opt(sub, "sub")
    assert tuple(z.shape) == g.code_shape(), (z.shape, g.code_shape())
    assert tuple(w.shape) == g.filter_shape(), (w.shape, g.filter_shape())
    if out is None:
        out = torch.empty(g.image_shape(), device=z.device, dtype=torch.float32)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_synthesis_workspace_floats(ctypes.byref(gs)))
    ws = _scratch(z.device, n) if n else None
    rc = _lib.lib().cdl_synthesis_ws(ctypes.byref(gs), _ptr(z), _ptr(gate), _ptr(w), float(alpha),
                                     _ptr(mask), _ptr(sub), _ptr(out), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_synthesis_ws")
    return out


_WGRAD_WS = {}
_SCRATCH = {}


def _scratch(device, n):
    """One grow-only scratch buffer per (device, STREAM) for kernels that need transient partial sums: launches on one
    stream are ordered, so they may share it; two streams each get their own (a single per-device buffer would be a
    silent race between concurrent streams)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = _SCRATCH[key] = torch.empty(n, device=device, dtype=torch.float32)
    return buf


def wgrad(g: Geometry, z, x, alpha=1.0, gate=None):
    z, x, gate = _dev(z, "z"), _dev(x, "x"), _opt(gate, "gate")
    dw = torch.empty(g.filter_shape(), device=z.device, dtype=torch.float32)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_wgrad_workspace_floats(ctypes.byref(gs)))
    key = (z.device, torch.cuda.current_stream(z.device).cuda_stream, n)
    ws = _WGRAD_WS.get(key)                       # one scratch buffer per (device, stream, size), reused
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(max(n, 1), device=z.device, dtype=torch.float32)
    rc = _lib.lib().cdl_wgrad(ctypes.byref(gs), _ptr(z), _ptr(gate), _ptr(x), float(alpha), _ptr(dw),
                              _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_wgrad")
    return dw


def wgrad_pair(g: Geometry, z0, x0, alpha0, z1, x1, alpha1):
    """(alpha0 * z0 (x) x0, alpha1 * z1 (x) x1): the two ungated filter gradients of one reverse iteration, one
    launch where the matrix-core kernel covers the shape (cdl_wgrad_pair)."""
    z0, x0, z1, x1 = _dev(z0, "z0"), _dev(x0, "x0"), _dev(z1, "z1"), _dev(x1, "x1")
    dw0 = torch.empty(g.filter_shape(), device=z0.device, dtype=torch.float32)
    dw1 = torch.empty_like(dw0)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_wgrad_workspace_floats(ctypes.byref(gs)))
    key = (z0.device, torch.cuda.current_stream(z0.device).cuda_stream, n)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(max(n, 1), device=z0.device, dtype=torch.float32)
    rc = _lib.lib().cdl_wgrad_pair(ctypes.byref(gs), _ptr(z0), _ptr(x0), float(alpha0), _ptr(dw0), _ptr(z1), _ptr(x1),
                                   float(alpha1), _ptr(dw1), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_wgrad_pair")
    return dw0, dw1


def tau_grad(g: Geometry, gup, zout, c, dt_k):
    """Writes the (2,M) slice `dt_k` of the threshold gradient for one iteration."""
    gup, zout, c = _dev(gup, "g"), _dev(zout, "zout"), _opt(c, "c")
    scratch = torch.empty(16 * g.N * g.M, device=gup.device, dtype=torch.float32)      # CDL_TAU_SPLITS * N * M
    assert dt_k.is_contiguous() and dt_k.numel() == 2 * g.M
    gs = g.c_struct()
    base = dt_k.data_ptr()
    rc = _lib.lib().cdl_tau_grad(ctypes.byref(gs), _ptr(gup), _ptr(zout), _ptr(c),
                                 ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * g.M),
                                 _ptr(scratch), _stream())
    _lib.check(rc, "cdl_tau_grad")


def analysis_rev(g: Geometry, x, w, alpha, zin, zsup, c, dt_k, out=None):
    """One reverse-sweep step (cdl_analysis_rev_ws): out = [zsup != 0] (zin + alpha A x), and the (2, M) threshold
    gradient slice `dt_k` of `out` with respect to the code `zsup` -- cdl_analysis followed by cdl_tau_grad_gate, as one
    fat launch where the matrix-core analysis covers the geometry."""
    x, w, zin, zsup, c = _dev(x, "x"), _dev(w, "w"), _opt(zin, "zin"), _dev(zsup, "zsup"), _opt(c, "c")
    if out is None:
        out = _new(g.code_shape(), x.device)
    assert dt_k.is_contiguous() and dt_k.numel() == 2 * g.M
    gs = g.c_struct()
    n = int(_lib.lib().cdl_analysis_rev_workspace_floats(ctypes.byref(gs)))
    ws = _scratch(x.device, n)
    base = dt_k.data_ptr()
    rc = _lib.lib().cdl_analysis_rev_ws(ctypes.byref(gs), _ptr(x), _ptr(w), float(alpha), _ptr(zin), _ptr(zsup), _ptr(c),
                                        ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * g.M), _ptr(out), _ptr(ws), n,
                                        _stream())
    _lib.check(rc, "cdl_analysis_rev_ws")
    return out


def prox_csr(g: Geometry, u, z_prev, lam, gam1, z_after=None, gam2=None, out=None):
    """prox_CSR (z_after None) / prox_CSR_f2 of net.py:229-262; lam, gam* are (N,M) like tau."""
    u, z_prev, lam, gam1 = _dev(u, "u"), _dev(z_prev, "z_prev"), _dev(lam, "lam"), _dev(gam1, "gam1")
    z_after, gam2 = _opt(z_after, "z_after"), _opt(gam2, "gam2")
    for name, t in (("u", u), ("z_prev", z_prev), ("z_after", z_after)):
        if t is not None and tuple(t.shape) != g.code_shape():
            raise ValueError(f"{name}: shape {tuple(t.shape)} is not the code shape {g.code_shape()}")
    for name, t in (("lam", lam), ("gam1", gam1), ("gam2", gam2)):
        if t is not None and t.numel() != g.N * g.M:
            raise ValueError(f"{name}: expected {g.N * g.M} per-(sample, channel) thresholds")
    if out is None:
        out = torch.empty_like(u)
    gs = g.c_struct()
    rc = _lib.lib().cdl_prox_csr(ctypes.byref(gs), _ptr(u), _ptr(z_prev), _ptr(z_after), _ptr(lam), _ptr(gam1),
                                 _ptr(gam2), _ptr(out), _stream())
    _lib.check(rc, "cdl_prox_csr")
    return out


def analysis_prox(g: Geometry, x, w, alpha, zin, z_prev, lam, gam1, z_after=None, gam2=None, u_out=None, out=None):
    """cdl_analysis with the CSR map as epilogue: returns z = prox(zin + alpha*corr(x; w)); `u_out`
    (a code-shaped tensor) receives the pre-shrinkage value when given."""
    x, w = _dev(x, "x"), _dev(w, "w")
    zin, z_prev, z_after = _opt(zin, "zin"), _dev(z_prev, "z_prev"), _opt(z_after, "z_after")
    assert tuple(x.shape) == g.image_shape(), (x.shape, g.image_shape())
    assert tuple(w.shape) == g.filter_shape(), (w.shape, g.filter_shape())
    for name, t in (("zin", zin), ("z_prev", z_prev), ("z_after", z_after), ("u_out", u_out)):
        if t is not None and (tuple(t.shape) != g.code_shape() or not t.is_contiguous()):
            raise ValueError(f"{name}: expected a contiguous tensor of the code shape {g.code_shape()}")
    if out is None:
        out = torch.empty(g.code_shape(), device=x.device, dtype=torch.float32)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_analysis_workspace_floats(ctypes.byref(gs)))
    ws = _scratch(x.device, n) if n else None
    rc = _lib.lib().cdl_analysis_prox_ws(ctypes.byref(gs), _ptr(x), _ptr(w), float(alpha), _ptr(zin), _ptr(z_prev),
                                         _ptr(z_after), _ptr(_dev(lam, "lam")), _ptr(_dev(gam1, "gam1")),
                                         _ptr(_opt(gam2, "gam2")), _ptr(u_out), _ptr(out), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_analysis_prox_ws")
    return out


def prox_csr_bwd(g: Geometry, gz, u, z_prev, lam, gam1, c, dlam, dgam1, z_after=None, gam2=None, dgam2=None,
                 gz_prev=None, gz_after=None, out=None):
    """Reverse of prox_csr: returns gu; accumulates into gz_prev / gz_after; writes the (2,M) slices
    dlam, dgam1[, dgam2] of the threshold gradients of this iteration."""
    gz, u, z_prev = _dev(gz, "gz"), _dev(u, "u"), _dev(z_prev, "z_prev")
    z_after, gam2, c = _opt(z_after, "z_after"), _opt(gam2, "gam2"), _opt(c, "c")
    for t in (dlam, dgam1, dgam2):
        assert t is None or (t.is_contiguous() and t.numel() == 2 * g.M)
    for t in (gz_prev, gz_after):
        assert t is None or (t.is_contiguous() and tuple(t.shape) == g.code_shape())
    if out is None:
        out = torch.empty_like(gz)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_prox_csr_scratch_floats(ctypes.byref(gs)))
    scratch = torch.empty(n, device=gz.device, dtype=torch.float32)
    rc = _lib.lib().cdl_prox_csr_bwd(ctypes.byref(gs), _ptr(gz), _ptr(u), _ptr(z_prev), _ptr(z_after),
                                     _ptr(_dev(lam, "lam")), _ptr(_dev(gam1, "gam1")), _ptr(gam2), _ptr(c),
                                     _ptr(out), _ptr(gz_prev), _ptr(gz_after), _ptr(dlam), _ptr(dgam1),
                                     _ptr(dgam2), _ptr(scratch), n, _stream())
    _lib.check(rc, "cdl_prox_csr_bwd")
    return out


def residual_geometry(x, w):
    """Geometry of a ResidualBlock conv (net.py:105-120): C == M channels, unit stride, padding P//2."""
    M = int(w.shape[0])
    if tuple(w.shape[:2]) != (M, M) or x.shape[1] != M:
        raise ValueError(f"ResidualBlock expects (M,M,...) filters and M-channel input, got {tuple(w.shape)} / {tuple(x.shape)}")
    P = tuple(w.shape[2:])
    return Geometry.make(x.shape[0], M, M, x.shape[2:], P, tuple(p // 2 for p in P), 1)


def residual_forward(g: Geometry, x, w1, w2):
    """(h, out) = (relu(conv1 x), relu(conv2 h + x)); h is what the reverse pass keeps."""
    x, w1, w2 = _dev(x, "x"), _dev(w1, "w1"), _dev(w2, "w2")
    assert tuple(x.shape) == g.code_shape() and tuple(w1.shape) == g.filter_shape() == tuple(w2.shape)
    h, out = torch.empty_like(x), torch.empty_like(x)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_residual_scratch_floats(ctypes.byref(gs)))
    ws = _scratch(x.device, n) if n else None
    rc = _lib.lib().cdl_residual_forward(ctypes.byref(gs), _ptr(x), _ptr(w1), _ptr(w2), _ptr(h), _ptr(out),
                                         _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_residual_forward")
    return h, out


def residual_backward(g: Geometry, x, h, out, w1, w2, g_out):
    """(dx, dw1, dw2) for g_out = dL/dout."""
    x, h, out = _dev(x, "x"), _dev(h, "h"), _dev(out, "out")
    w1, w2, g_out = _dev(w1, "w1"), _dev(w2, "w2"), _dev(g_out, "g_out")
    new = lambda t: torch.empty(t.shape, device=t.device, dtype=torch.float32)      # contiguous, whatever t's strides
    dx, dh = new(x), new(x)
    dw1, dw2 = new(w1), new(w2)
    gs = g.c_struct()
    n = int(_lib.lib().cdl_residual_scratch_floats(ctypes.byref(gs)))
    ws = _scratch(x.device, n) if n else None
    rc = _lib.lib().cdl_residual_backward(ctypes.byref(gs), _ptr(x), _ptr(h), _ptr(out), _ptr(w1), _ptr(w2),
                                          _ptr(g_out), _ptr(dx), _ptr(dw1), _ptr(dw2), _ptr(dh), _ptr(ws), n,
                                          _stream())
    _lib.check(rc, "cdl_residual_backward")
    return dx, dw1, dw2


def project_filters_(w):
    """In-place unit-ball projection of every (m,c) filter of w (M,C,*P)."""
    if not w.is_cuda:
        raise RuntimeError("project(): parameters must live on the ROCm device (no CPU path)")
    assert w.dtype == torch.float32 and w.is_contiguous()
    nf = w.shape[0] * w.shape[1]
    rc = _lib.lib().cdl_project_filters(_ptr(w), nf, w.numel() // nf, _stream())
    _lib.check(rc, "cdl_project_filters")
    return w


def project_filter_banks_(banks):
    """In-place unit-ball projection of every filter of every bank (same-shaped (M,C,*P) tensors): one launch."""
    banks = list(banks)
    if not banks:
        return
    w0 = banks[0]
    for w in banks:
        if not w.is_cuda:
            raise RuntimeError("project(): parameters must live on the ROCm device (no CPU path)")
        assert w.dtype == torch.float32 and w.is_contiguous() and w.shape == w0.shape
    nf = w0.shape[0] * w0.shape[1]
    rc = _lib.lib().cdl_project_filter_banks(_ptr_table(banks), len(banks), nf, w0.numel() // nf, _stream())
    _lib.check(rc, "cdl_project_filter_banks")


def gabor_filters(alpha, a, w0, psi, P, transpose):
    alpha, a, w0, psi = (_dev(v, n) for v, n in ((alpha, "alpha"), (a, "a"), (w0, "w0"), (psi, "psi")))
    order, M, C = psi.shape
    w = torch.empty((M, C, P, P), device=psi.device, dtype=torch.float32)
    rc = _lib.lib().cdl_gabor_filters(_ptr(alpha), _ptr(a), _ptr(w0), _ptr(psi), _ptr(w), order, M, C,
                                      P, int(bool(transpose)), _stream())
    _lib.check(rc, "cdl_gabor_filters")
    return w


def gabor_filters_bwd(alpha, a, w0, psi, dw, P, transpose):
    alpha, a, w0, psi, dw = (_dev(v, n) for v, n in ((alpha, "alpha"), (a, "a"), (w0, "w0"),
                                                      (psi, "psi"), (dw, "dw")))
    order, M, C = psi.shape
    outs = [torch.empty_like(v) for v in (alpha, a, w0, psi)]
    rc = _lib.lib().cdl_gabor_filters_bwd(_ptr(alpha), _ptr(a), _ptr(w0), _ptr(psi), _ptr(dw),
                                          *(_ptr(o) for o in outs), order, M, C, P,
                                          int(bool(transpose)), _stream())
    _lib.check(rc, "cdl_gabor_filters_bwd")
    return outs


def gabor_filter_banks(params, P, transposes):
    """All banks of a net in one launch.  params: [(alpha, a, w0, psi)] per bank; returns the list of (M,C,P,P)
    filters (views of one allocation)."""
    nb = len(params)
    params = [tuple(_dev(v, n) for v, n in zip(pr, ("alpha", "a", "w0", "psi"))) for pr in params]
    order, M, C = params[0][3].shape
    out = torch.empty((nb, M, C, P, P), device=params[0][3].device, dtype=torch.float32)
    ws = list(out.unbind(0))
    flags = (ctypes.c_int * nb)(*[int(bool(t)) for t in transposes])
    rc = _lib.lib().cdl_gabor_filter_banks(nb, *(_ptr_table([pr[i] for pr in params]) for i in range(4)), flags,
                                           _ptr_table(ws), order, M, C, P, _stream())
    _lib.check(rc, "cdl_gabor_filter_banks")
    return ws


def gabor_filter_banks_bwd(params, dws, P, transposes):
    """Adjoint of gabor_filter_banks in one launch: per bank (dalpha, da, dw0, dpsi) for upstream dws[k] (None: zeros)."""
    nb = len(params)
    params = [tuple(_dev(v, n) for v, n in zip(pr, ("alpha", "a", "w0", "psi"))) for pr in params]
    dws = [None if d is None else _dev(d, "dw") for d in dws]
    order, M, C = params[0][3].shape
    nq = order * M * C
    dev = params[0][3].device
    blocks = torch.empty((nb, 6 * nq), device=dev, dtype=torch.float32)
    flags = (ctypes.c_int * nb)(*[int(bool(t)) for t in transposes])
    dwtab = (ctypes.c_void_p * nb)(*[None if d is None else d.data_ptr() for d in dws])
    rc = _lib.lib().cdl_gabor_filter_banks_bwd(nb, *(_ptr_table([pr[i] for pr in params]) for i in range(4)), flags,
                                               dwtab, _ptr_table(list(blocks.unbind(0))), order, M, C, P, _stream())
    _lib.check(rc, "cdl_gabor_filter_banks_bwd")
    outs = []
    for k, pr in enumerate(params):
        b = blocks[k]
        outs.append((b[:nq].view_as(pr[0]), b[nq:3 * nq].view_as(pr[1]), b[3 * nq:5 * nq].view_as(pr[2]),
                     b[5 * nq:].view_as(pr[3])))
    return outs


# ------------------------------------------------------------------------------------------ fused MFMA path
PRECISION = {"split3": 0, "bf16": 1, "split4": 2}


class exact_fp32:
    """`with exact_fp32(): ...` -- the shape-generic entry points called from this thread inside the block stay on the fp32
    VALU kernels (cdl_set_exact_fp32, include/cdlnet_hip.h); `exact_fp32(False)` is a no-op block."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            self.prev = _lib.lib().cdl_set_exact_fp32(1)
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.lib().cdl_set_exact_fp32(self.prev)
        return False
# layouts of the fat tensors that stay inside a fused sweep (include/cdlnet_hip.h, CDL_LAY_*): "nchw" is the
# reference's layout, "blocked" the pixel-blocked fp32 layout (same values, 16-byte accesses, the default),
# "blocked_bf16" opt-in bf16 STORAGE of the codes (half the bytes; outside the 1e-5 parity gate)
LAYOUT = {"nchw": 0, "blocked": 1, "blocked_bf16": 2, "rsc": 3}
TILES_REVERSED = 16


def _lay_in(layout):
    return LAYOUT[layout] << 5


def _lay_out(layout):
    return LAYOUT[layout] << 7


def fused_code_buffer(g: "Geometry", layout, device, count=1):
    """`count` code tensors in `layout`: NCHW tensors for "nchw", flat byte-sized fp32 / bf16 buffers otherwise."""
    if layout == "nchw":
        return torch.empty((count,) + g.code_shape(), device=device, dtype=torch.float32)
    gs = g.c_struct()
    nbytes = int(_lib.lib().cdl_fused2d_code_bytes(ctypes.byref(gs), LAYOUT[layout]))
    assert nbytes > 0
    if layout == "blocked":
        return torch.empty((count, nbytes // 4), device=device, dtype=torch.float32)
    return torch.empty((count, nbytes // 2), device=device, dtype=torch.bfloat16)


def fused_to_nchw(g: "Geometry", t, layout):
    """A blocked code tensor as (N, M, H, W) fp32 (host-side plumbing for tests and tools: a torch permute)."""
    if layout == "nchw":
        return t
    H, W = g.dims[1], g.dims[2]
    XB = (W + 31) // 32
    v = t.reshape(g.N, H, XB, g.M // 4, 32, 4).to(torch.float32)
    return v.permute(0, 3, 5, 1, 2, 4).reshape(g.N, g.M, H, XB * 32)[..., :W].contiguous()


def fused_from_nchw(g: "Geometry", z, layout):
    """Inverse of fused_to_nchw (padding pixels are zero)."""
    if layout == "nchw":
        return z
    H, W = g.dims[1], g.dims[2]
    XB = (W + 31) // 32
    zp = torch.nn.functional.pad(z, (0, XB * 32 - W))
    v = zp.reshape(g.N, g.M // 4, 4, H, XB, 32).permute(0, 3, 4, 1, 5, 2).contiguous().reshape(1, -1)
    return v[0].to(torch.bfloat16 if layout == "blocked_bf16" else torch.float32)


def fused_supported(g: Geometry) -> bool:
    gs = g.c_struct()
    return bool(_lib.lib().cdl_fused2d_supported(ctypes.byref(gs)))


def fused_prep(wA, wB):
    """bf16 hi/lo MFMA fragments for one launch: analysis bank wA with the synthesis bank wB that
    consumes its output (B_{k+1}, or D after the last iteration)."""
    wA, wB = _dev(wA, "wA"), _dev(wB, "wB")
    M, P = wA.shape[0], wA.shape[-1]
    nbytes = _lib.lib().cdl_fused2d_frag_bytes(M)
    frags = torch.empty(nbytes, device=wA.device, dtype=torch.uint8)
    rc = _lib.lib().cdl_fused2d_prep(_ptr(wA), _ptr(wB), _ptr(frags), M, P, _stream())
    _lib.check(rc, "cdl_fused2d_prep")
    return frags


def fused_patches(g: Geometry, device):
    gs = g.c_struct()
    n = _lib.lib().cdl_fused2d_patch_floats(ctypes.byref(gs))
    return torch.empty(n, device=device, dtype=torch.float32)


def fused_map(g: Geometry, device):
    """Empty support/sign bit map of one code tensor: (N, 4, H, W) int32 words (cdl_fused2d_map_words)."""
    gs = g.c_struct()
    n = int(_lib.lib().cdl_fused2d_map_words(ctypes.byref(gs)))
    assert n == g.N * 4 * g.dims[1] * g.dims[2]
    return torch.empty((g.N, 4, g.dims[1], g.dims[2]), device=device, dtype=torch.int32)


def fused_support_map(g: Geometry, z, out=None):
    """The bit map the reverse stage reads instead of the fat z (built from a tensor; the training forward
    gets it for free from cdl_fused2d_iter_fwd)."""
    z = _dev(z, "z")
    assert tuple(z.shape) == g.code_shape()
    if out is None:
        out = fused_map(g, z.device)
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_support_map(ctypes.byref(gs), _ptr(z), _ptr(out), _stream())
    _lib.check(rc, "cdl_fused2d_support_map")
    return out


def fused_iter(g: Geometry, r, zin, tau, frags, sgn, patches, precision="split3", out=None, map_out=None,
               lay_in="nchw", lay_out="nchw"):
    """One fused iteration.  lay_in / lay_out: layouts of zin / the result (a flat buffer unless "nchw")."""
    r, tau = _dev(r, "r"), _dev(tau, "tau")
    if zin is not None and lay_in == "nchw":
        zin = _dev(zin, "zin")
    assert tuple(r.shape) == g.image_shape()
    if out is None:
        out = fused_code_buffer(g, lay_out, r.device)[0]
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_iter_fwd(ctypes.byref(gs), _ptr(r), _ptr(zin), _ptr(tau), _ptr(frags),
                                         float(sgn), _ptr(out), _ptr(patches), _ptr(map_out),
                                         PRECISION[precision] | _lay_in(lay_in) | _lay_out(lay_out), _stream())
    _lib.check(rc, "cdl_fused2d_iter_fwd")
    return out


def fused_assemble(g: Geometry, patches, mask=None, sub=None, alpha=1.0, out=None):
    """out = mask * alpha * (overlap-sum of the patches) - sub."""
    mask, sub = _opt(mask, "mask"), _opt(sub, "sub")
    if out is None:
        out = torch.empty(g.image_shape(), device=patches.device, dtype=torch.float32)
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_assemble(ctypes.byref(gs), _ptr(patches), _ptr(mask), _ptr(sub),
                                         float(alpha), _ptr(out), _stream())
    _lib.check(rc, "cdl_fused2d_assemble")
    return out


def fused_tiles(g: Geometry) -> int:
    gs = g.c_struct()
    return int(_lib.lib().cdl_fused2d_tiles(ctypes.byref(gs)))


def fused_stage_bwd(g: Geometry, thin, base, gate, frags, patches, dtau_partial, do_synth,
                    precision="split3", out=None, lay_in="nchw", lay_out="nchw", r2=None, alpha=1.0, workspace=None):
    """One reverse-sweep stage: du = [z' != 0] * (base + corr(thin; W1)); patches = W2^T du.  `gate` is the
    bit map of z' (int32, from the forward or fused_support_map) or z' itself (the map is built first).
    With `r2` (thin) and `workspace` (fused_wgrad_workspace) the launch also accumulates dA = alpha * du (x) im2col(r2)
    (cdl_fused2d_stage_bwd_da) and (du, dA) is returned."""
    thin = _dev(thin, "thin")
    if base is not None and lay_in == "nchw":
        base = _dev(base, "base")
    if gate.dtype != torch.int32:
        gate = fused_support_map(g, gate)
    assert gate.is_cuda and gate.is_contiguous() and gate.numel() == g.N * 4 * g.dims[1] * g.dims[2]
    if out is None:
        out = fused_code_buffer(g, lay_out, thin.device)[0]
    gs = g.c_struct()
    flags = PRECISION[precision] | _lay_in(lay_in) | _lay_out(lay_out)
    if r2 is not None:
        r2 = _dev(r2, "r2")
        dA = torch.empty(g.filter_shape(), device=thin.device, dtype=torch.float32)
        rc = _lib.lib().cdl_fused2d_stage_bwd_da(ctypes.byref(gs), _ptr(thin), _ptr(base), _ptr(gate), _ptr(frags),
                                                 _ptr(out), _ptr(patches), _ptr(dtau_partial), int(bool(do_synth)),
                                                 _ptr(r2), float(alpha), _ptr(dA), _ptr(workspace), flags, _stream())
        _lib.check(rc, "cdl_fused2d_stage_bwd_da")
        return out, dA
    rc = _lib.lib().cdl_fused2d_stage_bwd(ctypes.byref(gs), _ptr(thin), _ptr(base), _ptr(gate),
                                          _ptr(frags), _ptr(out), _ptr(patches), _ptr(dtau_partial),
                                          int(bool(do_synth)), flags, _stream())
    _lib.check(rc, "cdl_fused2d_stage_bwd")
    return out


def fused_dtau_reduce(g: Geometry, dtau_partial, c, dt_k):
    assert dt_k.is_contiguous() and dt_k.numel() == 2 * g.M
    gs = g.c_struct()
    base = dt_k.data_ptr()
    rc = _lib.lib().cdl_fused2d_dtau_reduce(ctypes.byref(gs), _ptr(dtau_partial), _ptr(_opt(c, "c")),
                                            ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * g.M),
                                            _stream())
    _lib.check(rc, "cdl_fused2d_dtau_reduce")


def fused_wgrad_workspace(g: Geometry, device):
    gs = g.c_struct()
    n = _lib.lib().cdl_fused2d_wgrad_workspace_floats(ctypes.byref(gs))
    return torch.empty(n, device=device, dtype=torch.float32)


def fused_wgrad(g: Geometry, workspace, X0=None, T0=None, alpha0=1.0, X1=None, T1=None, alpha1=1.0,
                precision="split3", layout="nchw"):
    """Up to two filter gradients in one launch: dw_a = alpha_a * sum X_a (x) im2col(T_a); `layout` of X0, X1."""
    outs = []
    args = []
    for X, T, al in ((X0, T0, alpha0), (X1, T1, alpha1)):
        if X is None:
            outs.append(None)
            args += [None, None, 0.0, None]
        else:
            X, T = (_dev(X, "X") if layout == "nchw" else X), _dev(T, "T")
            dw = torch.empty(g.filter_shape(), device=X.device, dtype=torch.float32)
            outs.append(dw)
            args += [_ptr(X), _ptr(T), float(al), _ptr(dw)]
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_wgrad(ctypes.byref(gs), *args, _ptr(workspace),
                                      PRECISION[precision] | _lay_in(layout), _stream())
    _lib.check(rc, "cdl_fused2d_wgrad")
    return outs


def _ptr_table(tensors):
    """Host array of device pointers (kept alive by the caller for the duration of the call)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def fused_forward(g: Geometry, yp, mask_p, tau, A, B, keep, precision="split3", layout="blocked"):
    """Whole forward sweep in one C call.  keep=True: every z_k and r_k gets its own buffer (training) and
    each launch also writes the support/sign bit map of its z_{k+1}; keep=False: two ping-pong buffers
    each, no maps.  `layout`: how z_1..z_{K-1} are stored (they never leave the sweeps; z_K is always NCHW).
    Returns (xp, z_K, codes, resid, maps); codes[:-1] are flat buffers in `layout` unless it is "nchw"."""
    K = len(A)
    yp, tau = _dev(yp, "yp"), _dev(tau, "tau")
    mask_p = _opt(mask_p, "mask")
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    dev = yp.device
    nz = (K - 1) if keep else min(K - 1, 2)              # inner codes z_1..z_{K-1}
    nr = (K - 1) if keep else min(K - 1, 2)
    # one allocation per family (views into it): at cfg1's size the allocator calls cost more than the kernels
    zK = torch.empty(g.code_shape(), device=dev, dtype=torch.float32)
    zbuf = fused_code_buffer(g, layout, dev, max(nz, 1))
    rbuf = torch.empty((max(nr, 1),) + g.image_shape(), device=dev, dtype=torch.float32)
    z = [zbuf[k % nz] for k in range(K - 1)] + [zK]
    r = [rbuf[k % nr] for k in range(K - 1)] if K > 1 else []
    maps = list(torch.empty((K, g.N, 4, g.dims[1], g.dims[2]), device=dev, dtype=torch.int32).unbind(0)) if keep else []
    xp = torch.empty(g.image_shape(), device=dev, dtype=torch.float32)
    frags = torch.empty(K * _lib.lib().cdl_fused2d_frag_bytes(g.M), device=dev, dtype=torch.uint8)
    patches = fused_patches(g, dev)
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_forward(ctypes.byref(gs), K, _ptr(yp), _ptr(mask_p), _ptr(tau), _ptr_table(A),
                                        _ptr_table(B), _ptr_table(z), _ptr_table(r) if r else None,
                                        _ptr_table(maps) if maps else None, _ptr(xp), _ptr(frags), _ptr(patches),
                                        PRECISION[precision] | _lay_in(layout), _stream())
    _lib.check(rc, "cdl_fused2d_forward")
    return xp, z[K - 1], (z if keep else [z[K - 1]]), (r if keep else []), maps


def fused_backward(g: Geometry, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, precision="split3", maps=None,
                   layout="blocked"):
    """Whole reverse sweep in one C call; returns (dA list, dB list); dt (K,2,M) is written in place.
    maps: the forward's bit maps of z_1..z_K (built here from the codes when not given).  `layout`: that of
    codes[:-1] (as fused_forward returned them) and of the du work buffers; codes[-1] = z_K, g_z: NCHW.
    g_xp None (a loss on the code only) is a zero image gradient."""
    K = len(A)
    dev = yp.device
    if not maps:
        maps = [fused_support_map(g, fused_to_nchw(g, t, layout if k < K - 1 else "nchw")) for k, t in enumerate(codes)]
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    codes = [(_dev(t, "z") if (layout == "nchw" or k == K - 1) else t) for k, t in enumerate(codes)]
    resid = [_dev(t, "r") for t in resid]
    if g_xp is None:
        g_xp = torch.zeros(g.image_shape(), device=dev, dtype=torch.float32)
    g_xp, g_z, c, mask_p = _dev(g_xp, "g_xp"), _opt(g_z, "g_z"), _opt(c, "c"), _opt(mask_p, "mask")
    dAB = torch.empty((2 * K,) + g.filter_shape(), device=dev, dtype=torch.float32)     # one allocation, 2K views
    dA, dB = list(dAB[:K].unbind(0)), list(dAB[K:].unbind(0))
    dub = fused_code_buffer(g, layout, dev, 2 if K > 1 else 1)
    du0, du1 = dub[0], dub[1 if K > 1 else 0]
    q = torch.empty(g.image_shape(), device=dev, dtype=torch.float32)
    frags = torch.empty(K * _lib.lib().cdl_fused2d_frag_bytes(g.M), device=dev, dtype=torch.uint8)
    patches = fused_patches(g, dev)
    dtp = torch.empty((fused_tiles(g), g.M), device=dev, dtype=torch.float32)
    ws = fused_wgrad_workspace(g, dev)
    assert dt.is_contiguous() and dt.numel() == K * 2 * g.M
    gs = g.c_struct()
    rc = _lib.lib().cdl_fused2d_backward(
        ctypes.byref(gs), K, _ptr(yp), _ptr(mask_p), _ptr(c), _ptr_table(A), _ptr_table(B), _ptr_table(codes),
        _ptr_table(resid) if resid else None, _ptr_table(list(maps)), _ptr(g_xp), _ptr(g_z), _ptr_table(dA),
        _ptr_table(dB), _ptr(dt),
        _ptr(du0), _ptr(du1), _ptr(q), _ptr(frags), _ptr(patches), _ptr(dtp), _ptr(ws),
        PRECISION[precision] | _lay_in(layout), _stream())
    _lib.check(rc, "cdl_fused2d_backward")
    return dA, dB


# ------------------------------------------------------------------------------------------ fused generic path
# cdl_fusedg.hip: the fused iteration for any C, 2-D / 3-D, unit stride, P in {3,5,7}, M <= 64 (BASELINE configs[2]
# and [3]), and -- behind the same entry points -- cdl_strip.hip: one image channel, stride 1 or 2, up to 192 subbands
# (the shipped CDLNet-s2030 architecture); same call structure as the 2-D flagship path above, codes in the
# reference's (N,M,[D,]H,W) layout.
def fusedg_supported(g: Geometry) -> bool:
    gs = g.c_struct()
    return bool(_lib.lib().cdl_fusedg_supported(ctypes.byref(gs)))


def _fusedg_sizes(g: Geometry):
    gs = g.c_struct()
    L = _lib.lib()
    return (int(L.cdl_fusedg_frag_bytes(ctypes.byref(gs))), int(L.cdl_fusedg_patch_floats(ctypes.byref(gs))),
            int(L.cdl_fusedg_tiles(ctypes.byref(gs))), int(L.cdl_fusedg_map_words(ctypes.byref(gs))))


def _fusedg_map_shape(g: Geometry):
    """(N, planes) + code dims.  The tile kernel (unit stride, M <= 64): 4 planes; the strip kernel (cdl_strip.hip): 4
    planes per PAIR of 32-channel tiles, on the code grid."""
    mw = _fusedg_sizes(g)[3]
    sp = tuple(d // st for d, st in zip(g.dims, g.stride))          # (Dz, Hz, Wz), leading 1 for 2-D
    per = 1
    for d in sp:
        per *= d
    return (g.N, mw // (g.N * per)) + sp


def fusedg_map(g: Geometry, device):
    return torch.empty(_fusedg_map_shape(g), device=device, dtype=torch.int32)


def fusedg_prep(g: Geometry, wA, wB):
    wA, wB = _dev(wA, "wA"), _dev(wB, "wB")
    frags = torch.empty(_fusedg_sizes(g)[0], device=wA.device, dtype=torch.uint8)
    gs = g.c_struct()
    _lib.check(_lib.lib().cdl_fusedg_prep(ctypes.byref(gs), _ptr(wA), _ptr(wB), _ptr(frags), _stream()), "cdl_fusedg_prep")
    return frags


def fusedg_patches(g: Geometry, device):
    return torch.empty(_fusedg_sizes(g)[1], device=device, dtype=torch.float32)


def _rsc_dims(g: Geometry):
    """(N, M, rows, Wz, strips): rows = code depth x code height (the layout blocks every code row of every depth)."""
    cs = g.code_shape()
    rows = 1
    for d in cs[2:-1]:
        rows *= d
    return cs[0], cs[1], rows, cs[-1], (cs[-1] + 31) // 32


def fusedg_rsc_buffer(g: Geometry, device, count=1):
    """Flat fp32 buffer(s) for code tensors in the strip kernels' row-strip channel-major layout ("rsc":
    [n][code depth][code row][ceil(Wz/32)][M][32 columns]; include/cdlnet_hip.h CDL_LAY_RSC)."""
    N, M, rows, wz, nsx = _rsc_dims(g)
    return torch.empty((count, N * M * rows * nsx * 32), device=device, dtype=torch.float32)


def fusedg_to_rsc(g: Geometry, z):
    """(N,M,[Dz,]Hz,Wz) -> flat rsc buffer (host-side plumbing for tests)."""
    N, M, rows, wz, nsx = _rsc_dims(g)
    zp = torch.zeros((N, M, rows, nsx * 32), device=z.device, dtype=torch.float32)
    zp[..., :wz] = z.reshape(N, M, rows, wz)
    return zp.reshape(N, M, rows, nsx, 32).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def fusedg_from_rsc(g: Geometry, buf):
    N, M, rows, wz, nsx = _rsc_dims(g)
    z = buf.reshape(N, rows, nsx, M, 32).permute(0, 3, 1, 2, 4).reshape(N, M, rows, nsx * 32)
    return z[..., :wz].contiguous().reshape(g.code_shape())


def fusedg_iter(g: Geometry, r, zin, tau, frags, sgn, patches, out=None, map_out=None, lay_in="nchw", lay_out="nchw"):
    r, zin, tau = _dev(r, "r"), _opt(zin, "zin"), _dev(tau, "tau")
    assert tuple(r.shape) == g.image_shape()
    if out is None:
        out = (torch.empty(g.code_shape(), device=r.device, dtype=torch.float32) if lay_out == "nchw"
               else fusedg_rsc_buffer(g, r.device)[0])
    gs = g.c_struct()
    rc = _lib.lib().cdl_fusedg_iter_fwd(ctypes.byref(gs), _ptr(r), _ptr(zin), _ptr(tau), _ptr(frags), float(sgn),
                                        _ptr(out), _ptr(patches), _ptr(map_out), _lay_in(lay_in) | _lay_out(lay_out),
                                        _stream())
    _lib.check(rc, "cdl_fusedg_iter_fwd")
    return out


def fusedg_assemble(g: Geometry, patches, mask=None, sub=None, alpha=1.0, out=None):
    mask, sub = _opt(mask, "mask"), _opt(sub, "sub")
    if out is None:
        out = torch.empty(g.image_shape(), device=patches.device, dtype=torch.float32)
    gs = g.c_struct()
    rc = _lib.lib().cdl_fusedg_assemble(ctypes.byref(gs), _ptr(patches), _ptr(mask), _ptr(sub), float(alpha),
                                        _ptr(out), _stream())
    _lib.check(rc, "cdl_fusedg_assemble")
    return out


def fusedg_support_map(g: Geometry, z):
    """Bit map of a code tensor in the layout the fused stages use: (N, 4 * pairs, code dims) words; channel
    32R + 8q + 4h + e is bit 16(R&1) + 4q + e of plane 4(R>>1) + 2h (support, [z != 0]) and of plane 4(R>>1) + 2h + 1
    (sign bit, set only where there is support) -- one plane quadruple per pair of 32-channel tiles (M <= 64: the 4 planes of the tile kernel).
    Host-side plumbing for tests."""
    z = _dev(z, "z")
    N, M = z.shape[:2]
    shape = _fusedg_map_shape(g)
    sp = shape[2:]
    zz = z.reshape((N, M) + sp)
    out = torch.zeros(shape, device=z.device, dtype=torch.int64)
    for ch in range(M):
        R, rem = divmod(ch, 32)
        q, rem = divmod(rem, 8)
        h, e = divmod(rem, 4)
        bit = 16 * (R & 1) + 4 * q + e
        pl = 4 * (R >> 1) + 2 * h
        nz = zz[:, ch] != 0
        out[:, pl] |= nz.to(torch.int64) << bit
        out[:, pl + 1] |= (torch.signbit(zz[:, ch]) & nz).to(torch.int64) << bit
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out)
    return out.to(torch.int32).contiguous()


def fusedg_stage_bwd(g: Geometry, thin, base, gate_map, frags, patches, dtau_partial, do_synth, out=None,
                     lay_in="nchw", lay_out="nchw"):
    thin, base = _dev(thin, "thin"), _opt(base, "base")
    assert gate_map.dtype == torch.int32 and gate_map.is_contiguous()
    if out is None:
        out = (torch.empty(g.code_shape(), device=thin.device, dtype=torch.float32) if lay_out == "nchw"
               else fusedg_rsc_buffer(g, thin.device)[0])
    gs = g.c_struct()
    rc = _lib.lib().cdl_fusedg_stage_bwd(ctypes.byref(gs), _ptr(thin), _ptr(base), _ptr(gate_map), _ptr(frags),
                                         _ptr(out), _ptr(patches), _ptr(dtau_partial), int(bool(do_synth)),
                                         _lay_in(lay_in) | _lay_out(lay_out), _stream())
    _lib.check(rc, "cdl_fusedg_stage_bwd")
    return out


def fusedg_dtau_reduce(g: Geometry, dtau_partial, c, dt_k):
    assert dt_k.is_contiguous() and dt_k.numel() == 2 * g.M
    gs = g.c_struct()
    base = dt_k.data_ptr()
    rc = _lib.lib().cdl_fusedg_dtau_reduce(ctypes.byref(gs), _ptr(dtau_partial), _ptr(_opt(c, "c")),
                                           ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * g.M), _stream())
    _lib.check(rc, "cdl_fusedg_dtau_reduce")


def fusedg_code_layout(g: Geometry, training=True) -> str:
    """Layout the fused generic sweeps keep their internal codes in: "rsc" for the strip kernels' shapes (with
    `training`: when the matrix-core filter-gradient kernel takes the geometry, since its VALU fallbacks read the
    reference layout only), else "nchw" (cdl_fusedg_code_layout)."""
    gs = g.c_struct()
    return "rsc" if int(_lib.lib().cdl_fusedg_code_layout(ctypes.byref(gs), int(bool(training)))) == LAYOUT["rsc"] else "nchw"


def _fusedg_code_buffers(g: Geometry, layout, device, count):
    """`count` code tensors of a sweep: (N,M,..) tensors for "nchw", flat buffers for "rsc"."""
    if layout == "nchw":
        return list(torch.empty((max(count, 1),) + g.code_shape(), device=device, dtype=torch.float32).unbind(0))
    return list(fusedg_rsc_buffer(g, device, max(count, 1)).unbind(0))


def fusedg_forward(g: Geometry, yp, mask_p, tau, A, B, keep, layout="nchw"):
    """Whole forward sweep in one C call (cdl_fusedg_forward); same contract as fused_forward: codes[:-1] come back
    in `layout` ("nchw", or "rsc" where fusedg_code_layout allows it), codes[-1] = z_K as (N,M,..)."""
    K = len(A)
    yp, tau, mask_p = _dev(yp, "yp"), _dev(tau, "tau"), _opt(mask_p, "mask")
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    dev = yp.device
    fb, pf, tiles, mw = _fusedg_sizes(g)
    nz = (K - 1) if keep else min(K - 1, 2)                 # internal codes z_1..z_{K-1}; z_K has its own tensor
    nr = (K - 1) if keep else min(K - 1, 2)
    zint = _fusedg_code_buffers(g, layout, dev, nz)
    zK = torch.empty(g.code_shape(), device=dev, dtype=torch.float32)
    rbuf = torch.empty((max(nr, 1),) + g.image_shape(), device=dev, dtype=torch.float32)
    z = [zint[k % nz] for k in range(K - 1)] + [zK]
    r = [rbuf[k % nr] for k in range(K - 1)] if K > 1 else []
    maps = list(torch.empty((K,) + _fusedg_map_shape(g), device=dev, dtype=torch.int32).unbind(0)) if keep else []
    xp = torch.empty(g.image_shape(), device=dev, dtype=torch.float32)
    frags = torch.empty(K * fb, device=dev, dtype=torch.uint8)
    patches = torch.empty(pf, device=dev, dtype=torch.float32)
    gs = g.c_struct()
    rc = _lib.lib().cdl_fusedg_forward(ctypes.byref(gs), K, _ptr(yp), _ptr(mask_p), _ptr(tau), _ptr_table(A),
                                       _ptr_table(B), _ptr_table(z), _ptr_table(r) if r else None,
                                       _ptr_table(maps) if maps else None, _ptr(xp), _ptr(frags), _ptr(patches),
                                       _lay_in(layout), _stream())
    _lib.check(rc, "cdl_fusedg_forward")
    return xp, zK, (z if keep else [zK]), (r if keep else []), maps


def fusedg_backward(g: Geometry, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None, layout="nchw"):
    """Whole reverse sweep in one C call (cdl_fusedg_backward); returns (dA list, dB list), fills dt (K,2,M).
    `layout`: that of codes[:-1] (and of the du buffers allocated here)."""
    K = len(A)
    dev = yp.device
    if not maps:
        nchw = [t if (layout == "nchw" or i == K - 1) else fusedg_from_rsc(g, t) for i, t in enumerate(codes)]
        maps = [fusedg_support_map(g, t) for t in nchw]
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    codes = [_dev(t, "z") for t in codes]
    resid = [_dev(t, "r") for t in resid]
    if g_xp is None:
        g_xp = torch.zeros(g.image_shape(), device=dev, dtype=torch.float32)
    g_xp, g_z, c, mask_p = _dev(g_xp, "g_xp"), _opt(g_z, "g_z"), _opt(c, "c"), _opt(mask_p, "mask")
    fb, pf, tiles, mw = _fusedg_sizes(g)
    dAB = torch.empty((2 * K,) + g.filter_shape(), device=dev, dtype=torch.float32)
    dA, dB = list(dAB[:K].unbind(0)), list(dAB[K:].unbind(0))
    du = _fusedg_code_buffers(g, layout, dev, 2 if K > 1 else 1)
    q = torch.empty(g.image_shape(), device=dev, dtype=torch.float32)
    frags = torch.empty(K * fb, device=dev, dtype=torch.uint8)
    patches = torch.empty(pf, device=dev, dtype=torch.float32)
    dtp = torch.empty((tiles, g.M), device=dev, dtype=torch.float32)
    gs = g.c_struct()
    nws = int(_lib.lib().cdl_wgrad_workspace_floats(ctypes.byref(gs)))
    ws = torch.empty(max(nws, 1), device=dev, dtype=torch.float32)
    assert dt.is_contiguous() and dt.numel() == K * 2 * g.M
    rc = _lib.lib().cdl_fusedg_backward(
        ctypes.byref(gs), K, _ptr(yp), _ptr(mask_p), _ptr(c), _ptr_table(A), _ptr_table(B), _ptr_table(codes),
        _ptr_table(resid) if resid else None, _ptr_table(list(maps)), _ptr(g_xp), _ptr(g_z), _ptr_table(dA),
        _ptr_table(dB), _ptr(dt), _ptr(du[0]), _ptr(du[1 if K > 1 else 0]), _ptr(q), _ptr(frags), _ptr(patches),
        _ptr(dtp), _ptr(ws), nws, _lay_in(layout), _stream())
    _lib.check(rc, "cdl_fusedg_backward")
    return dA, dB


# ------------------------------------------------------------------------------------------
# whole sweeps of the shape-generic loop (cdl_sweep.hip)
def _new(shape, device):
    return torch.empty(shape, device=device, dtype=torch.float32)


def ista_scratch(g: Geometry, device):
    gs = g.c_struct()
    n = int(_lib.lib().cdl_ista_scratch_floats(ctypes.byref(gs)))
    return _scratch(device, max(n, 1)), n


def ista_forward(g: Geometry, yp, mask_p, tau, A, B, keep, z_prev=None, z_after=None, gam1=None, gam2=None):
    """Generic forward sweep in one C call.  Plain ST loop (z_prev None) or the CSR maps.  keep=True:
    every z_k, r_k (and u_k for CSR) gets its own buffer.  Returns (xp, z_K, codes, resid, us)."""
    K = len(A)
    yp, tau, mask_p = _dev(yp, "yp"), _dev(tau, "tau"), _opt(mask_p, "mask")
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    z_prev, z_after = _opt(z_prev, "z_prev"), _opt(z_after, "z_after")
    gam1, gam2 = _opt(gam1, "gam1"), _opt(gam2, "gam2")
    dev = yp.device
    nz = K if keep else min(K, 2)
    nr = (K - 1) if keep else min(K - 1, 2)
    zbuf = _new((nz,) + g.code_shape(), dev)
    rbuf = _new((max(nr, 1),) + g.image_shape(), dev)
    z = [zbuf[k % nz] for k in range(K)]
    r = [rbuf[k % nr] for k in range(K - 1)] if K > 1 else []
    u = list(_new((K,) + g.code_shape(), dev).unbind(0)) if (keep and z_prev is not None) else []
    xp = _new(g.image_shape(), dev)
    ws, n = ista_scratch(g, dev)
    gs = g.c_struct()
    rc = _lib.lib().cdl_ista_forward(ctypes.byref(gs), K, _ptr(yp), _ptr(mask_p), _ptr(tau), _ptr(z_prev),
                                     _ptr(z_after), _ptr(gam1), _ptr(gam2), _ptr_table(A), _ptr_table(B),
                                     _ptr_table(z), _ptr_table(r) if r else None, _ptr_table(u) if u else None,
                                     _ptr(xp), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_ista_forward")
    return xp, z[K - 1], (z if keep else [z[K - 1]]), (r if keep else []), u


def ista_backward(g: Geometry, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, us=None, z_prev=None,
                  z_after=None, lam=None, gam1=None, gam2=None, dg1=None, dg2=None, gz_prev=None, gz_after=None):
    """Generic reverse sweep in one C call; returns (dA, dB) and fills dt [, dg1, dg2, gz_prev, gz_after]."""
    K = len(A)
    dev = yp.device
    A = [_dev(w, "A") for w in A]
    B = [_dev(w, "B") for w in B]
    dAB = _new((2 * K,) + g.filter_shape(), dev)
    dA, dB = list(dAB[:K].unbind(0)), list(dAB[K:].unbind(0))
    g0, g1, q = _new(g.code_shape(), dev), _new(g.code_shape(), dev), _new(g.image_shape(), dev)
    ws, n = ista_scratch(g, dev)
    gs = g.c_struct()
    rc = _lib.lib().cdl_ista_backward(
        ctypes.byref(gs), K, _ptr(_dev(yp, "yp")), _ptr(_opt(mask_p, "mask")), _ptr(_opt(c, "c")),
        _ptr(_opt(z_prev, "z_prev")), _ptr(_opt(z_after, "z_after")), _ptr(_opt(lam, "lam")),
        _ptr(_opt(gam1, "gam1")), _ptr(_opt(gam2, "gam2")), _ptr_table(A), _ptr_table(B),
        _ptr_table([_dev(t, "z") for t in codes]), _ptr_table([_dev(t, "r") for t in resid]) if resid else None,
        _ptr_table([_dev(t, "u") for t in us]) if us else None, _ptr(_opt(g_xp, "g_xp")), _ptr(_opt(g_z, "g_z")),
        _ptr_table(dA), _ptr_table(dB), _ptr(dt), _ptr(dg1), _ptr(dg2), _ptr(gz_prev), _ptr(gz_after),
        _ptr(g0), _ptr(g1), _ptr(q), _ptr(ws), n, _stream())
    _lib.check(rc, "cdl_ista_backward")
    return dA, dB


def fused_timing(enable: bool):
    """Start / stop the HIP-event timing of the fused sweeps' fat kernels (cdl_fused2d_timing)."""
    _lib.check(_lib.lib().cdl_fused2d_timing(1 if enable else 0), "cdl_fused2d_timing")


def fused_timing_read():
    """{class: (average ms, launches)} for the forward stage, reverse stage and filter-gradient launches
    recorded since fused_timing(True)."""
    ms = (ctypes.c_double * 4)()
    cnt = (ctypes.c_int * 4)()
    _lib.check(_lib.lib().cdl_fused2d_timing_read(ms, cnt), "cdl_fused2d_timing_read")
    names = ("stage_fwd", "stage_bwd", "wgrad", "stage_first")
    return {n: ((ms[i] / cnt[i]) if cnt[i] else 0.0, int(cnt[i])) for i, n in enumerate(names)}


# every public wrapper runs with its tensors' device current (see _on_tensor_device)
for _name, _obj in list(globals().items()):
    if callable(_obj) and not _name.startswith("_") and getattr(_obj, "__module__", None) == __name__ \
            and not isinstance(_obj, type):
        globals()[_name] = _on_tensor_device(_obj)
del _name, _obj
