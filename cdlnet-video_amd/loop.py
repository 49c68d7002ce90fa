"""The unrolled-ISTA loop as one autograd node driving the HIP kernels.

Forward (reference model/net.py:76-92, 192-212, 659-675):
    yp, mean, pads, mask_p = pre_process(y)
    z_1     = ST(A_0 yp, tau_0)
    z_{k+1} = ST(z_k - A_k(mask * B_k z_k - yp), tau_k)          k = 1..K-1
    xhat    = post_process(B_0 z_K)
Backward is the hand-derived reverse sweep (what `loss.backward()` does through ATen in the
reference, train.py:98); with du_k = [z_{k+1} != 0] * g_{k+1}:
    dtau_k  = -sum sign(z_{k+1}) du_k
    q_k     = mask * (-A_k^T du_k)
    dA_k    = -du_k (x) r_k,   r_k = mask * B_k z_k - yp  (kept from the forward sweep; it is thin)
    dB_k    =  z_k  (x) q_k
    g_k     = du_k + B_k^T q_k
Only parameter gradients are produced (the reference never differentiates w.r.t. the image).
"""
import torch

from . import ops

# Kernel selection for the forward sweep: "auto" uses the fused MFMA kernels whenever the geometry
# has one (cdl_fused2d_supported) and the shape-generic kernels otherwise; "generic" forces the latter.
# PRECISION applies to the fused 2-D kernels only: "split3" (split-bf16 with three products per multiply: fp32-grade, the
# default), "split4" (all four products: exact fp32 products -- what an objective that differences two forward passes
# needs, see precision_scope / train.mcsure_loss) or "bf16".  "fp32" is a different tier altogether: no fused kernels and no
# matrix cores -- the three-launch iteration on the fp32 VALU kernels (plain fp32 FMAs, several times slower), for the cases
# where 16-17 significant bits per MFMA operand are not enough (cancelling filter gradients: DESIGN.md section 6).
BACKEND = "auto"
ARITHMETIC = tuple(ops.PRECISION) + ("fp32",)
PRECISION = "split3"
# How the fused sweeps store the codes that never leave them (z_1..z_{K-1}, du_k): "blocked" (pixel-blocked
# fp32: same values as "nchw", wider memory accesses; the default), "nchw", or "blocked_bf16" (opt-in bf16
# STORAGE: half the bytes of the dominant tensors, outside the 1e-5 parity gate -- PSNR parity only).
CODE_LAYOUT = "blocked"


def set_backend(name):
    global BACKEND
    if name not in ("auto", "generic"):
        raise ValueError(name)
    BACKEND = name


def set_precision(name):
    global PRECISION
    if name not in ARITHMETIC:
        raise ValueError(name)
    PRECISION = name


class precision_scope:
    """`with precision_scope("split4"): ...` -- the fused 2-D sweeps started inside run in that arithmetic; their reverse
    sweeps use the arithmetic of their forward whenever backward() is called.  `precision_scope("fp32")`: every network
    call started inside (any net of this package) runs, forward and backward, on the fp32 VALU kernels."""

    def __init__(self, name):
        if name not in ARITHMETIC:
            raise ValueError(name)
        self.name = name

    def __enter__(self):
        global PRECISION
        self.saved, PRECISION = PRECISION, self.name
        return self

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self.saved
        return False


def set_code_layout(name):
    global CODE_LAYOUT
    if name not in ops.LAYOUT:
        raise ValueError(name)
    CODE_LAYOUT = name


# Callables to run once per backward pass, queued on the autograd engine from the end of a reverse sweep (they run
# after the engine has accumulated every gradient of that pass): the data-parallel gradient exchange
# (parallel.GradientBucket.attach).
_BACKWARD_END = []
_queued = [False]


def on_backward_end(fn):
    """Register `fn()` to run at the end of every backward pass that contains a reverse sweep of this module;
    returns a callable that removes it."""
    _BACKWARD_END.append(fn)
    return lambda: _BACKWARD_END.remove(fn) if fn in _BACKWARD_END else None


def _queue_backward_end():
    if not _BACKWARD_END or _queued[0]:
        return                                    # e.g. MC-SURE: two sweeps in one pass, one exchange
    _queued[0] = True

    def run():
        _queued[0] = False
        for fn in list(_BACKWARD_END):
            fn()

    torch.autograd.Variable._execution_engine.queue_callback(run)


def _forward_generic(g, yp, mask_p, tau, A, B, keep_codes, keep_resid):
    """Whole sweep from one C call (cdl_ista_forward): same launches as the stepwise form below."""
    keep = keep_codes or keep_resid
    xp, z, codes, resid, _ = ops.ista_forward(g, yp, mask_p, tau, A, B, keep)
    return xp, z, codes, (resid if keep_resid else []), []


def _forward_generic_stepwise(g, yp, mask_p, tau, A, B, keep_codes, keep_resid):
    """Same sweep driven launch by launch from Python (kept for tests and experiments)."""
    K = len(A)
    codes, resid = [], []
    z = ops.analysis(g, yp, A[0], 1.0, None, None, tau[0])
    codes.append(z)
    for k in range(1, K):
        r = ops.synthesis(g, z, B[k], 1.0, None, mask_p, yp)
        z = ops.analysis(g, r, A[k], -1.0, z, None, tau[k])
        if keep_codes:
            codes.append(z)
        if keep_resid:
            resid.append(r)
    xp = ops.synthesis(g, z, B[0], 1.0)
    return xp, z, codes, resid, []


def _forward_fused(g, yp, mask_p, tau, A, B, keep_codes, keep_resid, layout=None):
    """The whole sweep from one C call (cdl_fused2d_forward): per iteration one fused launch + a thin
    assemble, enqueued back to back with no per-launch host work.  codes[:-1] come back in `layout`
    (default CODE_LAYOUT), codes[-1] = z_K as (N,M,H,W)."""
    keep = keep_codes or keep_resid
    xp, z, codes, resid, maps = ops.fused_forward(g, yp, mask_p, tau, A, B, keep, PRECISION,
                                                  layout or CODE_LAYOUT)
    return xp, z, codes, (resid if keep_resid else []), (maps if keep_resid else [])


def _forward_fused_stepwise(g, yp, mask_p, tau, A, B, keep_codes, keep_resid, layout="nchw"):
    """Same sweep driven launch by launch from Python (kept for tests and experiments)."""
    K = len(A)
    frags = [ops.fused_prep(A[k], B[(k + 1) % K]) for k in range(K)]   # last one pairs A_{K-1} with D = B_0
    patches = ops.fused_patches(g, yp.device)
    codes, resid, maps = [], [], []
    r, z = yp, None
    for k in range(K):
        bits = ops.fused_map(g, yp.device) if keep_resid else None
        z = ops.fused_iter(g, r, z, tau[k], frags[k], 1.0 if k == 0 else -1.0, patches, PRECISION, map_out=bits,
                           lay_in=layout, lay_out="nchw" if k == K - 1 else layout)
        if keep_codes or k == 0:
            codes.append(z)
        if keep_resid:
            maps.append(bits)
        if k < K - 1:
            r = ops.fused_assemble(g, patches, mask_p, yp, 1.0)
            if keep_resid:
                resid.append(r)
    xp = ops.fused_assemble(g, patches, None, None, 1.0)
    return xp, z, codes, resid, maps


def _forward_fusedg(g, yp, mask_p, tau, A, B, keep_codes, keep_resid, layout="nchw"):
    """cdl_fusedg_forward: the fused iteration for C > 1 / 3-D / P in {3,5,7} / M <= 64, and (the strip kernel) for one
    image channel, stride 1 / 2, M <= 192: one fused launch + a thin assemble per iteration.  codes[:-1] come back in
    `layout` ("nchw", or "rsc" where ops.fusedg_code_layout(g) says so), codes[-1] = z_K as (N,M,..)."""
    keep = keep_codes or keep_resid
    xp, z, codes, resid, maps = ops.fusedg_forward(g, yp, mask_p, tau, A, B, keep, layout)
    return xp, z, codes, (resid if keep_resid else []), (maps if keep_resid else [])


def _backward_fusedg(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None, layout="nchw"):
    """cdl_fusedg_backward: per iteration one fused reverse stage (du_k, threshold partials, patches of q_k), a thin
    assemble and the two filter gradients.  `layout`: that of codes[:-1]."""
    if g_xp is None and g_z is None:
        return [torch.zeros_like(w) for w in A], [torch.zeros_like(w) for w in B]
    return ops.fusedg_backward(g, yp, mask_p, c, list(A), list(B), list(codes), list(resid), g_xp, g_z, dt,
                               maps=list(maps) if maps else None, layout=layout)


def _backward_generic(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None):
    """Reverse sweep from one C call (cdl_ista_backward)."""
    if g_xp is None and g_z is None:
        return [torch.zeros_like(w) for w in A], [torch.zeros_like(w) for w in B]
    return ops.ista_backward(g, yp, mask_p, c, list(A), list(B), list(codes), list(resid), g_xp, g_z, dt)


def _backward_generic_stepwise(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None):
    """Same reverse sweep driven launch by launch from Python (kept for tests and experiments)."""
    dA, dB = [None] * K, [None] * K
    zK = codes[K - 1]
    if g_xp is not None:
        dB[0] = ops.wgrad(g, zK, g_xp, 1.0)
        gk = ops.analysis(g, g_xp, B[0], 1.0, g_z, None, None)      # B_0^T g_xp (+ g_z)
    else:
        dB[0] = torch.zeros_like(B[0])
        gk = g_z.contiguous() if g_z is not None else torch.zeros_like(zK)
    for k in range(K - 1, 0, -1):
        z_next, z_k, r_k = codes[k], codes[k - 1], resid[k - 1]
        ops.tau_grad(g, gk, z_next, c, dt[k])
        q = ops.synthesis(g, gk, A[k], -1.0, z_next, mask_p, None)
        dA[k] = ops.wgrad(g, gk, r_k, -1.0, gate=z_next)
        dB[k] = ops.wgrad(g, z_k, q, 1.0)
        gk = ops.analysis(g, q, B[k], 1.0, gk, z_next, None)
    ops.tau_grad(g, gk, codes[0], c, dt[0])
    dA[0] = ops.wgrad(g, gk, yp, 1.0, gate=codes[0])
    return dA, dB


def _backward_fused(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None, layout=None, precision=None):
    """Reverse sweep from one C call (cdl_fused2d_backward): per iteration one stage launch (1 fat read +
    the 2-bit map of z_{k+1}, 1 fat write), a thin assemble, and one MFMA filter-gradient launch (2 fat
    reads).  maps: the forward's bit maps (rebuilt from the codes when absent); `layout`: that of codes[:-1]."""
    if g_xp is None and g_z is None:
        return [torch.zeros_like(w) for w in A], [torch.zeros_like(w) for w in B]
    return ops.fused_backward(g, yp, mask_p, c, list(A), list(B), list(codes), list(resid), g_xp, g_z, dt,
                              precision or PRECISION, maps=list(maps) if maps else None, layout=layout or CODE_LAYOUT)


def _backward_fused_stepwise(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None, layout="nchw"):
    """Same reverse sweep driven launch by launch from Python (kept for tests and experiments)."""
    assert maps or layout == "nchw"
    prec = PRECISION
    dev = yp.device
    dA, dB = [None] * K, [None] * K
    patches = ops.fused_patches(g, dev)
    ws = ops.fused_wgrad_workspace(g, dev)
    dtp = torch.empty((ops.fused_tiles(g), g.M), device=dev, dtype=torch.float32)
    thin, du_next = g_xp, g_z
    (dB[0],) = ops.fused_wgrad(g, ws, codes[K - 1], g_xp, 1.0, precision=prec)[:1]
    for k in range(K - 1, -1, -1):
        frags = ops.fused_prep(B[(k + 1) % K], A[k])          # analysis-like bank, synthesis-like bank
        du = ops.fused_stage_bwd(g, thin, du_next, maps[k] if maps else codes[k], frags, patches, dtp, k >= 1, prec,
                                 lay_in="nchw" if k == K - 1 else layout, lay_out=layout)
        ops.fused_dtau_reduce(g, dtp, c, dt[k])
        if k >= 1:
            q = ops.fused_assemble(g, patches, mask_p, None, -1.0)
            dA[k], dB[k] = ops.fused_wgrad(g, ws, du, resid[k - 1], -1.0, codes[k - 1], q, 1.0, prec, layout=layout)
            thin = q
        else:
            (dA[0],) = ops.fused_wgrad(g, ws, du, yp, 1.0, precision=prec, layout=layout)[:1]
        du_next = du
    return dA, dB


def _arithmetic_aware(cls):
    """Class decorator of the autograd Functions below: a call made under PRECISION == "fp32" runs its forward AND (whenever it
    happens) its backward with the generic entry points pinned to the fp32 VALU kernels (ops.exact_fp32)."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx.exact = PRECISION == "fp32"
        with ops.exact_fp32(ctx.exact):
            return fwd(ctx, *args)

    def backward(ctx, *grads):
        with ops.exact_fp32(ctx.exact):
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _no_data_gradients(ctx):
    """The reverse sweeps produce parameter (and neighbour-code) gradients only.  The reference's loop is
    differentiable in the observation, the mask and sigma as well (no caller in the reference uses that); asking for
    those here must fail loudly rather than return a silent None."""
    for idx, name in ((0, "y"), (1, "mask"), (2, "sigma")):
        if ctx.needs_input_grad[idx]:
            raise NotImplementedError(f"cdlnet_video_amd: the gradient with respect to `{name}` is not implemented "
                                      f"(the HIP reverse sweep returns parameter gradients only); detach() it")


@_arithmetic_aware
class UnrolledISTA(torch.autograd.Function):
    """(y, mask, c, t, A_0..A_{K-1}, B_0..B_{K-1}) -> (xhat, z_K[, z_1..z_{K-1}])."""

    @staticmethod
    def forward(ctx, y, mask, c, t, cfg, *weights):
        K = cfg["K"]
        A, B = weights[:K], weights[K:]
        s = cfg["s"]
        yp, mean, pads, mask_p = ops.preprocess(y, s, mask)
        N, C = yp.shape[:2]
        M = A[0].shape[0]
        nd = yp.dim() - 2
        P = tuple(A[0].shape[2:])
        g = ops.Geometry.make(N, C, M, yp.shape[2:], P, tuple(p // 2 for p in P), [s] * nd)
        tau = ops.thresholds(t, c, N)

        ctx.set_materialize_grads(False)          # an unused z output must not cost a fat zero tensor
        _no_data_gradients(ctx)
        keep = any(ctx.needs_input_grad)          # all False under torch.no_grad()
        want_codes = cfg.get("all_codes", False)
        auto = BACKEND == "auto" and not ctx.exact     # "fp32": the fused kernels are matrix-core kernels
        ctx.fused = auto and ops.fused_supported(g)
        ctx.fusedg = auto and not ctx.fused and ops.fusedg_supported(g)
        if ctx.fusedg:
            # codes handed to the caller (forward_generator) must be (N,M,..); otherwise they stay in the sweeps' layout
            ctx.layout = "nchw" if want_codes else ops.fusedg_code_layout(g, training=keep)
            xp, z, codes, resid, maps = _forward_fusedg(g, yp, mask_p, tau, A, B, keep or want_codes, keep, ctx.layout)
        elif ctx.fused:
            # codes handed to the caller (forward_generator) must be (N,M,H,W); otherwise they stay internal
            ctx.layout = "nchw" if want_codes else CODE_LAYOUT
            ctx.precision = PRECISION                  # the reverse sweep runs in the forward's arithmetic
            xp, z, codes, resid, maps = _forward_fused(g, yp, mask_p, tau, A, B, keep or want_codes, keep, ctx.layout)
        else:
            xp, z, codes, resid, maps = _forward_generic(g, yp, mask_p, tau, A, B, keep or want_codes, keep)
        xhat = ops.postprocess(xp, mean, pads)

        ctx.geom, ctx.pads, ctx.K = g, pads, K
        ctx.has_mask, ctx.has_c = mask_p is not None, c is not None
        if keep:
            ctx.save_for_backward(yp, mask_p if mask_p is not None else yp.new_empty(0),
                                  c if c is not None else yp.new_empty(0), t, *weights,
                                  *codes, *resid, *maps)
            ctx.n_maps = len(maps)
        outs = (xhat, z)
        if want_codes:
            extra = tuple(codes[:-1])
            ctx.mark_non_differentiable(*extra)
            outs = outs + extra
        return outs

    @staticmethod
    def backward(ctx, g_xhat, g_z, *_unused):
        K, g = ctx.K, ctx.geom
        saved = ctx.saved_tensors
        yp, mask_p, c, t = saved[:4]
        mask_p = mask_p if ctx.has_mask else None
        c = c if ctx.has_c else None
        A = saved[4:4 + K]
        B = saved[4 + K:4 + 2 * K]
        codes = saved[4 + 2 * K:4 + 3 * K]            # z_1..z_K
        resid = saved[4 + 3 * K:4 + 3 * K + (K - 1)]  # r_1..r_{K-1}
        maps = saved[4 + 3 * K + (K - 1):]            # fused path: support/sign bit maps of z_1..z_K
        assert len(maps) == ctx.n_maps
        dt = torch.zeros((K, 2, g.M), device=yp.device, dtype=torch.float32)
        g_xp = ops.postprocess_bwd(g_xhat.contiguous(), ctx.pads) if g_xhat is not None else None
        if g_z is not None:
            g_z = g_z.contiguous()
        if ctx.fusedg:
            dA, dB = _backward_fusedg(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=maps, layout=ctx.layout)
        elif ctx.fused:                                # a loss on z only is a zero image gradient to the sweep
            dA, dB = _backward_fused(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=maps,
                                     layout=ctx.layout, precision=ctx.precision)
        else:
            dA, dB = _backward_generic(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt)

        _queue_backward_end()
        return (None, None, None, dt.reshape(t.shape), None, *dA, *dB)


# ------------------------------------------------------------------------------------------
# CSR temporal variants (reference model/net.py:426-463, 525-568):
# the same loop with the shrinkage replaced by prox_CSR / prox_CSR_f2 around a neighbour frame's code.
#     u_0 = A_0 yp                       z_1     = prox(u_0; zp[, za], lam_0, gam_0)
#     u_k = z_k - A_k(mask B_k z_k - yp) z_{k+1} = prox(u_k; ...)
# The reverse sweep needs u_k (the derivative masks of the nested shrinkages are not recoverable from
# z_{k+1}), so the training forward keeps u_k next to z_k (both written by the analysis kernel, whose
# epilogue is the proximal map: cdl_analysis_prox); everything else is _backward_generic with the gating
# done by cdl_prox_csr_bwd, which also accumulates the neighbour-code gradients.
def _forward_csr(g, yp, mask_p, lam, gam1, gam2, zp, za, A, B, keep):
    xp, z, codes, resid, us = ops.ista_forward(g, yp, mask_p, lam, A, B, keep, zp, za, gam1, gam2)
    return xp, z, us, codes, resid


def _forward_csr_stepwise(g, yp, mask_p, lam, gam1, gam2, zp, za, A, B, keep):
    K = len(A)
    us, codes, resid = [], [], []
    g2 = (lambda k: gam2[k]) if za is not None else (lambda k: None)
    new = lambda: torch.empty(g.code_shape(), device=yp.device, dtype=torch.float32)
    u = new() if keep else None
    z = ops.analysis_prox(g, yp, A[0], 1.0, None, zp, lam[0], gam1[0], za, g2(0), u_out=u)
    spare = None if keep else new()                      # inference: z ping-pongs between two buffers
    for k in range(1, K):
        if keep:
            us.append(u)
            codes.append(z)
            u = new()
        r = ops.synthesis(g, z, B[k], 1.0, None, mask_p, yp)
        z_next = ops.analysis_prox(g, r, A[k], -1.0, z, zp, lam[k], gam1[k], za, g2(k), u_out=u, out=spare)
        spare, z = (None, z_next) if keep else (z, z_next)
        if keep:
            resid.append(r)
    if keep:
        us.append(u)
        codes.append(z)
    xp = ops.synthesis(g, z, B[0], 1.0)
    return xp, z, us, codes, resid


@_arithmetic_aware
class TemporalISTA(torch.autograd.Function):
    """(y, mask, c, z_prev, z_after|None, t, g1, g2|None, A.., B..) -> (xhat, z_K); gradients for the
    neighbour codes, the three threshold families and both filter banks."""

    @staticmethod
    def forward(ctx, y, mask, c, zp, za, t, g1, g2, cfg, *weights):
        K, s = cfg["K"], cfg["s"]
        A, B = weights[:K], weights[K:]
        yp, mean, pads, mask_p = ops.preprocess(y, s, mask)
        N, C = yp.shape[:2]
        P = tuple(A[0].shape[2:])
        nd = yp.dim() - 2
        g = ops.Geometry.make(N, C, A[0].shape[0], yp.shape[2:], P, tuple(p // 2 for p in P), [s] * nd)
        if tuple(zp.shape) != g.code_shape() or (za is not None and tuple(za.shape) != g.code_shape()):
            raise ValueError(f"neighbour code shape {tuple(zp.shape)} does not match this frame's {g.code_shape()}")
        zp = zp.contiguous()
        za = za.contiguous() if za is not None else None
        lam, gam1 = ops.thresholds(t, c, N), ops.thresholds(g1, c, N)
        gam2 = ops.thresholds(g2, c, N) if za is not None else None
        ctx.set_materialize_grads(False)
        _no_data_gradients(ctx)
        keep = any(ctx.needs_input_grad)
        xp, z, us, codes, resid = _forward_csr(g, yp, mask_p, lam, gam1, gam2, zp, za, A, B, keep)
        xhat = ops.postprocess(xp, mean, pads)
        ctx.geom, ctx.pads, ctx.K = g, pads, K
        ctx.has_mask, ctx.has_c, ctx.has_after = mask_p is not None, c is not None, za is not None
        if keep:
            empty = yp.new_empty(0)
            ctx.save_for_backward(yp, mask_p if mask_p is not None else empty, c if c is not None else empty,
                                  zp, za if za is not None else empty, t, g1, g2 if za is not None else empty,
                                  lam, gam1, gam2 if za is not None else empty, *weights, *us, *codes, *resid)
        return xhat, z

    @staticmethod
    def backward(ctx, g_xhat, g_z):
        K, g = ctx.K, ctx.geom
        sv = ctx.saved_tensors
        yp, mask_p, c, zp, za, t, g1, g2, lam, gam1, gam2 = sv[:11]
        mask_p = mask_p if ctx.has_mask else None
        c = c if ctx.has_c else None
        za, gam2 = (za, gam2) if ctx.has_after else (None, None)
        A, B = sv[11:11 + K], sv[11 + K:11 + 2 * K]
        us = sv[11 + 2 * K:11 + 3 * K]
        codes = sv[11 + 3 * K:11 + 4 * K]
        resid = sv[11 + 4 * K:]
        dev = yp.device
        dt = torch.zeros((K, 2, g.M), device=dev, dtype=torch.float32)
        dg1 = torch.zeros_like(dt)
        dg2 = torch.zeros_like(dt) if za is not None else None
        need_zp, need_za = ctx.needs_input_grad[3], za is not None and ctx.needs_input_grad[4]
        gzp = torch.zeros_like(zp) if need_zp else None
        gza = torch.zeros_like(zp) if need_za else None
        g_xp = ops.postprocess_bwd(g_xhat.contiguous(), ctx.pads) if g_xhat is not None else None
        g_z = g_z.contiguous() if g_z is not None else None
        if g_xp is None and g_z is None:
            dA, dB = [torch.zeros_like(w) for w in A], [torch.zeros_like(w) for w in B]
        else:
            dA, dB = ops.ista_backward(g, yp, mask_p, c, list(A), list(B), list(codes), list(resid), g_xp, g_z,
                                       dt, list(us), zp, za, lam, gam1, gam2, dg1, dg2, gzp, gza)
        _queue_backward_end()
        return (None, None, None, gzp, gza, dt.reshape(t.shape), dg1.reshape(g1.shape),
                dg2.reshape(g2.shape) if dg2 is not None else None, None, *dA, *dB)


# ------------------------------------------------------------------------------------------
# CDLNetVideo(residual=True) (reference model/net.py:199-212): a ResidualBlock
# rewrites the code after every iteration, so the sweep is a chain of per-iteration autograd nodes -- one ISTA
# iteration (the launches of _forward/_backward_generic_stepwise for one k), one block, ..., the synthesis.
@_arithmetic_aware
class _ISTAIteration(torch.autograd.Function):
    """(z_k | None, t_k (2,M..), A_k, B_k) -> z_{k+1} = ST(z_k - A_k(mask B_k z_k - yp), tau_k)."""

    @staticmethod
    def forward(ctx, zin, t_k, wA, wB, g, yp, mask_p, c):
        tau = ops.thresholds(t_k.reshape((1,) + tuple(t_k.shape)), c, g.N)[0]
        if zin is None:
            r = yp
            z = ops.analysis(g, yp, wA, 1.0, None, None, tau)
        else:
            r = ops.synthesis(g, zin, wB, 1.0, None, mask_p, yp)
            z = ops.analysis(g, r, wA, -1.0, zin, None, tau)
        ctx.g, ctx.first, ctx.t_shape = g, zin is None, tuple(t_k.shape)
        ctx.mask_p, ctx.c = mask_p, c
        ctx.save_for_backward(zin if zin is not None else yp.new_empty(0), r, z, wA, wB)
        return z

    @staticmethod
    def backward(ctx, gz):
        g = ctx.g
        zin, r, z, wA, wB = ctx.saved_tensors
        gk = gz.contiguous()
        dt_k = torch.zeros((2, g.M), device=gk.device, dtype=torch.float32)
        ops.tau_grad(g, gk, z, ctx.c, dt_k)
        dt_k = dt_k.reshape(ctx.t_shape)
        if ctx.first:
            dA = ops.wgrad(g, gk, r, 1.0, gate=z)
            return None, dt_k, dA, None, None, None, None, None
        q = ops.synthesis(g, gk, wA, -1.0, z, ctx.mask_p, None)
        dA = ops.wgrad(g, gk, r, -1.0, gate=z)
        dB = ops.wgrad(g, zin, q, 1.0)
        gzin = ops.analysis(g, q, wB, 1.0, gk, z, None)
        return gzin, dt_k, dA, dB, None, None, None, None


@_arithmetic_aware
class ResidualBlockFn(torch.autograd.Function):
    """(x, w1, w2) -> relu(conv2(relu(conv1 x)) + x)  (net.py:113-120)."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        g = ops.residual_geometry(x, w1)
        x = x.contiguous()                    # what the kernels read is what backward must see
        h, out = ops.residual_forward(g, x, w1, w2)
        ctx.g = g
        ctx.save_for_backward(x, h, out, w1, w2)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, h, out, w1, w2 = ctx.saved_tensors
        dx, dw1, dw2 = ops.residual_backward(ctx.g, x, h, out, w1, w2, g_out.contiguous())
        return dx, dw1, dw2


@_arithmetic_aware
class _Dictionary(torch.autograd.Function):
    """(z_K, B_0) -> xhat = post_process(D z_K)."""

    @staticmethod
    def forward(ctx, z, wB, g, mean, pads):
        xp = ops.synthesis(g, z, wB, 1.0)
        ctx.g, ctx.pads = g, pads
        ctx.save_for_backward(z, wB)
        return ops.postprocess(xp, mean, pads)

    @staticmethod
    def backward(ctx, g_xhat):
        z, wB = ctx.saved_tensors
        g_xp = ops.postprocess_bwd(g_xhat.contiguous(), ctx.pads)
        dB = ops.wgrad(ctx.g, z, g_xp, 1.0)
        gz = ops.analysis(ctx.g, g_xp, wB, 1.0, None, None, None)
        _queue_backward_end()
        return gz, dB, None, None, None


def run_residual(y, mask, c, t, A, B, s, blocks, all_codes=False):
    """CDLNetVideo.forward with residual=True: blocks[k] = (w1, w2) applied after iteration k.
    Returns (xhat, z) or, with all_codes, (xhat, z, ST outputs of every iteration) as forward_generator
    yields them (net.py:218-224: the code BEFORE its block)."""
    K = len(A)
    if torch.is_grad_enabled() and (y.requires_grad or (torch.is_tensor(mask) and mask.requires_grad)
                                    or (c is not None and c.requires_grad)):
        raise NotImplementedError("cdlnet_video_amd: gradients with respect to y / mask / sigma are not implemented")
    yp, mean, pads, mask_p = ops.preprocess(y, s, mask)
    N, C = yp.shape[:2]
    P = tuple(A[0].shape[2:])
    g = ops.Geometry.make(N, C, A[0].shape[0], yp.shape[2:], P, tuple(p // 2 for p in P), [s] * (yp.dim() - 2))
    z, shrunk = None, []
    for k in range(K):
        z = _ISTAIteration.apply(z, t[k], A[k], B[k], g, yp, mask_p, c)
        shrunk.append(z)
        z = ResidualBlockFn.apply(z, *blocks[k])
    xhat = _Dictionary.apply(z, B[0], g, mean, pads)
    return (xhat, z, *shrunk) if all_codes else (xhat, z)


def run_csr(y, mask, c, z_prev, z_after, t, g1, g2, A, B, s):
    """Front end of the neighbour branches; the no-neighbour branch is `run`."""
    cfg = {"K": len(A), "s": int(s)}
    return TemporalISTA.apply(y, mask, c, z_prev, z_after, t, g1, g2, cfg, *A, *B)


def run(y, mask, c, t, A, B, s, all_codes=False):
    """Convenience front end used by the modules."""
    cfg = {"K": len(A), "s": int(s), "all_codes": bool(all_codes)}
    return UnrolledISTA.apply(y, mask, c, t, cfg, *A, *B)
