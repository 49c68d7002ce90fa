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
# PRECISION applies to the fused kernels only: "split3" (fp32-grade, default) or "bf16".
BACKEND = "auto"
PRECISION = "split3"


def set_backend(name):
    global BACKEND
    if name not in ("auto", "generic"):
        raise ValueError(name)
    BACKEND = name


def set_precision(name):
    global PRECISION
    if name not in ops.PRECISION:
        raise ValueError(name)
    PRECISION = name


def _forward_generic(g, yp, mask_p, tau, A, B, keep_codes, keep_resid):
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
    return xp, z, codes, resid


def _forward_fused(g, yp, mask_p, tau, A, B, keep_codes, keep_resid):
    """The whole sweep from one C call (cdl_fused2d_forward): per iteration one fused launch + a thin
    assemble, enqueued back to back with no per-launch host work."""
    keep = keep_codes or keep_resid
    xp, z, codes, resid = ops.fused_forward(g, yp, mask_p, tau, A, B, keep, PRECISION)
    return xp, z, codes, (resid if keep_resid else [])


def _forward_fused_stepwise(g, yp, mask_p, tau, A, B, keep_codes, keep_resid):
    """Same sweep driven launch by launch from Python (kept for tests and experiments)."""
    K = len(A)
    frags = [ops.fused_prep(A[k], B[(k + 1) % K]) for k in range(K)]   # last one pairs A_{K-1} with D = B_0
    patches = ops.fused_patches(g, yp.device)
    codes, resid = [], []
    r, z = yp, None
    for k in range(K):
        z = ops.fused_iter(g, r, z, tau[k], frags[k], 1.0 if k == 0 else -1.0, patches, PRECISION)
        if keep_codes or k == 0:
            codes.append(z)
        if k < K - 1:
            r = ops.fused_assemble(g, patches, mask_p, yp, 1.0)
            if keep_resid:
                resid.append(r)
    xp = ops.fused_assemble(g, patches, None, None, 1.0)
    return xp, z, codes, resid


def _backward_generic(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt):
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


def _backward_fused(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt):
    """Reverse sweep from one C call (cdl_fused2d_backward): per iteration one stage launch (2 fat
    reads, 1 fat write), a thin assemble, and one MFMA filter-gradient launch (2 fat reads)."""
    return ops.fused_backward(g, yp, mask_p, c, list(A), list(B), list(codes), list(resid), g_xp, g_z, dt,
                              PRECISION)


def _backward_fused_stepwise(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt):
    """Same reverse sweep driven launch by launch from Python (kept for tests and experiments)."""
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
        du = ops.fused_stage_bwd(g, thin, du_next, codes[k], frags, patches, dtp, k >= 1, prec)
        ops.fused_dtau_reduce(g, dtp, c, dt[k])
        if k >= 1:
            q = ops.fused_assemble(g, patches, mask_p, None, -1.0)
            dA[k], dB[k] = ops.fused_wgrad(g, ws, du, resid[k - 1], -1.0, codes[k - 1], q, 1.0, prec)
            thin = q
        else:
            (dA[0],) = ops.fused_wgrad(g, ws, du, yp, 1.0, precision=prec)[:1]
        du_next = du
    return dA, dB


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
        keep = any(ctx.needs_input_grad)          # all False under torch.no_grad()
        want_codes = cfg.get("all_codes", False)
        ctx.fused = BACKEND == "auto" and ops.fused_supported(g)
        sweep = _forward_fused if ctx.fused else _forward_generic
        xp, z, codes, resid = sweep(g, yp, mask_p, tau, A, B, keep or want_codes, keep)
        xhat = ops.postprocess(xp, mean, pads)

        ctx.geom, ctx.pads, ctx.K = g, pads, K
        ctx.has_mask, ctx.has_c = mask_p is not None, c is not None
        if keep:
            ctx.save_for_backward(yp, mask_p if mask_p is not None else yp.new_empty(0),
                                  c if c is not None else yp.new_empty(0), t, *weights,
                                  *codes, *resid)
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
        resid = saved[4 + 3 * K:]                     # r_1..r_{K-1}
        dt = torch.zeros((K, 2, g.M), device=yp.device, dtype=torch.float32)
        g_xp = ops.postprocess_bwd(g_xhat.contiguous(), ctx.pads) if g_xhat is not None else None
        if g_z is not None:
            g_z = g_z.contiguous()
        sweep = _backward_fused if (ctx.fused and g_xp is not None) else _backward_generic
        dA, dB = sweep(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt)

        return (None, None, None, dt.reshape(t.shape), None, *dA, *dB)


def run(y, mask, c, t, A, B, s, all_codes=False):
    """Convenience front end used by the modules."""
    cfg = {"K": len(A), "s": int(s), "all_codes": bool(all_codes)}
    return UnrolledISTA.apply(y, mask, c, t, cfg, *A, *B)
