"""TEST INFRASTRUCTURE -- PyTorch fp32 CPU restatement of the CDLNet-video hot path.

Never imported by the product package; see oracle/__init__.py.

Parity status: PINNED by tests/golden/*.npz (generated from the unmodified
reference by tools/make_golden.py; tests/test_oracle_golden.py replays them).

Everything is functional: parameters travel in a plain dict that uses the
reference's state_dict key names (`t`, `A.{k}.weight`, `B.{k}.weight`, and for
the Gabor net `A.{k}.alpha|a|w0|psi`), so a reference checkpoint's
`net_state_dict` can be fed in unchanged.

Reference lines restated (all relative to /root/reference):
  soft_threshold        model/net.py:11-14
  split_pad/stride_pads model/utils.py:35-51, 103-111
  preprocess            model/utils.py:5-22, 70-87
  postprocess           model/utils.py:24-33, 58-68, 89-101, 113-122
  ista_codes/ista       model/net.py:76-104 (2-D), 192-227 (3-D), 659-687 (Gabor)
  gabor_bank            model/gabor.py:7-28, 46-51
  power_iteration       model/solvers.py:3-22
  unit_ball             model/solvers.py:24-28
  init_dictionary       model/net.py:30-57, 136-176
  init_gabor            model/net.py:589-642
  project_              model/net.py:66-74, 184-190, 652-657
  awgn / bayer_mask     utils.py:13-55
  train_step            train.py:76-102, train3d.py:90-116
  psnr                  analyze.py:104, analyze3d.py:131-133
  hh_filter / nle_mad   model/wvlt.py:13-41, model/nle.py:17-27 (taps restated: PyWavelets is not installed)
  mcsure_loss_and_grads train.py:87-93 (restated from the text; the reference has it inline in fit())
  prox_csr / prox_csr_f2  model/net.py:229-262
  ista_csr              model/net.py:426-463 (CDLNet_CSR.forward), 525-568 (CDLNet_CSRf2.forward)
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- shrinkage
def soft_threshold(x, t):
    """sign(x) * max(|x| - t, 0); t broadcasts; negative t is legal (3-D trainer never projects)."""
    return torch.sign(x) * torch.relu(torch.abs(x) - t)


# --------------------------------------------------------------------------- padding
def split_pad(length, s):
    """(before, after) so that length + before + after is a multiple of s; floor/ceil split."""
    rem = length % s
    if rem == 0:
        return (0, 0)
    extra = s - rem
    lo = extra // 2
    return (lo, extra - lo)


def stride_pads(spatial, s):
    """F.pad-ordered tuple (last dim first) for spatial dims, e.g. (l, r, t, b[, f, k])."""
    out = []
    for length in reversed(tuple(spatial)):
        out.extend(split_pad(int(length), s))
    return tuple(out)


def preprocess(y, s, mask=None):
    """Per-sample mean removal (mask aware) and reflect padding to a multiple of the stride.

    Returns (yp, mean, pads, mask_p); mask_p is None when no mask tensor was given.
    """
    red = tuple(range(1, y.dim()))
    if mask is not None:
        mean = y.sum(dim=red, keepdim=True) / mask.sum(dim=red, keepdim=True)
        x = mask * (y - mean)
    else:
        mean = y.mean(dim=red, keepdim=True)
        x = y - mean
    pads = stride_pads(y.shape[2:], s)
    mask_p = mask
    if any(pads):
        x = F.pad(x, pads, mode="reflect")
        if mask is not None:
            mask_p = F.pad(mask, pads, mode="reflect")
    return x, mean, pads, mask_p


def crop(x, pads):
    """Inverse of the reflect padding (correct for every pad pattern).

    NOTE: the reference's `unpad_3d` (model/utils.py:113-122) only branches on
    (pad_back, pad_right) and so returns an empty / uncropped tensor for some
    mixed pad patterns; fixtures avoid those and this function does the right
    thing everywhere (documented deviation, DESIGN.md "reference quirks").
    """
    nsp = len(pads) // 2
    idx = [slice(None), slice(None)]
    for d in range(nsp):                       # d-th spatial dim, pads are last-dim-first
        lo, hi = pads[2 * (nsp - 1 - d)], pads[2 * (nsp - 1 - d) + 1]
        size = x.shape[2 + d]
        idx.append(slice(lo, size - hi))
    return x[tuple(idx)]


def postprocess(xp, mean, pads):
    return crop(xp, pads) + mean


# --------------------------------------------------------------------------- operators
def _conv_pad(P, ndim):
    """Zero padding used by the reference: (P-1)//2 in 2-D (net.py:32), P[i]//2 in 3-D (net.py:138)."""
    if ndim == 2:
        P = (P, P) if isinstance(P, int) else tuple(P)
        return tuple((p - 1) // 2 for p in P)
    P = (P, P, P) if isinstance(P, int) else tuple(P)
    return tuple(p // 2 for p in P)


def analysis(x, w, s, pad):
    """A: strided correlation C -> M, bias free."""
    fn = F.conv2d if x.dim() == 4 else F.conv3d
    return fn(x, w, stride=s, padding=pad)


def synthesis(z, w, s, pad):
    """B: transposed strided correlation M -> C, output_padding = s-1."""
    fn = F.conv_transpose2d if z.dim() == 4 else F.conv_transpose3d
    return fn(z, w, stride=s, padding=pad, output_padding=s - 1)


def _thresholds(t, k, c):
    """tau_k = t[k,0] + c * t[k,1]; c is 0, a float, or an (N,1,..) tensor."""
    return t[k, :1] + c * t[k, 1:2]


def shrink_on_support(u, tau, code):
    """ST(u, tau) with the support and signs PRESCRIBED by `code` (another evaluation's z_{k+1}):
    [code != 0] * (u - sign(code) * tau).  Equals soft_threshold(u, tau) wherever the two evaluations
    agree on the support; given the support, the net is a smooth (polynomial) function of its
    parameters, so gradients of two fp32 evaluations can be compared tightly (tests only)."""
    return (code != 0).to(u.dtype) * (u - torch.sign(code) * tau)


def ista_codes(yp, mask_p, A, B, t, c, s, pad, supports=None):
    """Generator over the K sparse codes z_1..z_K (reference forward_generator).
    supports: optional K codes whose support / signs replace the shrinkage's own (shrink_on_support)."""
    K = len(A)
    st = (lambda u, tau, k: soft_threshold(u, tau)) if supports is None else \
         (lambda u, tau, k: shrink_on_support(u, tau, supports[k]))
    z = st(analysis(yp, A[0], s, pad), _thresholds(t, 0, c), 0)
    yield z
    for k in range(1, K):
        resid = synthesis(z, B[k], s, pad)
        if mask_p is not None:
            resid = mask_p * resid
        resid = resid - yp
        z = st(z - analysis(resid, A[k], s, pad), _thresholds(t, k, c), k)
        yield z


def _weights_from_state(sd, K, gabor=None):
    """Pull the A/B filter lists out of a state_dict-style mapping."""
    if gabor is None:
        A = [sd[f"A.{k}.weight"] for k in range(K)]
        B = [sd[f"B.{k}.weight"] for k in range(K)]
        return A, B
    P = gabor
    A = [gabor_bank(sd[f"A.{k}.alpha"], sd[f"A.{k}.a"], sd[f"A.{k}.w0"], sd[f"A.{k}.psi"], P, True)
         for k in range(K)]
    B = [gabor_bank(sd[f"B.{k}.alpha"], sd[f"B.{k}.a"], sd[f"B.{k}.w0"], sd[f"B.{k}.psi"], P, False)
         for k in range(K)]
    return A, B


def ista(sd, y, *, K, P, s=1, sigma=None, adaptive=False, mask=None, ndim=2, gabor=False,
         all_codes=False, supports=None):
    """Full forward: returns (xhat, z_K) or (xhat, [z_1..z_K]) when all_codes.

    `sd` maps the reference's state_dict keys to tensors.  `mask=None` is the
    reference's scalar `mask=1`.
    """
    yp, mean, pads, mask_p = preprocess(y, s, mask)
    c = 0.0 if (sigma is None or not adaptive) else sigma / 255.0
    pad = _conv_pad(P, ndim)
    A, B = _weights_from_state(sd, K, gabor=P if gabor else None)
    codes = list(ista_codes(yp, mask_p, A, B, sd["t"], c, s, pad, supports))
    xp = synthesis(codes[-1], B[0], s, pad)          # D is B[0] (net.py:34)
    xhat = postprocess(xp, mean, pads)
    return (xhat, codes) if all_codes else (xhat, codes[-1])


# --------------------------------------------------------------------------- residual blocks (3-D)
def residual_block(x, w1, w2):
    """model/net.py:113-120: relu(conv2(relu(conv1 x)) + x), bias-free 'same' convolutions, stride 1."""
    pad = tuple(p // 2 for p in w1.shape[2:])
    h = torch.relu(F.conv3d(x, w1, padding=pad))
    return torch.relu(F.conv3d(h, w2, padding=pad) + x)


def ista_video_residual(sd, y, *, K, P, s=1, sigma=None, adaptive=False, mask=None, all_codes=False):
    """CDLNetVideo.forward with residual=True (model/net.py:192-212): a ResidualBlock rewrites the code after
    every iteration.  all_codes: also the ST outputs (what forward_generator yields, net.py:218-224)."""
    yp, mean, pads, mask_p = preprocess(y, s, mask)
    c = 0.0 if (sigma is None or not adaptive) else sigma / 255.0
    pad = _conv_pad(P, 3)
    A, B = _weights_from_state(sd, K)
    blk = lambda k, z: residual_block(z, sd[f"residual_blocks.{k}.conv1.weight"], sd[f"residual_blocks.{k}.conv2.weight"])
    shrunk = [soft_threshold(analysis(yp, A[0], s, pad), _thresholds(sd["t"], 0, c))]
    z = blk(0, shrunk[0])
    for k in range(1, K):
        resid = synthesis(z, B[k], s, pad)
        if mask_p is not None:
            resid = mask_p * resid
        shrunk.append(soft_threshold(z - analysis(resid - yp, A[k], s, pad), _thresholds(sd["t"], k, c)))
        z = blk(k, shrunk[-1])
    xhat = postprocess(synthesis(z, B[0], s, pad), mean, pads)
    return (xhat, z, shrunk) if all_codes else (xhat, z)


# --------------------------------------------------------------------------- blind noise level (MAD)
# PyWavelets' 'bior4.4' filter bank (pywt.Wavelet('bior4.4').filter_bank: dec_lo, dec_hi, rec_lo, rec_hi),
# restated from the published CDF 9/7 pair because pywt is not installed here ("parity unpinned" against
# the library itself; tests pin the table through the perfect-reconstruction identity).
BIOR44 = {
    "dec_lo": [0.0, 0.03782845550726404, -0.023849465019556843, -0.11062440441843718, 0.37740285561283066,
               0.8526986790088938, 0.37740285561283066, -0.11062440441843718, -0.023849465019556843,
               0.03782845550726404],
    "dec_hi": [0.0, -0.06453888262869706, 0.04068941760916406, 0.41809227322161724, -0.7884856164055829,
               0.41809227322161724, 0.04068941760916406, -0.06453888262869706, 0.0, 0.0],
    "rec_lo": [0.0, -0.06453888262869706, -0.04068941760916406, 0.41809227322161724, 0.7884856164055829,
               0.41809227322161724, -0.04068941760916406, -0.06453888262869706, 0.0, 0.0],
    "rec_hi": [0.0, -0.03782845550726404, -0.023849465019556843, 0.11062440441843718, 0.37740285561283066,
               -0.8526986790088938, 0.37740285561283066, 0.11062440441843718, -0.023849465019556843,
               -0.03782845550726404],
}


def hh_filter():
    """model/wvlt.py:13-41 reduced to the one band nle_mad takes: filter_bank_2D('bior4.4')[0][3:4], i.e.
    outer(dec_hi, dec_hi) flipped along both axes (`nonsep` flips so that conv2d's correlation convolves)."""
    hi = torch.tensor(BIOR44["dec_hi"], dtype=torch.float32)
    return torch.outer(hi, hi).flip(0, 1)[None, None]


def nle_mad(y):
    """model/nle.py:17-27."""
    C = y.shape[1]
    hh = torch.cat([hh_filter()] * C)
    band = F.conv2d(y, hh, stride=2, groups=C)
    return (torch.median(band.abs().reshape(y.shape[0], -1), dim=1)[0] / 0.6745).reshape(-1, 1, 1, 1)


# --------------------------------------------------------------------------- CSR temporal variants
def prox_csr(u, z_prev, lam, gam):
    """Two nested shrinkages around the neighbour frame's code (net.py:229-242).  The evaluation
    order is the reference's, term by term: the maps are discontinuous for negative thresholds, so a
    re-association that moves a value across 0 by one ulp changes the result by |threshold|."""
    ls = lam * torch.sign(z_prev)
    return soft_threshold(soft_threshold(u - z_prev - ls, lam * gam) + z_prev + ls, lam)


def prox_csr_f2(u, z_prev, z_after, lam, gam1, gam2):
    """Three nested shrinkages around both neighbour codes (net.py:244-262), reference order."""
    ca = z_prev + lam * torch.sign(z_prev) + lam * gam2 * torch.sign(z_prev - z_after)
    cb = z_after + lam * torch.sign(z_after) + lam * gam1 * torch.sign(z_after - z_prev)
    inner = soft_threshold(u - ca, gam1 * lam)
    mid = soft_threshold(inner - cb + lam * gam1 * torch.sign(u - ca), gam2 * lam)
    return soft_threshold(mid + cb - lam * gam1 * torch.sign(u - ca), lam)


def ista_csr(sd, y, z_prev=None, z_after=None, *, K, P, s=1, sigma=None, adaptive=False, mask=None,
             variant="csr"):
    """CDLNet_CSR.forward (variant "csr": A2/B2/t2 when there is no neighbour, else A/B/t/g) and
    CDLNet_CSRf2.forward (variant "f2": one bank, thresholds t/g1/g2, four branches).  Returns
    (xhat, z_K); D is B[0] in every branch (net.py:386, 460)."""
    yp, mean, pads, mask_p = preprocess(y, s, mask)
    c = 0.0 if (sigma is None or not adaptive) else sigma / 255.0
    pad = _conv_pad(P, 2)
    A, B = _weights_from_state(sd, K)
    t = sd["t"]
    if variant == "csr":
        if z_after is not None:
            raise ValueError("CDLNet_CSR has no z_after")
        if z_prev is None:
            A_, B_, t = [sd[f"A2.{k}.weight"] for k in range(K)], [sd[f"B2.{k}.weight"] for k in range(K)], sd["t2"]
            shrink = lambda u, k: soft_threshold(u, _thresholds(t, k, c))
        else:
            A_, B_ = A, B
            shrink = lambda u, k: prox_csr(u, z_prev, _thresholds(t, k, c), _thresholds(sd["g"], k, c))
    else:
        A_, B_ = A, B
        if z_prev is not None and z_after is not None:
            shrink = lambda u, k: prox_csr_f2(u, z_prev, z_after, _thresholds(t, k, c),
                                              _thresholds(sd["g1"], k, c), _thresholds(sd["g2"], k, c))
        elif z_prev is not None:
            shrink = lambda u, k: prox_csr(u, z_prev, _thresholds(t, k, c), _thresholds(sd["g1"], k, c))
        elif z_after is not None:
            shrink = lambda u, k: prox_csr(u, z_after, _thresholds(t, k, c), _thresholds(sd["g2"], k, c))
        else:
            shrink = lambda u, k: soft_threshold(u, _thresholds(t, k, c))
    z = shrink(analysis(yp, A_[0], s, pad), 0)
    for k in range(1, K):
        resid = synthesis(z, B_[k], s, pad)
        if mask_p is not None:
            resid = mask_p * resid
        z = shrink(z - analysis(resid - yp, A_[k], s, pad), k)
    xhat = postprocess(synthesis(z, B[0], s, pad), mean, pads)
    return xhat, z


# --------------------------------------------------------------------------- Gabor dictionary
def gabor_bank(alpha, a, w0, psi, P, transpose=False):
    """Filters (M,C,P,P) = sum_o alpha_o * exp(-|a_o*(x-x0)|^2) * cos(w0_o.(x-x0) + psi_o).

    The analysis ("transpose") filter uses (-w0, -psi).
    """
    if transpose:
        w0, psi = -w0, -psi
    ax = torch.arange(P, dtype=alpha.dtype) - (P - 1) / 2.0
    gy, gx = torch.meshgrid(ax, ax, indexing="ij")          # (P,P): row offset, col offset
    a0, a1 = a[..., 0, None, None], a[..., 1, None, None]   # (order,M,C,1,1)
    f0, f1 = w0[..., 0, None, None], w0[..., 1, None, None]
    envelope = torch.exp(-((a0 * gy) ** 2 + (a1 * gx) ** 2))
    carrier = torch.cos(f0 * gy + f1 * gx + psi[..., None, None])
    return (alpha * envelope * carrier).sum(dim=0)


def gabor_alias(sd, K, shared):
    """Re-create the reference's parameter sharing (net.py:607-622) on a flat state dict:
    shared tensors become the *same* object so gradients accumulate like aliased Parameters."""
    out = dict(sd)
    for k in range(1, K):
        if "alpha" in shared:
            out[f"A.{k}.alpha"] = out["A.0.alpha"]
            if k > 1:                                   # B[0] (= D) keeps its own scale
                out[f"B.{k}.alpha"] = out["B.1.alpha"]
        for tag, field in (("a_", "a"), ("w0", "w0"), ("psi", "psi")):
            if tag in shared:
                out[f"A.{k}.{field}"] = out[f"A.0.{field}"]
                out[f"B.{k}.{field}"] = out[f"B.0.{field}"]
    for f in ("alpha", "a", "w0", "psi"):
        if f"D.{f}" in out:
            out[f"D.{f}"] = out[f"B.0.{f}"]
    return out


# --------------------------------------------------------------------------- init / projection
def power_iteration(op, b, iters=200, tol=1e-6):
    """Largest eigenvalue of `op` by the power method; stops on |delta eig| < tol."""
    prev = 0.0
    eig = 0.0
    for _ in range(iters):
        b = op(b)
        b = b / torch.linalg.vector_norm(b)
        eig = float(torch.sum(b * op(b)))
        if abs(eig - prev) < tol:
            break
        prev = eig
    return eig


def unit_ball(w, dims):
    """Scale each filter whose l2 norm over `dims` exceeds 1 back onto the unit sphere."""
    nrm = torch.linalg.vector_norm(w, dim=dims, keepdim=True)
    return w * torch.clamp(1.0 / nrm, max=1.0)


def init_dictionary(K, M, P, s, C, t0, ndim=2, depth=3, init=True, with_g=True):
    """Reference constructor semantics under the *current* torch RNG state.

    Call right after `torch.manual_seed(seed)` *and after* consuming the same
    RNG draws the reference constructor consumes before `randn(M,C,P..)`: the
    K Conv + K ConvTranspose default inits.  `tools/make_golden.py` stores the
    resulting tensors instead of relying on that, so tests do not depend on it.
    """
    if ndim == 2:
        shape = (M, C, P, P)
        tshape = (K, 2, M, 1, 1)
        probe = (1, C, 128, 128)
    else:
        Pt = (P, P, P) if isinstance(P, int) else tuple(P)
        shape = (M, C) + Pt
        tshape = (K, 2, M, 1, 1, 1)
        probe = (1, C, depth, 128, 128)
    W = torch.randn(shape)
    pad = _conv_pad(P, ndim)
    L = 1.0
    if init:
        L = power_iteration(lambda x: synthesis(analysis(x, W, s, pad), W, s, pad), torch.rand(probe))
        W = W / math.sqrt(L)
    sd = {"t": t0 * torch.ones(tshape)}
    if ndim == 2 and with_g:
        sd["g"] = t0 * torch.ones(tshape)
    for k in range(K):
        sd[f"A.{k}.weight"] = W.clone()
        sd[f"B.{k}.weight"] = W.clone()
    sd["D.weight"] = sd["B.0.weight"]
    return sd, L


def project_(sd, K, ndim=2, gabor=False):
    """In-place projection after an optimiser step: t >= 0, filters into the unit ball."""
    sd["t"].clamp_(min=0.0)
    if gabor:
        return
    dims = (2, 3) if ndim == 2 else (2, 3, 4)
    for k in range(K):
        sd[f"A.{k}.weight"].copy_(unit_ball(sd[f"A.{k}.weight"], dims))
        sd[f"B.{k}.weight"].copy_(unit_ball(sd[f"B.{k}.weight"], dims))


# --------------------------------------------------------------------------- data side
def awgn(x, sigma, generator=None):
    """y = x + n*sigma/255; sigma a number, or (lo, hi) -> per-sample U(lo,hi) shaped (N,1,..)."""
    if isinstance(sigma, (list, tuple)):
        shape = (x.shape[0],) + (1,) * (x.dim() - 1)
        sigma = sigma[0] + (sigma[1] - sigma[0]) * torch.rand(shape, generator=generator)
    noise = torch.randn(x.shape, generator=generator)
    return x + noise * (sigma / 255.0), sigma


def bayer_mask(x):
    """RGGB mask on a 3-channel (N,3,H,W) image (utils.py:13-19)."""
    m = torch.zeros_like(x)
    m[:, 0, 0::2, 0::2] = 1
    m[:, 1, 0::2, 1::2] = 1
    m[:, 1, 1::2, 0::2] = 1
    m[:, 2, 1::2, 1::2] = 1
    return m


def psnr(x, xhat):
    """-10 log10(mean((x-xhat)^2)), peak 1.0, one number over the whole tensor."""
    return -10.0 * math.log10(float(torch.mean((x - xhat) ** 2)))


# --------------------------------------------------------------------------- training step
def trainable(sd, K, gabor=False):
    """Ordered list of (key, leaf tensor) exactly as `net.parameters()` de-duplicates them."""
    keys = ["t"]
    if "g" in sd:
        keys.append("g")
    if gabor:
        seen = set()
        for bank in ("A", "B"):
            for k in range(K):
                for f in ("alpha", "a", "w0", "psi"):
                    key = f"{bank}.{k}.{f}"
                    if id(sd[key]) not in seen:
                        seen.add(id(sd[key]))
                        keys.append(key)
    else:
        keys += [f"A.{k}.weight" for k in range(K)] + [f"B.{k}.weight" for k in range(K)]
    return keys


def loss_and_grads(sd, x, y, *, K, P, s, sigma, adaptive, mask=None, ndim=2, gabor=False, supports=None,
                   code_weight=0.0):
    """MSE(x, xhat) [+ code_weight * mean(z_K^2)] and d loss / d parameter for every trainable key (dict).
    supports: see ista_codes."""
    keys = trainable(sd, K, gabor)
    leaves = {}
    work = dict(sd)
    for key in keys:
        leaf = sd[key].detach().clone().requires_grad_(True)
        leaves[key] = leaf
        work[key] = leaf
    # keep aliasing semantics (shared Gabor parameters) for non-leaf duplicates
    for key, val in sd.items():
        for lk in keys:
            if val is sd[lk] and key != lk:
                work[key] = leaves[lk]
    xhat, zK = ista(work, y, K=K, P=P, s=s, sigma=sigma, adaptive=adaptive, mask=mask, ndim=ndim,
                    gabor=gabor, supports=supports)
    loss = torch.mean((x - xhat) ** 2)
    if code_weight:
        loss = loss + code_weight * torch.mean(zK ** 2)
    loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else None) for k in keys}
    return float(loss.detach()), grads, xhat.detach()


def mcsure_loss_and_grads(sd, obsrv, b, *, K, P, s, sigma, adaptive, mask=None, ndim=2, h=1e-3, supports=None,
                          supports_b=None):
    """The unsupervised objective of train.py:87-93 (one extra forward at obsrv + h*b) and its gradients:
    mean((obsrv - xhat)^2) + 2 * mean((sigma/255)^2 * b * (xhat_b - xhat)) / h.
    supports / supports_b: prescribed code supports of the two passes (see ista_codes)."""
    keys = trainable(sd, K, False)
    work = dict(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work.update(leaves)
    kw = dict(K=K, P=P, s=s, sigma=sigma, adaptive=adaptive, mask=mask, ndim=ndim)
    xhat, _ = ista(work, obsrv, supports=supports, **kw)
    xhat_b, _ = ista(work, obsrv.clone() + h * b, supports=supports_b, **kw)
    div = 2.0 * torch.mean(((sigma / 255.0) ** 2) * b * (xhat_b - xhat)) / h
    loss = torch.mean((obsrv - xhat) ** 2) + div
    loss.backward()
    return float(loss.detach()), {k: leaves[k].grad for k in keys}


def train_step(sd, x, y, *, K, P, s, sigma, adaptive, mask=None, ndim=2, lr=1e-3, clip=None,
               project=True):
    """One reference optimiser step (fresh Adam state): returns loss, grads, updated params."""
    keys = trainable(sd, K)
    params = [sd[k].detach().clone().requires_grad_(True) for k in keys]
    work = dict(sd)
    for k, p in zip(keys, params):
        work[k] = p
    work["D.weight"] = work["B.0.weight"]
    opt = torch.optim.Adam(params, lr=lr)
    xhat, _ = ista(work, y, K=K, P=P, s=s, sigma=sigma, adaptive=adaptive, mask=mask, ndim=ndim)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    loss = loss.detach()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in zip(keys, params)}
    if clip is not None:
        torch.nn.utils.clip_grad_norm_(params, clip)
    opt.step()
    new = {k: p.detach().clone() for k, p in zip(keys, params)}
    if project:
        with torch.no_grad():
            project_(new, K, ndim)
    return float(loss), grads, new
