#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

Runs only in the build container (needs /root/reference).  The reference is
imported as-is; the only shim is an empty `torchvision` stub inserted into
sys.modules because `utils.py:4` imports `to_tensor` from a package that is not
installed and that the hot path never calls.  For the Gabor net
the private `_output_padding` call (gabor.py:59) needs `num_spatial_dims` on
torch >= 2; it is supplied through functools.partial on the bound method.

Each fixture stores inputs, parameters (state_dict keys), the reference outputs
(xhat, every code z_k where cheap, grads).  Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import functools
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.transforms.functional"].to_tensor = lambda *a, **k: None
    sys.path.insert(0, REF)
    import model.net as net          # noqa: E402
    import utils as ref_utils        # noqa: E402
    return net, ref_utils


def npify(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npify(arrays))
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def perturb(net, scale=0.05, t_lo=None, t_hi=None):
    """De-tie the K copies of W and randomise thresholds so every k is distinguishable."""
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n in ("t", "g"):
                continue
            p.add_(scale * p.abs().mean() * torch.randn_like(p))
        if t_lo is not None:
            net.t.uniform_(t_lo, t_hi)


def state(net):
    return {"sd/" + k: v.clone() for k, v in net.state_dict().items()}


def grads_of(net):
    return {"grad/" + n: (p.grad.clone() if p.grad is not None else None)
            for n, p in net.named_parameters()}


def patch_gabor(net, scale=0.15):
    """Supply num_spatial_dims to the private torch call and shrink alpha (stands in for 1/sqrt(L))."""
    seen = set()
    for mod in list(net.A) + list(net.B):
        mod._output_padding = functools.partial(mod._output_padding, num_spatial_dims=2)
        if id(mod.alpha) not in seen:
            seen.add(id(mod.alpha))
            with torch.no_grad():
                mod.alpha.mul_(scale)
        for p in (mod.alpha, mod.a, mod.w0, mod.psi):     # de-tie A from B and k from k'
            if id(p) not in seen:
                seen.add(id(p))
                with torch.no_grad():
                    p.add_(0.05 * p.abs().mean() * torch.randn_like(p))


def smooth(shape, gen):
    """Smooth-ish field in [0,1]: low-pass filtered uniform noise, min-max normalised."""
    x = torch.rand(shape, generator=gen)
    k = 5
    nd = len(shape) - 2
    w = torch.ones((shape[1], 1) + (k,) * nd) / k ** nd
    conv = torch.nn.functional.conv2d if nd == 2 else torch.nn.functional.conv3d
    x = conv(torch.nn.functional.pad(x, (k // 2,) * (2 * nd), mode="reflect"), w, groups=shape[1])
    x = (x - x.amin()) / (x.amax() - x.amin())
    return x


def main():
    os.makedirs(OUT, exist_ok=True)
    net_mod, ref_utils = import_reference()
    CDLNet, CDLNetVideo, GDLNet = net_mod.CDLNet, net_mod.CDLNetVideo, net_mod.GDLNet
    g = torch.Generator().manual_seed(1234)

    # ---- F1: 2-D s=1 C=1, float sigma, forward + all codes + grads -------------------
    torch.manual_seed(1)
    net = CDLNet(K=3, M=8, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    perturb(net, t_lo=2e-3, t_hi=2e-2)
    x = smooth((2, 1, 32, 32), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    codes = list(net.forward_generator(y, 25.0))
    xhat, z = net(y, 25.0)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f1_2d_s1", x=x, y=y, sigma=25.0, xhat=xhat, loss=loss,
         **{f"code{k}": c for k, c in enumerate(codes[:-1])}, gen_xhat=codes[-1],
         **state(net), **grads_of(net),
         hyper=np.array([3, 8, 5, 1, 1]), t0=5e-3)

    # ---- F2: 2-D s=2, odd size 33x31, per-sample sigma tensor ------------------------
    torch.manual_seed(2)
    net = CDLNet(K=4, M=6, P=7, s=2, C=1, t0=1e-2, adaptive=True, init=True)
    perturb(net, t_lo=1e-3, t_hi=3e-2)
    x = smooth((3, 1, 33, 31), g)
    sig = torch.tensor([15.0, 25.0, 40.0]).reshape(3, 1, 1, 1)
    y = x + torch.randn(x.shape, generator=g) * sig / 255
    xhat, z = net(y, sig)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f2_2d_s2_odd", x=x, y=y, sigma=sig, xhat=xhat, z=z, loss=loss, **state(net),
         **grads_of(net), hyper=np.array([4, 6, 7, 2, 1]))

    # ---- F3: C=3 + Bayer mask (JDD), per-sample sigma ---------------------------------
    torch.manual_seed(3)
    net = CDLNet(K=3, M=8, P=7, s=1, C=3, t0=5e-3, adaptive=True, init=True)
    perturb(net, t_lo=1e-3, t_hi=1e-2)
    x = smooth((2, 3, 24, 28), g)
    mask = ref_utils.gen_bayer_mask(x)
    sig = torch.tensor([5.0, 18.0]).reshape(2, 1, 1, 1)
    y = mask * (x + torch.randn(x.shape, generator=g) * sig / 255)
    xhat, z = net(y, sig, mask=mask)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f3_jdd_c3_mask", x=x, y=y, mask=mask, sigma=sig, xhat=xhat, z=z, loss=loss,
         **state(net), **grads_of(net), hyper=np.array([3, 8, 7, 1, 3]))

    # ---- F3b: C=3 + mask + stride 2 + odd size (mask is reflect padded too) -----------
    torch.manual_seed(33)
    net = CDLNet(K=3, M=6, P=5, s=2, C=3, t0=5e-3, adaptive=True, init=True)
    perturb(net, t_lo=1e-3, t_hi=1e-2)
    x = smooth((2, 3, 21, 19), g)
    mask = ref_utils.gen_bayer_mask(x)
    y = mask * (x + torch.randn(x.shape, generator=g) * 10 / 255)
    xhat, z = net(y, 10.0, mask=mask)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f3b_jdd_s2_odd", x=x, y=y, mask=mask, sigma=10.0, xhat=xhat, z=z, loss=loss,
         **state(net), **grads_of(net), hyper=np.array([3, 6, 5, 2, 3]))

    # ---- F4a: 3-D P=[5,5,5] s=1 -------------------------------------------------------
    torch.manual_seed(4)
    net = CDLNetVideo(K=3, M=6, P=[5, 5, 5], s=1, C=1, t0=5e-3, adaptive=True, depth=8, init=True)
    perturb(net, t_lo=1e-3, t_hi=1e-2)
    x = smooth((1, 1, 8, 16, 16), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    codes = list(net.forward_generator(y, 25.0))[:-1]   # last yield uses the 2-D post_process
    xhat, z = net(y, 25.0)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f4a_3d_p555", x=x, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss,
         **{f"code{k}": c for k, c in enumerate(codes)}, **state(net), **grads_of(net),
         hyper=np.array([3, 6, 5, 1, 1]), P3=np.array([5, 5, 5]))

    # ---- F4b: 3-D P=[9,9,5] s=2 (args3dmri.json shape family), per-sample sigma --------
    torch.manual_seed(5)
    net = CDLNetVideo(K=2, M=5, P=[9, 9, 5], s=2, C=1, t0=1e-2, adaptive=True, depth=8, init=True)
    perturb(net, t_lo=1e-3, t_hi=2e-2)
    x = smooth((2, 1, 8, 20, 12), g)
    sig = torch.tensor([20.0, 30.0]).reshape(2, 1, 1, 1, 1)
    y = x + torch.randn(x.shape, generator=g) * sig / 255
    xhat, z = net(y, sig)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f4b_3d_p995_s2", x=x, y=y, sigma=sig, xhat=xhat, z=z, loss=loss, **state(net),
         **grads_of(net), hyper=np.array([2, 5, 9, 2, 1]), P3=np.array([9, 9, 5]))

    # ---- F4c: 3-D s=2 with odd D/H/W (all three axes reflect padded: non-quirk unpad) ---
    torch.manual_seed(6)
    net = CDLNetVideo(K=2, M=4, P=[3, 5, 5], s=2, C=1, t0=5e-3, adaptive=True, depth=7, init=True)
    perturb(net, t_lo=1e-3, t_hi=1e-2)
    x = smooth((1, 1, 7, 13, 11), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    xhat, z = net(y, 25.0)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f4c_3d_s2_odd", x=x, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss, **state(net),
         **grads_of(net), hyper=np.array([2, 4, 5, 2, 1]), P3=np.array([3, 5, 5]))

    # ---- F5: Gabor, order 2, shared parameters, s=2 --------------------------------------
    torch.manual_seed(7)
    # init=False: the constructor's power method would call the torch>=2-incompatible
    # `_output_padding` before it can be wrapped; alpha is scaled by hand instead.
    net = GDLNet(K=3, M=4, P=7, s=2, C=1, t0=5e-3, order=2, adaptive=True,
                 shared="a_psi_w0_alpha", init=False)
    patch_gabor(net)
    with torch.no_grad():
        net.t.uniform_(1e-3, 1e-2)
    x = smooth((2, 1, 20, 22), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    xhat, z = net(y, 25.0)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    filt = {f"filt/A.{k}": net.A[k].get_filter(transpose=True) for k in range(3)}
    filt.update({f"filt/B.{k}": net.B[k].get_filter() for k in range(3)})
    save("f5_gabor_shared", x=x, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss, **state(net),
         **grads_of(net), **filt, hyper=np.array([3, 4, 7, 2, 1]), order=2,
         shared="a_psi_w0_alpha")

    # ---- F5b: Gabor, order 1, nothing shared, s=1, C=1 ----------------------------------
    torch.manual_seed(8)
    net = GDLNet(K=2, M=6, P=5, s=1, C=1, t0=5e-3, order=1, adaptive=False, shared="", init=False)
    patch_gabor(net)
    with torch.no_grad():
        net.t.uniform_(1e-3, 1e-2)
    x = smooth((1, 1, 16, 16), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    xhat, z = net(y, 25.0)            # adaptive=False -> sigma ignored
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    save("f5b_gabor_plain", x=x, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss, **state(net),
         **grads_of(net), hyper=np.array([2, 6, 5, 1, 1]), order=1, shared="")

    # ---- F6: negative thresholds (3-D trainer never projects) + exact zeros in the input ---
    torch.manual_seed(9)
    net = CDLNet(K=3, M=6, P=5, s=1, C=1, t0=0.0, adaptive=True, init=True)
    perturb(net)
    with torch.no_grad():
        net.t.uniform_(-1e-2, 1e-2)
    x = smooth((1, 1, 16, 16), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    xhat, z = net(y, 25.0)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    st_in = torch.linspace(-0.05, 0.05, 41)
    st_out = torch.stack([net_mod.ST(st_in, torch.tensor(tt)) for tt in (-0.01, 0.0, 0.02)])
    save("f6_negative_t", x=x, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss, st_in=st_in,
         st_out=st_out, st_t=np.array([-0.01, 0.0, 0.02]), **state(net), **grads_of(net),
         hyper=np.array([3, 6, 5, 1, 1]))

    # ---- F7: one full reference training step (train.py:76-102) --------------------------
    torch.manual_seed(10)
    net = CDLNet(K=3, M=8, P=5, s=1, C=1, t0=1e-2, adaptive=True, init=True)
    perturb(net, scale=0.5)          # push some filters outside the unit ball so project() acts
    before = state(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = smooth((4, 1, 24, 24), g)
    sig = 20 + 10 * torch.rand(4, 1, 1, 1, generator=g)
    y = x + torch.randn(x.shape, generator=g) * sig / 255
    opt.zero_grad()
    xhat, _ = net(y, sig, mask=1)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    gr = grads_of(net)
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 5e-2)
    opt.step()
    net.project()
    after = {"after/" + k: v.clone() for k, v in net.state_dict().items()}
    save("f7_train_step", x=x, y=y, sigma=sig, xhat=xhat, loss=loss, grad_norm=total, **before,
         **gr, **after, hyper=np.array([3, 8, 5, 1, 1]), lr=1e-3, clip=5e-2)

    # ---- F7b: 3-D training step (train3d.py:90-116: clip 1, no project) ------------------
    torch.manual_seed(11)
    net = CDLNetVideo(K=2, M=4, P=[3, 3, 3], s=1, C=1, t0=1e-2, adaptive=True, depth=4, init=True)
    perturb(net)
    before = state(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = smooth((2, 1, 4, 12, 12), g)
    sig = 20 + 10 * torch.rand(2, 1, 1, 1, 1, generator=g)
    y = x + torch.randn(x.shape, generator=g) * sig / 255
    opt.zero_grad()
    xhat, _ = net(y, sig)
    loss = torch.mean((x - xhat) ** 2)
    loss.backward()
    gr = grads_of(net)
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1)
    opt.step()
    after = {"after/" + k: v.clone() for k, v in net.state_dict().items()}
    save("f7b_train_step_3d", x=x, y=y, sigma=sig, xhat=xhat, loss=loss, grad_norm=total,
         **before, **gr, **after, hyper=np.array([2, 4, 3, 1, 1]), P3=np.array([3, 3, 3]),
         lr=1e-4, clip=1.0)

    # ---- F8: constructor determinism: seed -> (W, L) --------------------------------------
    torch.manual_seed(12)
    net = CDLNet(K=2, M=4, P=5, s=1, C=1, t0=1e-2, adaptive=True, init=True)
    save("f8_init_2d", **state(net), seed=12, hyper=np.array([2, 4, 5, 1, 1]), t0=1e-2)
    torch.manual_seed(13)
    net = CDLNetVideo(K=2, M=3, P=[3, 5, 5], s=1, C=1, t0=1e-2, adaptive=True, depth=4, init=True)
    save("f8_init_3d", **state(net), seed=13, hyper=np.array([2, 3, 5, 1, 1]),
         P3=np.array([3, 5, 5]), depth=4, t0=1e-2)

    # ---- F9: helpers: pads, uball_project, awgn / bayer mask ------------------------------
    import model.utils as mu
    import model.solvers as ms
    pads2 = np.array([mu.calc_pad_2D(h, w, s) for h, w, s in
                      [(33, 31, 2), (32, 32, 2), (17, 20, 4), (7, 9, 3), (5, 5, 1)]])
    pads3 = np.array([mu.calc_pad_3D(d, h, w, s) for d, h, w, s in
                      [(7, 13, 11, 2), (8, 16, 16, 2), (5, 6, 7, 4)]])
    W = torch.randn(5, 2, 3, 3, generator=g) * 0.6
    W3 = torch.randn(4, 1, 3, 3, 3, generator=g) * 0.3
    bm = ref_utils.gen_bayer_mask(torch.zeros(1, 3, 6, 8))
    save("f9_helpers", pads2=pads2, pads2_in=np.array([(33, 31, 2), (32, 32, 2), (17, 20, 4),
                                                        (7, 9, 3), (5, 5, 1)]),
         pads3=pads3, pads3_in=np.array([(7, 13, 11, 2), (8, 16, 16, 2), (5, 6, 7, 4)]),
         W=W, W_proj=ms.uball_project(W), W3=W3, bayer=bm)
    # NOTE: ms.uball_project(W3, dim=(2,3,4)) (CDLNetVideo.project, net.py:189-190) raises on
    # torch 2.10 ("linalg.matrix_norm: dim must be a 2-tuple"), so the 3-D projection has no
    # reference output to pin; W3 is stored as input only ("parity unpinned" for that op).


if __name__ == "__main__":
    main()
