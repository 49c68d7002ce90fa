import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import rel_err
from oracle import cdl_oracle as O
import cdlnet_video_amd as cva
o = cva.ops
for (N, C, M, sp, P) in [(1, 48, 48, (5, 24, 36), (3, 3, 3)), (1, 64, 64, (2, 20, 40), (3, 3, 3)), (1, 32, 48, (2, 20, 40), (3, 3, 3)),
                         (1, 48, 32, (2, 20, 40), (3, 3, 3))]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, C) + sp, generator=g); z = torch.randn((N, M) + sp, generator=g)
    zg = z * (torch.rand(z.shape, generator=g) < 0.5)
    w = torch.randn((M, C) + P, generator=g) * 0.1
    pad = tuple(p // 2 for p in P)
    geom = o.Geometry.make(N, C, M, sp, P, pad, 1)
    ref = O.synthesis(z, w, 1, pad)
    got = o.synthesis(geom, z.cuda(), w.cuda())
    print(N, C, M, sp, "synth plain", rel_err(got.cpu(), ref))
    refg = O.synthesis(z * (zg != 0), w, 1, pad) + x * (x > 0)
    os.environ["CDL_MFMA_DENSE"] = "0"
    got0 = o.synthesis(geom, z.cuda(), w.cuda())
    print("   valu synth", rel_err(got0.cpu(), ref))
    os.environ["CDL_MFMA_DENSE"] = "1"
    ref_a = O.analysis(x, w, 1, pad)
    print("   analysis", rel_err(o.analysis(geom, x.cuda(), w.cuda()).cpu(), ref_a))
    wv = w.clone().requires_grad_(True)
    u = torch.randn(z.shape, generator=g)
    (O.analysis(x, wv, 1, pad) * (u * (zg != 0))).sum().backward()
    print("   wgrad", rel_err(o.wgrad(geom, u.cuda(), x.cuda(), 1.0, gate=zg.cuda()).cpu(), wv.grad))
