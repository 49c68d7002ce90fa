import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from conftest import rel_err
from oracle import cdl_oracle as O
import cdlnet_video_amd as cva
o = cva.ops
for (N, M, sp) in [(1, 48, (5, 24, 36)), (1, 32, (3, 17, 33)), (1, 64, (2, 20, 40))]:
    gen = torch.Generator().manual_seed(100 + M)
    x = torch.randn((N, M) + sp, generator=gen) * 0.5
    w1 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    w2 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    wgt = torch.randn((N, M) + sp, generator=gen)
    xo, w1o, w2o = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    h_ref = torch.relu(F.conv3d(xo, w1o, padding=1)); h_ref.retain_grad()
    pre2 = F.conv3d(h_ref, w2o, padding=1) + xo
    out_ref = torch.relu(pre2)
    (out_ref * wgt).sum().backward()
    geom = o.residual_geometry(x, w1)
    xd, w1d, w2d, gd = x.cuda(), w1.cuda(), w2.cuda(), wgt.cuda()
    h, out = o.residual_forward(geom, xd, w1d, w2d)
    print(M, sp, "h", rel_err(h.cpu(), h_ref.detach()), "out", rel_err(out.cpu(), out_ref.detach()),
          "support mismatch", ((h.cpu() != 0) != (h_ref.detach() != 0)).sum().item(), ((out.cpu() != 0) != (out_ref.detach() != 0)).sum().item())
    dh = o.synthesis(geom, gd, w2d, 1.0, gate=out)
    # h_ref.grad is the gradient AFTER the relu gate of h is NOT applied: grad wrt h (post-relu) = S(g2; w2)
    print("   dh", rel_err(dh.cpu(), h_ref.grad))
    for env in ("1", "0"):
        os.environ["CDL_MFMA_DENSE"] = env
        dx, dw1, dw2 = o.residual_backward(geom, xd, h, out, w1d, w2d, gd)
        print("   dense", env, "dx", rel_err(dx.cpu(), xo.grad), "dw1", rel_err(dw1.cpu(), w1o.grad), "dw2", rel_err(dw2.cpu(), w2o.grad))
    os.environ["CDL_MFMA_DENSE"] = "1"
