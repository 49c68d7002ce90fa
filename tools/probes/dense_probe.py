"""Timing probes of the dense convolution kernels (CDL_DENSE_DEBUG bits: 1 no global loads, 2 no MFMAs, 4 no
epilogue / LDS stores; results are NOT valid when set, timing only).  `dense_probe.py 0` = the real kernels only
(used under rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'probes' else '.'))
import _ablate                                  # noqa: E402,F401  (probe build of the library)
import torch
import cdlnet_video_amd as cva
o = cva.ops
M, shape = 64, (16, 128, 128)
gen = torch.Generator().manual_seed(7)
x = (torch.randn((1, M) + shape, generator=gen) * 0.5).cuda()
w1 = (torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5).cuda()
g = o.residual_geometry(x, w1)
out = torch.empty_like(x)
def ev(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for dbg in (sys.argv[1:] or ("0", "1", "2", "3", "4", "6", "7")):
    os.environ["CDL_DENSE_DEBUG"] = dbg
    cva._lib.reload_options()
    print("dbg", dbg, "analysis ms", round(ev(lambda: o.analysis(g, x, w1, out=out)), 4),
          "wgrad ms", round(ev(lambda: o.wgrad(g, out, x, 1.0, gate=x)), 4), flush=True)
