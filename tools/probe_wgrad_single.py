"""What a reverse stage that kept dA_k to itself would leave for the filter-gradient launch: k_wgrad2d with BOTH operand pairs
(dA_k and dB_k, as shipped) against the same kernel with ONE pair (dB_k alone; half of its waves idle, so an upper bound).
Flagship shape 64 x 64 x 256 x 256, P = 7, blocked fp32 codes.  One JSON line."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cdlnet_video_amd as cva  # noqa: E402

o = cva.ops
N, M, H, W, P = 64, 64, 256, 256, 7
g = o.Geometry.make(N, 1, M, (H, W), (P, P), (3, 3), 1)
dev = torch.device("cuda")
X0 = torch.randn(N * M * H * W, device=dev)
X1 = torch.randn(N * M * H * W, device=dev)
T0 = torch.randn(N, 1, H, W, device=dev)
T1 = torch.randn(N, 1, H, W, device=dev)
ws = o.fused_wgrad_workspace(g, dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


pair = timed(lambda: o.fused_wgrad(g, ws, X0, T0, 1.0, X1, T1, 1.0, layout="blocked"))
single = timed(lambda: o.fused_wgrad(g, ws, X0, T0, 1.0, layout="blocked"))
fat = N * M * H * W * 4
print(json.dumps({"probe": "k_wgrad2d pair vs single operand", "shape": [N, M, H, W], "layout": "blocked",
                  "pair_ms": round(pair, 4), "single_ms": round(single, 4),
                  "pair_TBps": round(2 * fat / pair / 1e9, 2), "single_TBps": round(fat / single / 1e9, 2),
                  "note": "launch + reduce, back to back (isolated, not in-step)"}))
