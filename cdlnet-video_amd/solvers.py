"""Init-time helpers (reference model/solvers.py).

`power_method` runs once in a constructor, on the CPU, on a 1xCx128x128 probe -- exactly
where the reference runs it (its nets are built on the CPU and moved afterwards,
train.py:188-198).  It is not part of the hot path and is the only place in this package
where ATen convolutions are used; every forward / backward / projection on the device goes
through libcdlnet_hip.so.
"""
import torch
import torch.nn.functional as F


def power_method(op, b, num_iter=1000, tol=1e-6, verbose=True):
    """Largest eigenvalue of `op` (solvers.py:3-22): returns (eig, b, tol_reached)."""
    prev = torch.zeros(1)
    reached = False
    eig = prev
    for it in range(num_iter):
        b = op(b)
        b = b / torch.linalg.vector_norm(b)
        eig = torch.sum(b * op(b))
        if verbose:
            print(f"i:{it:3d} \t |e_new - e_old|:{abs(eig - prev).item():2.2e}")
        if abs(eig - prev) < tol:
            reached = True
            break
        prev = eig
    return eig.item(), b, reached


def gram_operator(w_analysis, w_synthesis, stride, padding):
    """x -> D(A x) on the CPU for the spectral normalisation at construction."""
    if w_analysis.dim() == 4:
        conv, convT = F.conv2d, F.conv_transpose2d
    else:
        conv, convT = F.conv3d, F.conv_transpose3d

    def op(x):
        return convT(conv(x, w_analysis, stride=stride, padding=padding), w_synthesis,
                     stride=stride, padding=padding, output_padding=stride - 1)
    return op


def uball_project(W, dim=(2, 3)):
    """Projection of every filter onto the unit l2 ball (solvers.py:24-28), on the device."""
    from . import ops
    out = W.detach().clone().contiguous()
    flat_dims = tuple(range(2, W.dim()))
    if tuple(dim) != flat_dims:
        raise ValueError("uball_project: filters are (M, C, *P); dim must cover all P axes")
    return ops.project_filters_(out)
