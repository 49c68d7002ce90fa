"""Frame-recurrent inference drivers of the CSR nets (reference analyzemri.py:87-182).

The reference's loops draw the noise themselves; here the caller hands over the already noisy (and, for
JDD, already masked) frames, so the same sequence of network calls can be replayed deterministically.
Every call runs through the HIP kernels (CDLNet_CSR / CDLNet_CSRf2 in net.py); these functions only
carry the sparse code of one frame into the call for its neighbour.
"""
import torch


def _sigma_at(sigma, t):
    return sigma[t] if isinstance(sigma, (list, tuple)) else sigma


def _mask_at(mask, t):
    return mask[t] if isinstance(mask, (list, tuple)) else mask


@torch.no_grad()
def csr_inference_loop(net, frames, sigma=None, mask=1, bootstrap_curr=None):
    """analyzemri.py:87-156 with a CDLNet_CSR: frame 0 alone (second bank), frame 1 given z_0, frame 0
    again given z_1; then every frame t >= 1 given the running code.  `frames` is a sequence of
    (B, C, H, W) noisy frames; `sigma` / `mask` one value or one per frame.  `bootstrap_curr` is the
    realisation of frame 1 used for the bootstrap call (the reference draws a second, independent one
    for the loop; default: the same frame).  Returns the list of denoised frames."""
    if len(frames) < 2:
        raise ValueError("the recurrent loop needs at least two frames")
    s0, s1 = _sigma_at(sigma, 0), _sigma_at(sigma, 1)
    m0, m1 = _mask_at(mask, 0), _mask_at(mask, 1)
    _, z_prev = net(frames[0], None, s0, mask=m0)
    _, z_curr = net(frames[1] if bootstrap_curr is None else bootstrap_curr, z_prev, s1, mask=m1)
    first, z_prev = net(frames[0], z_curr, s0, mask=m0)
    results = [first]
    for t in range(1, len(frames)):
        xhat, z_prev = net(frames[t], z_prev, _sigma_at(sigma, t), mask=_mask_at(mask, t))
        results.append(xhat)
    return results


@torch.no_grad()
def csr_inference_v2(net, frames, sigma=None, mask=1):
    """analyzemri.py:162-182 with a CDLNet_CSRf2: a causal pass that records every frame's code
    (frame t given z_{t-1}), then a second pass in which frame t sees the recorded codes on both
    sides of it exactly as the reference indexes them: z_prev_list[t] (the code of frame t-1, None for
    t = 0) as `z_prev` and z_prev_list[t+1] (frame t's own first-pass code) as `z_after`."""
    T = len(frames)
    codes = [None] * (T + 2)
    for t in range(T):
        _, codes[t + 1] = net(frames[t], codes[t], None, _sigma_at(sigma, t), mask=_mask_at(mask, t))
    # second pass: frame t only needs recorded codes, so frames 1..T-1 (both neighbours given) go through the
    # network as ONE batch -- the samples of a batch are independent in every kernel, so the result per frame is
    # what the reference's frame-by-frame loop computes; frame 0 (no previous code) takes the other branch alone
    out = [net(frames[0], None, codes[1], _sigma_at(sigma, 0), mask=_mask_at(mask, 0))[0]]
    if T > 1:
        same_sigma = not isinstance(sigma, (list, tuple))
        same_mask = not isinstance(mask, (list, tuple))
        if same_sigma and same_mask and not torch.is_tensor(sigma) and not torch.is_tensor(mask):
            B = frames[0].shape[0]
            xb, _ = net(torch.cat(list(frames[1:])), torch.cat(codes[1:T]), torch.cat(codes[2:T + 1]), sigma, mask=mask)
            out.extend(xb[i * B:(i + 1) * B] for i in range(T - 1))
        else:
            out.extend(net(frames[t], codes[t], codes[t + 1], _sigma_at(sigma, t), mask=_mask_at(mask, t))[0]
                       for t in range(1, T))
    return out
