"""Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-device.  Every image / clip is an independent unit of
the hot path (per-sample mean, per-sample sigma, no cross-sample op), so the batch axis shards
with no collective on the data path; training needs exactly one exchange per step: a sum
all-reduce of the parameter gradients.  The whole model is < 1 M floats (0.8-3.2 MB), so the
exchange is latency-bound: all gradients travel as ONE flat fp32 bucket (one RCCL call) rather
than per-parameter calls.
"""
import torch
import torch.distributed as dist


def _all_reduce_sum(t, group=None):
    """Sum all-reduce; a gloo group (CPU tests, rehearsals) gets device tensors staged through the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class GradientBucket:
    """Flat fp32 view of every trainable parameter's gradient; `sync()` averages it over ranks.

    The reverse sweep returns every filter gradient as a view of ONE allocation (dA_0..dA_{K-1}, dB_0..
    dB_{K-1}, in parameter order), so `adopt()` moves a step's gradients into the bucket with one copy per
    run of memory-adjacent gradients (2-3 small copies for a CDLNet), not one per parameter; afterwards
    every `.grad` IS its bucket view, so clipping, Adam and logging read the reduced values in place."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, device=ref.device, dtype=torch.float32)
        self.views, self.offsets = [], []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            self.offsets.append(off)
            off += p.numel()
        self.copies = 0                      # device copies issued by the last adopt() (tests / diagnostics)
        self.syncs = 0                       # sync() calls so far
        self._hook = None

    def attach(self):
        """Issue `sync()` from the END OF THE REVERSE SWEEP instead of from the training loop: the sweep's autograd
        node queues it on the engine (loop.on_backward_end), so it runs as soon as the last gradient has been
        accumulated -- the exchange is enqueued right behind the sweep's last kernel with no trainer code in
        between.  (The bucket is 0.8-3.2 MB and latency-bound, SURVEY.md section 5: it travels as one call; slicing
        it per iteration to overlap the sweep would multiply the latency term it is bound by.)"""
        from . import loop
        if self._hook is None:
            self._hook = loop.on_backward_end(self.sync)
        return self

    def detach(self):
        if self._hook is not None:
            self._hook()
            self._hook = None

    def adopt(self):
        """Make every .grad a view into the bucket (gradients that are not there yet are copied in, runs of
        adjacent ones with a single copy; parameters without a gradient get zeros)."""
        self.copies = 0
        n = len(self.params)
        i = 0
        while i < n:
            p, v = self.params[i], self.views[i]
            g = p.grad
            if g is None:
                v.zero_()
                i += 1
                continue
            if g.data_ptr() == v.data_ptr():
                i += 1
                continue
            # extend the run while the next gradient starts where this one ends (same storage, contiguous)
            j, end = i, g.data_ptr() + g.numel() * 4
            ok = g.is_contiguous() and g.dtype == torch.float32
            while ok and j + 1 < n:
                gn = self.params[j + 1].grad
                if (gn is None or not gn.is_contiguous() or gn.dtype != torch.float32 or gn.data_ptr() != end
                        or gn.untyped_storage().data_ptr() != g.untyped_storage().data_ptr()):
                    break
                end += gn.numel() * 4
                j += 1
            if ok and j > i:
                count = (end - g.data_ptr()) // 4
                src = torch.empty(0, device=g.device, dtype=torch.float32).set_(
                    g.untyped_storage(), g.storage_offset(), (count,), (1,))
                self.flat[self.offsets[i]:self.offsets[i] + count].copy_(src)
            else:
                v.copy_(g)
                j = i
            self.copies += 1
            i = j + 1
        for p, v in zip(self.params, self.views):
            p.grad = v

    def sync(self):
        """Sum over ranks, divide by world size (mean-of-means == global-batch mean for equal shards)."""
        self.adopt()
        self.syncs += 1
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            _all_reduce_sum(self.flat, self.group)
            self.flat.div_(dist.get_world_size(self.group))


def broadcast_parameters(module, src=0, group=None):
    """One-time parameter broadcast so every rank starts from rank `src`'s weights."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        tensors = [p.data for p in module.parameters()]
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def shard_batch(batch, rank=None, world=None):
    """Contiguous split of the leading (sample / clip) axis; ragged tails go to the low ranks."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    n = batch.shape[0]
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return batch[start:stop]


def all_reduce_scalar(value, op="mean", group=None, device=None):
    """Loss / PSNR logging across ranks.  A python float becomes a tensor on `device` (RCCL needs a device
    tensor; gloo takes either)."""
    if torch.is_tensor(value):
        t = value.detach().clone().reshape(1).to(torch.float32)
    else:
        t = torch.tensor([float(value)], dtype=torch.float32, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce_sum(t, group)
        if op == "mean":
            t /= dist.get_world_size(group)
    return t


def phase_consensus(psnr_sum, batches, bad, device, group=None):
    """What every rank must agree on before `fit` decides to backtrack: the phase PSNR averaged over ALL
    ranks' batches and whether ANY rank saw a nan / inf loss.  Returns (psnr, bad) identical on every rank."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return (psnr_sum / max(batches, 1)), bool(bad)
    nan_psnr = not (psnr_sum == psnr_sum) or psnr_sum in (float("inf"), float("-inf"))
    t = torch.tensor([0.0 if nan_psnr else float(psnr_sum), float(batches), 1.0 if (bad or nan_psnr) else 0.0],
                     dtype=torch.float64 if device is None or torch.device(device).type == "cpu" else torch.float32,
                     device=device)
    _all_reduce_sum(t, group)
    return float(t[0]) / max(float(t[1]), 1.0), bool(t[2] > 0)
