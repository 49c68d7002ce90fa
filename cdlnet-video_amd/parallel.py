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


class GradientBucket:
    """Flat fp32 view of every trainable parameter's gradient; `sync()` averages it over ranks."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, device=ref.device, dtype=torch.float32)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def adopt(self):
        """Make every .grad a view into the bucket, so backward writes land in it directly."""
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            elif p.grad is None:
                v.zero_()
            p.grad = v

    def sync(self):
        """Sum over ranks, divide by world size (mean-of-means == global-batch mean for equal shards)."""
        self.adopt()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
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


def all_reduce_scalar(value, op="mean", group=None):
    """Loss / PSNR logging across ranks."""
    t = value.detach().clone().reshape(1) if torch.is_tensor(value) else torch.tensor([float(value)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if op == "mean":
            t /= dist.get_world_size(group)
    return t
