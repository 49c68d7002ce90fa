"""Worker of tests/test_gpu_parallel.py: one data-parallel rank running REAL training steps (HIP kernels) on its
shard of a global batch; both ranks share the one GPU of the box, gradients travel through a gloo process group
(staged through the host) -- the same GradientBucket / train_step code path bench.py --gpus N runs over RCCL."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402


def main():
    rank, world, out_path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cdlnet_video_amd as cva
    from cdlnet_video_amd.parallel import GradientBucket, broadcast_parameters, shard_batch
    torch.manual_seed(100 + rank)                                  # different weights until the broadcast
    net = cva.CDLNet(K=3, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=False)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_ not in ("t", "g"):
                p.mul_(0.05)
            elif n_ == "t":
                p.fill_(5e-3)
    net = net.cuda()
    broadcast_parameters(net, src=0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    bucket = GradientBucket(net.parameters()).attach()             # sync() is queued by the reverse sweep itself
    x_all = cva.utils.synthetic_clip((4, 1, 40, 72), seed=7)       # the GLOBAL batch, same on every rank
    noise = torch.randn(x_all.shape, generator=torch.Generator().manual_seed(8)) * 25 / 255
    x, nz = shard_batch(x_all, rank, world).cuda(), shard_batch(noise, rank, world).cuda()
    copies, losses = [], []
    first_grads = None
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        xhat, _ = net(x + nz, 25.0)
        loss = torch.mean((x - xhat) ** 2)
        loss.backward()                                            # ... + the exchange: mean over ranks
        assert bucket.syncs == step + 1, bucket.syncs              # exactly one per backward pass, no trainer call
        copies.append(bucket.copies)
        if step == 0:
            first_grads = {n_: p.grad.detach().cpu().clone() for n_, p in net.named_parameters() if p.grad is not None}
        torch.nn.utils.clip_grad_norm_(net.parameters(), 5e-2)
        opt.step()
        net.project()
        losses.append(float(loss))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"same": all(torch.equal(gathered[0], g_) for g_ in gathered), "copies": copies,
                    "losses": losses, "first_grads": first_grads}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
