"""Multi-GPU partition of the hot path: one process per GPU, iteration striping, ONE
all-reduce (RCCL over xGMI when the backend is "nccl") of the accumulation buffer.

The reference has no multi-GPU path: samples are independent given the
(pixel, iteration)-keyed Philox stream, so rank r of G renders iterations r, r+G, ... for all
pixels into its own running mean; the G means are combined as sum(n_r * mean_r) / sum(n_r).
Payload: W*H*3 fp32 (24.9 MB at 1080p).
"""
import torch
import torch.distributed as dist


def stripe(rank, world, first_iteration=0):
    """(iteration of the first launch, iter_stride, blue-noise pre-advance steps) for `rank`."""
    return first_iteration + rank, world, rank


def combine_means(accum, n_local, group=None):
    """In place: accum <- global mean.  accum: [n_pixels, 3] fp32 running mean over this rank's
    n_local iterations.  One all-reduce for the weighted sums; the counts are known a priori when
    every rank renders the same number of iterations, otherwise they ride along as one extra row."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum
    world = dist.get_world_size(group)
    flat = accum.reshape(-1)
    buf = torch.empty(flat.numel() + 1, dtype=torch.float32, device=accum.device)
    torch.mul(flat, float(n_local), out=buf[:-1])
    buf[-1] = float(n_local)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    torch.div(buf[:-1], buf[-1], out=flat)
    del world
    return accum
