"""Multi-GPU partition of the hot path: one process per GPU, iteration striping, ONE
all-reduce (RCCL over xGMI) of the accumulation buffer.

The reference has no multi-GPU path: samples are independent given the
(pixel, iteration)-keyed Philox stream, so rank r of G renders iterations r, r+G, ... for all
pixels into its own running mean; the G means are combined as sum(n_r * mean_r) / sum(n_r).
Payload: W*H*3 fp32 (24.9 MB at 1080p).

The collective itself lives BELOW the C ABI (`vpt_allreduce_accum`, csrc/vpt_host.hip: scale kernel ->
ONE ncclAllReduce of W*H*3 + 1 floats (the count rides in the last one) -> divide kernel, all on one HIP stream, no host synchronisation), so a C++
host gets it without Python (tools/vpt_cli.cpp --ranks).  This module is the Python host's use of it:
torch.distributed only carries the 128-byte RCCL id from rank 0 to the others (and the barriers of
bench.py).  `combine_means` falls back to a torch.distributed all-reduce when the context has no
communicator -- the gloo path the CPU tests run (tests/test_dist_gloo.py).
"""
import torch
import torch.distributed as dist


def stripe(rank, world, first_iteration=0):
    """(iteration of the first launch, iter_stride, blue-noise pre-advance steps) for `rank`."""
    return first_iteration + rank, world, rank


def init_comm(ctx, group=None):
    """Create ctx's RCCL communicator over the ranks of the (already initialised) torch.distributed
    group: rank 0 makes the id, the store / a broadcast carries it.  Collective call."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init(world, rank, box[0])
    return ctx


def combine_means(accum, n_local, group=None, ctx=None, stream=None):
    """In place: accum <- global mean.  accum: [n_pixels, 3] fp32 running mean over this rank's
    n_local iterations.

    ctx with a communicator (init_comm): the C-ABI collective, enqueued on `stream` (a raw HIP stream
    handle; None = the context's own stream) -- asynchronous, ordered with the renders on that stream.
    Otherwise: one torch.distributed all-reduce of the weighted sums, the count riding along as one
    extra element (gloo on CPU tensors in the tests)."""
    if ctx is not None and getattr(ctx, "comm_nranks", 0) > 0:
        ctx.allreduce_accum(accum, n_local, stream)
        return accum
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum
    flat = accum.reshape(-1)
    buf = torch.empty(flat.numel() + 1, dtype=torch.float32, device=accum.device)
    torch.mul(flat, float(n_local), out=buf[:-1])
    buf[-1] = float(n_local)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    torch.div(buf[:-1], buf[-1], out=flat)
    return accum
