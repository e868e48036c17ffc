"""Multi-GPU: one process per GPU, each renders a disjoint samples-per-pixel range of the SAME
frame; the float accumulation buffers are then summed to rank 0 with one reduce (RCCL over xGMI
when the backend is "nccl"). Philox blocks are numbered by the global sample index, so the union
of the ranges is the frame a single GPU would render, up to float summation order.

The reference has no counterpart (one process, one shared vector, Path Tracer.cpp:292); the split
comes from BASELINE.json's north_star.
"""
import os


def weak_range(rank, spp_per_gpu):
    """Fixed work per GPU: rank r takes [r * spp, (r + 1) * spp)."""
    return rank * spp_per_gpu, (rank + 1) * spp_per_gpu


def split_range(total_spp, world, rank):
    """Fixed total work: [0, total_spp) cut into `world` contiguous, disjoint, near-equal ranges."""
    return (total_spp * rank) // world, (total_spp * (rank + 1)) // world


def init(backend=None, force=False):
    """Joins the process group described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*. Returns
    (rank, local_rank, world). A single process needs no group (force=True makes one anyway: a one-rank
    RCCL group exercises the same calls as N ranks, which a one-GPU box can otherwise not run)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend in (None, "nccl"):
            import torch
            backend = "nccl"
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())   # the caller has set the rank's device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def reduce_frame(accum, dst=0, force=False):
    """Sum the per-rank accumulation buffers into rank `dst` (one message of W*H*3 floats). force: also issue the
    collective in a one-rank group (same call, nothing to add)."""
    import torch.distributed as dist
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
