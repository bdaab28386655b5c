"""Voxel-grid sharding across the GPUs of a node.

The output grid is cut along z (the slowest-varying output index, das.glsl:132-136) into
contiguous slabs, one per rank; every rank needs the whole RF frame (one RCCL broadcast over
xGMI from the ingest rank) and nothing else: no reduction collective.  A slab computed with
beamformer_hip_set_output_shard() is bit-identical to the same planes of an unsharded frame,
so pulling an image is a concatenation in rank order.
"""


def slab(rank, world, planes):
    """(z_first, z_count) of `rank`; slabs differ by at most one plane and cover [0, planes)."""
    first = rank * planes // world
    return first, (rank + 1) * planes // world - first


def slabs(world, planes):
    return [slab(r, world, planes) for r in range(world)]


def broadcast_rf(rf_tensor, src=0):
    """The one collective of the path: the raw RF frame from the ingest rank to all ranks
    (backend nccl == RCCL on ROCm; gloo in CPU tests)."""
    import torch.distributed as dist
    dist.broadcast(rf_tensor, src=src)
    return rf_tensor


def gather_frame(local_slab, world, rank, dst=0):
    """Concatenate slabs along z on `dst` (pull-image of a sharded frame)."""
    import torch
    import torch.distributed as dist
    parts = [None] * world if rank == dst else None
    dist.gather_object(local_slab, parts, dst=dst)
    if rank != dst:
        return None
    return torch.cat([torch.as_tensor(p) for p in parts], dim=0)
