"""Worker for tests/test_sharding.py: world_size-2 rehearsal of bench.py's multi-GPU path on
CPU (gloo).  The oracle stands in for the HIP kernels as the per-rank compute; what is under
test is the host logic: slab bounds, the RF broadcast, slab-local beamforming with whole-grid
voxel normalisation, gather order."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ogl_beamforming_amd import configs, sharding  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    acq = configs.config(4, 0.0625)               # every rank can build the parameter block
    rf = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1).copy())
    if rank != 0:
        rf.zero_()                                # only the ingest rank owns the data
    sharding.broadcast_rf(rf, src=0)
    rf_np = rf.numpy().view(acq.rf.dtype).reshape(acq.rf.shape)

    Z = max(1, acq.bp.output_points[2])
    z_first, z_count = sharding.slab(rank, world, Z)
    local, pairs = oracle.beamform(acq.bp, rf_np, acq.filters, z=(z_first, z_count), threads=2)
    frame = sharding.gather_frame(torch.from_numpy(np.ascontiguousarray(local).view(np.float32)), world, rank)
    total = torch.tensor([pairs], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        full, full_pairs = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=2)
        got = frame.numpy().view(np.complex64).reshape(full.shape)
        ok = np.array_equal(got, full, equal_nan=True) and int(total.item()) == full_pairs
        with open(sys.argv[1], "w") as f:
            f.write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
