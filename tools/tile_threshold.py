"""Where the block-staged factored kernel (das_tile.hip, das path 5) beats the per-voxel factored kernel (das_factored.hip): BASELINE
config 2's acquisition (128 channels x 31 plane waves, cubic, IQ) beamformed onto N x N grids of the same extent -- the finer the
grid, the smaller the spread of a 64 x 16-voxel tile.  For each N: what the library picks on its own, then the block-staged kernel
forced (flags 0x100 | 0x10: no channel split either) and forbidden (flag 0x200), with the share of (block, chunk) pairs the block-staged kernel had to run through
its gather loop.  das_select.cpp's rule for path 5 (estimated spread <= 58 samples) comes from this table.
Run from the repository root on a GPU box:  PYTHONPATH=. python tools/tile_threshold.py [--json out.json]"""
import argparse
import ctypes as C
import json

import numpy as np
import torch

from ogl_beamforming_amd import configs, lib, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--points", default="256,384,448,480,512,640,768,896,1024,1536,2048")
ap.add_argument("--json", default="")
args = ap.parse_args()

torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
base = configs.config(2)
rf = np.ascontiguousarray(base.rf)
dev = torch.from_numpy(rf.view(np.uint8).reshape(-1)).cuda()
rows = []
for N in (int(v) for v in args.points.split(",")):
    bp = base.bp
    bp.output_points[0], bp.output_points[1] = N, N
    for s, fp in enumerate(base.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    row = {"points": N}
    for name, mode in (("automatic", 0), ("block_staged", 0x110), ("per_voxel", 0x200)):
        L.beamformer_hip_set_das_path(mode)
        assert L.beamformer_push_simple_parameters(C.byref(bp))
        t = P.HipFrameTimings()
        best = 1e9
        for _ in range(5):
            assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0), lib.last_error()
            assert L.beamformer_hip_get_last_frame_timings(C.byref(t)), lib.last_error()
            kinds = [int(t.stage_kind[k]) for k in range(int(t.stage_count))]
            best = min(best, float(t.stage_ms[kinds.index(int(P.ShaderKind.DAS))]))
        row[name + "_ms"] = round(best, 4)
        row[name + "_path"] = int(t.das_path)
        if int(t.das_path) == 5:
            total = int(t.tile_staged_chunks) + int(t.tile_gather_chunks)
            row[name + "_gather_share"] = round(int(t.tile_gather_chunks) / max(1, total), 4)
            row[name + "_window"] = int(lib.describe_das(bp, base.filters)[4].tile_window_samples)
    L.beamformer_hip_set_das_path(0)
    row["block_staged_over_per_voxel"] = round(row["block_staged_ms"] / row["per_voxel_ms"], 3)
    rows.append(row)
    print(json.dumps(row), flush=True)
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
