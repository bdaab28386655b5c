"""The staged kernel's wave-uniform transmit tables (das_staged.hip, UNI: 64 x 16 tiles, global table + scalar loads) against
the same kernel with the tables in LDS (the shape the host would otherwise pick), on config 4's acquisition with the lateral
extent of the grid along x scaled down until 64-voxel tiles fit a 32-sample window.  BEAMFORMER_HIP_STAGED_NOUNIFORM is read per
frame, so both run in one process.  Run from the repository root on a GPU box:
PYTHONPATH=. python tools/staged_uniform.py [--json out.json]"""
import argparse
import ctypes as C
import json
import os

import numpy as np
import torch

from ogl_beamforming_amd import configs, lib, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--scales", default="0.5,0.35")
ap.add_argument("--planes", type=int, default=32)
ap.add_argument("--json", default="")
args = ap.parse_args()

torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
rows = []
for scale in (float(v) for v in args.scales.split(",")):
    Cn, A, S = 256, 75, 2048
    half = (Cn - 1) / 2 * 0.15e-3
    path = S / 25e6 * configs.SPEED_OF_SOUND
    z0, z1 = 0.12 * path, 0.30 * path
    acq = configs.rca(f"fine{scale}", Cn, A, S, (512, 512, 512), (-half * scale, -half, z0), (half * scale, half, z1), seed=4, cw=True,
                      pitch=0.15e-3, orientation=0x12, f_number=0.5, angles=np.linspace(-18.5, 18.5, A))
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert L.beamformer_hip_set_output_shard(0, 256 - args.planes // 2, args.planes), lib.last_error()
    dev = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).cuda()
    row = {"x_extent_scale": scale, "planes": args.planes, "transmits": A}
    frames = {}
    for name, env, shape in (("tables_in_lds", "1", None), ("uniform", None, None)):
        lib.set_hook("STAGED_NOUNIFORM", env or None)
        lib.set_hook("STAGED_SHAPE", shape or None)
        t = P.HipFrameTimings()
        best = 1e9
        for _ in range(4):
            assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0), lib.last_error()
            assert L.beamformer_hip_get_last_frame_timings(C.byref(t)), lib.last_error()
            kinds = [int(t.stage_kind[k]) for k in range(int(t.stage_count))]
            best = min(best, float(t.stage_ms[kinds.index(int(P.ShaderKind.DAS))]))
        row[name + "_ms"] = best
        row[name + "_path"] = int(t.das_path)
    lib.set_hook("STAGED_SHAPE", None)
    row["uniform_over_tables_in_lds"] = row["uniform_ms"] / row["tables_in_lds_ms"]
    rows.append(row)
    print(json.dumps(row), flush=True)
assert L.beamformer_hip_set_output_shard(0, 0, 0)
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
