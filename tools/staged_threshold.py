"""Where the LDS-staged DAS kernel (das_staged.hip) starts to beat the gather kernel (das_separable.hip): config 4's
geometry (256 channels, 512 x 512 voxels per plane, a slab of planes) with the transmit count swept.  The staged
kernel pays two block barriers and a window copy per channel whatever the transmit count; executor.cpp's
kStagedMinTransmits comes from this table.  Run from the repository root on a GPU box:
PYTHONPATH=. python tools/staged_threshold.py [--json out.json]"""
import argparse
import ctypes as C
import json

import numpy as np
import torch

from ogl_beamforming_amd import configs, lib, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--transmits", default="4,6,8,12,16,24,32,48,75,128")
ap.add_argument("--planes", type=int, default=16)
ap.add_argument("--json", default="")
ap.add_argument("--cubic", action="store_true", help="cubic interpolation of IQ samples: das_staged_cubic.hip against the factored kernel (what path 2 runs for cubic)")
ap.add_argument("--real", action="store_true", help="real samples (no Demodulate): das_staged_real.hip against the gather kernel")
args = ap.parse_args()

torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
rows = []
for A in (int(v) for v in args.transmits.split(",")):
    Cn, S = 256, 2048
    half = (Cn - 1) / 2 * 0.15e-3
    path = S / 25e6 * configs.SPEED_OF_SOUND
    z0, z1 = 0.12 * path, 0.30 * path
    acq = configs.rca(f"sweep{A}", Cn, A, S, (512, 512, 512), (-half, -half, z0), (half, half, z1), seed=4, cw=True, pitch=0.15e-3,
                      orientation=0x12, f_number=0.5, angles=np.linspace(-18.5, 18.5, A), demodulate=not args.real,
                      interp=P.InterpolationMode.Cubic if args.cubic else P.InterpolationMode.Linear)
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert L.beamformer_hip_set_output_shard(0, 256 - args.planes // 2, args.planes), lib.last_error()
    rf = np.ascontiguousarray(acq.rf)
    dev = torch.from_numpy(rf.view(np.uint8).reshape(-1)).cuda()
    row = {"transmits": A, "planes": args.planes, "samples": "real" if args.real else "complex", "interpolation": "cubic" if args.cubic else "linear"}
    for name, mode in (("gather", 2), ("staged", 3)):
        L.beamformer_hip_set_das_path(mode)
        t = P.HipFrameTimings()
        best = 1e9
        for _ in range(4):
            assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0), lib.last_error()
            assert L.beamformer_hip_get_last_frame_timings(C.byref(t)), lib.last_error()
            kinds = [int(t.stage_kind[k]) for k in range(int(t.stage_count))]
            best = min(best, float(t.stage_ms[kinds.index(int(P.ShaderKind.DAS))]))
        row[name + "_ms"] = best
        row[name + "_path"] = int(t.das_path)
    L.beamformer_hip_set_das_path(0)
    row["staged_over_gather"] = row["staged_ms"] / row["gather_ms"]
    rows.append(row)
    print(json.dumps(row), flush=True)
assert L.beamformer_hip_set_output_shard(0, 0, 0)
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
