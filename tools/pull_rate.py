"""Cost of the pull-image half of the API: beamformer_get_last_frames (lib/ogl_beamformer_lib.c:656-702) into ordinary
(pageable) caller memory, per BASELINE config, next to the frame's compute time; and a push + pull loop as a C /
MATLAB client that looks at every frame would run it.  Run from the repository root on a GPU box:
PYTHONPATH=. python tools/pull_rate.py [--devices 0,0]"""
import argparse
import ctypes as C
import json
import time

import numpy as np
import torch

from ogl_beamforming_amd import configs, lib, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="")
ap.add_argument("--configs", default="1,2,3,4")
ap.add_argument("--json", default="")
args = ap.parse_args()

torch.cuda.set_device(0)
L = lib.library()
if args.devices:
    ids = [int(v) for v in args.devices.split(",")]
    assert L.beamformer_hip_set_devices((C.c_int32 * len(ids))(*ids), len(ids)), lib.last_error()
L.beamformer_set_global_timeout(0xFFFFFFFF)
rows = []
for cfg in (int(v) for v in args.configs.split(",")):
    acq = configs.config(cfg, 1.0)
    bp = acq.bp
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(bp))
    rf = np.ascontiguousarray(acq.rf)
    dev = torch.from_numpy(rf.view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()

    def push():
        assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0), lib.last_error()

    push()
    L.beamformer_hip_synchronize()
    info = P.HipFrameInfo()
    assert L.beamformer_hip_get_last_frame_info(C.byref(info))
    X, Y, Z = (max(1, v) for v in bp.output_points[:3])
    nbytes = (X * Y * Z * (8 if info.data_kind == int(P.DataKind.Float32Complex) else 4) + 63) // 64 * 64
    out = np.empty(nbytes // 4, dtype=np.float32)
    out.fill(0)                                            # touch the pages: a client reuses its buffer
    n = 200 if cfg == 1 else 3 if cfg >= 4 else 20
    # compute alone
    t0 = time.perf_counter()
    for _ in range(n):
        push()
    L.beamformer_hip_synchronize()
    compute = (time.perf_counter() - t0) / n
    # pull alone (frame already finished)
    pulls = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1), lib.last_error()
        pulls.append(time.perf_counter() - t0)
    pull = min(pulls)
    # push + pull every frame
    t0 = time.perf_counter()
    for _ in range(n):
        push()
        assert L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1), lib.last_error()
    both = (time.perf_counter() - t0) / n
    row = {"config": cfg, "frame_bytes": nbytes, "compute_ms": compute * 1e3, "pull_ms": pull * 1e3, "pull_GBps": nbytes / pull / 1e9,
           "push_and_pull_ms": both * 1e3, "pulls_ms": [p * 1e3 for p in pulls]}
    rows.append(row)
    print(json.dumps(row), flush=True)
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
