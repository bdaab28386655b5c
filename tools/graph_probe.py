"""Does a captured hipGraph beat the library's direct launches?  For each requested configuration one
frame (device-resident RF, every stage the plan holds) is captured with torch.cuda.CUDAGraph on the
library's stream and replayed; the same number of direct pushes is timed beside it.  Prints one JSON
object: per configuration {direct_us, replay_us | capture_error}.

    PYTHONPATH=. python tools/graph_probe.py [--configs 1,2,5] [--planes5 8] > profiles/r02_graph_probe.json

BASELINE.json's configs[4] names a "hipGraph-captured frame"; this is the measurement behind the
decision to launch a frame's <= 6 kernels directly instead."""
import argparse
import ctypes as C
import json
import time

import numpy as np
import torch

from ogl_beamforming_amd import configs, lib, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1,2,5")
ap.add_argument("--planes5", type=int, default=8, help="config 5 is probed on a slab of this many planes (a whole frame takes seconds)")
args = ap.parse_args()

torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
assert L.beamformer_hip_set_stream(C.c_void_p(stream.cuda_stream))
out = {"method": "N direct beamformer_hip_push_device_data_with_compute calls against N replays of one captured call (torch.cuda.CUDAGraph on the library's stream); "
                 "wall time per frame, device-resident RF"}

for n in [int(v) for v in args.configs.split(",")]:
    acq = configs.config(n, 1.0)
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    Z = max(1, acq.bp.output_points[2])
    planes = min(Z, args.planes5) if n == 5 else 0
    assert L.beamformer_hip_set_output_shard(0, (Z - planes) // 2 if planes else 0, planes)
    dev = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()

    def push():
        return L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0)

    for _ in range(20):
        assert push()                      # past the sampled first frames of the plan
    torch.cuda.synchronize()
    N = 2000 if n == 1 else 200 if n == 2 else 10
    t0 = time.perf_counter()
    for _ in range(N):
        assert push()
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / N
    entry = {"workload": acq.name + (f" ({planes}-plane slab)" if planes else ""), "frames_timed": N, "direct_us": direct * 1e6}
    # small frames record timing events on one frame in eight: capture right after a sampled one so that the
    # captured frame records none (big frames always record; the capture then has to carry the records)
    for _ in range(8):
        assert push()
        torch.cuda.synchronize()
        info = P.HipFrameInfo()
        L.beamformer_hip_get_last_frame_info(C.byref(info))
        if info.frame_id % 8 == (20 + N) % 8:
            break
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=stream):
            ok = push()
        assert ok
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            g.replay()
        torch.cuda.synchronize()
        entry["replay_us"] = (time.perf_counter() - t0) / N * 1e6
    except Exception as e:                                  # a record, memset or allocation that cannot be captured
        entry["capture_error"] = f"{type(e).__name__}: {str(e)[:300]}"
        torch.cuda.synchronize()
    out[f"config{n}"] = entry
    del g
    L.beamformer_hip_set_output_shard(0, 0, 0)

print(json.dumps(out, indent=1))
