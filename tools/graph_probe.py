"""Does a captured hipGraph beat the library's direct launches for a small frame?  Captures one
config-1 frame (device-resident RF: filter kernel + DAS kernel, no events on an unsampled frame)
with torch.cuda.CUDAGraph on the library's stream and replays it; compares with the same number
of direct pushes.      PYTHONPATH=. python tools/graph_probe.py"""
import ctypes as C, time
import numpy as np, torch
from ogl_beamforming_amd import configs, lib

torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
assert L.beamformer_hip_set_stream(C.c_void_p(stream.cuda_stream))
acq = configs.config(1, 1.0)
for s, fp in enumerate(acq.filters):
    assert L.beamformer_create_filter(C.byref(fp), s, 0)
assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
dev = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).cuda()
torch.cuda.synchronize()
push = lambda: L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0)
for _ in range(20):
    assert push()                      # past the sampled first frames of the plan
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    assert push()
torch.cuda.synchronize()
direct = (time.perf_counter() - t0) / N
# frames 21.. : make sure the captured one records no event (sampling period 8 -> capture right after a sampled one)
while True:
    assert push()
    t = lib.P.HipFrameTimings()
    torch.cuda.synchronize()
    info = lib.P.HipFrameInfo(); L.beamformer_hip_get_last_frame_info(C.byref(info))
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
    replay = (time.perf_counter() - t0) / N
    print(f"direct launches {direct * 1e6:.1f} us/frame   graph replay {replay * 1e6:.1f} us/frame")
except Exception as e:                                  # a record or memset that cannot be captured
    print(f"direct launches {direct * 1e6:.1f} us/frame   capture failed: {type(e).__name__}: {str(e)[:200]}")
