"""Frame time of every BASELINE config through both push entry points: device-resident RF
(beamformer_hip_push_device_data_with_compute) and host memory (beamformer_push_data_with_compute,
PCIe inclusive).  Run from the repository root on a GPU box:  PYTHONPATH=. python tools/hostpush.py"""
import ctypes as C, sys, time
import numpy as np, torch
from ogl_beamforming_amd import configs, lib, params as P
torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
for cfg, scale in ((1, 1.0), (2, 1.0), (3, 1.0), (4, 1.0)):
    acq = configs.config(cfg, scale)
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    rf = np.ascontiguousarray(acq.rf)
    dev = torch.from_numpy(rf.view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()
    N = 200 if cfg == 1 else (4 if cfg == 4 else 40)
    for mode in ("device", "host"):
        def push():
            if mode == "device":
                return L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0)
            return L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0)
        for _ in range(2 if cfg == 4 else 5): assert push()
        L.beamformer_hip_synchronize()
        t0 = time.perf_counter(); enq = 0.0
        for _ in range(N):
            a = time.perf_counter(); assert push(); enq += time.perf_counter() - a
        L.beamformer_hip_synchronize()
        el = time.perf_counter() - t0
        print(f"cfg{cfg} {mode:6s} rf {rf.nbytes/1e6:7.2f} MB  frame {el/N*1e6:9.1f} us  enqueue {enq/N*1e6:8.1f} us/frame", flush=True)
