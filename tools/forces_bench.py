"""FORCES at scale: 128 channels x 128 transmits -> 512 x 1024, {Demodulate, Decode, DAS}, on the automatic
path (factored kernel) and on the general kernel; prints frame times and their agreement.
    PYTHONPATH=. python tools/forces_bench.py"""
import ctypes as C, time, sys
import numpy as np, torch
sys.path.insert(0, '.')
from ogl_beamforming_amd import configs as cfg, lib, params as P
torch.cuda.set_device(0)
L = lib.library()
L.beamformer_set_global_timeout(0xFFFFFFFF)
S_, I = P.ShaderKind, P.InterpolationMode
path = 0.40 * 4096 / 25e6 * 1540.0
for interp in (I.Linear, I.Cubic):
    acq = cfg.forces("forces_big", 128, 128, 4096, (512, 1, 1024), (-19e-3, 0, 0.15 * path), (19e-3, 0, 0.40 * path), seed=9,
                     stages=(S_.Demodulate, S_.Decode, S_.DAS), interp=interp, f_number=0.5)
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    dev = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()
    frames = {}
    for mode in (0, 1):
        L.beamformer_hip_set_das_path(mode)
        push = lambda: L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0)
        for _ in range(2): assert push(), lib.last_error()
        L.beamformer_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(5): assert push()
        L.beamformer_hip_synchronize()
        el = (time.perf_counter() - t0) / 5
        t = P.HipFrameTimings(); L.beamformer_hip_get_last_frame_timings(C.byref(t))
        frames[mode] = lib.get_last_frame(acq.bp)
        print(f"FORCES 128x128 -> 512x1024 {I(interp).name}: mode {mode} das_path {t.das_path} frame {el*1e3:.2f} ms", flush=True)
    L.beamformer_hip_set_das_path(0)
    a, b = frames[0], frames[1]
    print("   factored vs general max rel diff", float(np.abs(a - b).max() / np.abs(b).max()))
