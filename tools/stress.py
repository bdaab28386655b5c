"""Soak test on a GPU box: 30 000 small frames through both push entry points, alternating four
parameter blocks (2-D single plane wave, 2-D compounding on the factored kernel, a row-column volume on the LDS-staged kernel, a cubic 2-D frame on the
block-staged factored kernel with chunks of both kinds) and occasionally re-pushing parameters (replans), with periodic bit-exact
checks of the exported frame and a watch on host RSS and device memory.
    PYTHONPATH=. python tools/stress.py [frames [device,device,...]]"""
import ctypes as C, os, sys, time
import numpy as np, torch
from ogl_beamforming_amd import configs, lib, params as P

def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6

torch.cuda.set_device(0)
L = lib.library()
if len(sys.argv) > 2:            # stress.py N 0,0,0 : the same soak through the multi-device mode
    ids = [int(v) for v in sys.argv[2].split(",")]
    assert L.beamformer_hip_set_devices((C.c_int32 * len(ids))(*ids), len(ids))
L.beamformer_set_global_timeout(0xFFFFFFFF)
assert L.beamformer_reserve_parameter_blocks(4)
L.beamformer_hip_set_das_path(0x110)          # das_tile.hip wherever it can run, no channel split (the fourth block; the others do not qualify)
acqs = [configs.config(1, 0.5), configs.config(2, 0.0625),
        configs.rca("staged", 32, 13, 512, (40, 36, 3), (-3e-3, -3e-3, 6e-3), (3e-3, 3e-3, 18e-3), seed=46, orientation=0x12, cw=True,
                    f_number=0.6, angles=np.linspace(-12, 12, 13)),
        configs.rca("tile", 96, 5, 512, (128, 32, 1), (-10.0e-3, 0, 3.0e-3), (10.0e-3, 0, 8.0e-3), seed=74, interp=P.InterpolationMode.Cubic, orientation=0x22,
                    f_number=0.3, pitch=0.2e-3, angles=np.linspace(-10, 10, 5))]
paths = []
golden = []
for slot, acq in enumerate(acqs):
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, slot)
    assert L.beamformer_push_simple_parameters_at(C.byref(acq.bp), slot)
    rf = np.ascontiguousarray(acq.rf)
    assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, slot)
    golden.append(lib.get_last_frame(acq.bp).copy())
    t = P.HipFrameTimings(); assert L.beamformer_hip_get_last_frame_timings(C.byref(t)); paths.append(int(t.das_path))
assert paths[2] == 2 and paths[3] == 5, paths
dev = [torch.from_numpy(np.ascontiguousarray(a.rf).view(np.uint8).reshape(-1)).cuda() for a in acqs]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]; rss0 = rss_mb(); t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
for i in range(N):
    slot = (i // 7) % 4
    acq = acqs[slot]
    if i % 1013 == 0:
        assert L.beamformer_push_simple_parameters_at(C.byref(acq.bp), slot)          # replan
    if i & 1:
        ok = L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev[slot].data_ptr()), dev[slot].numel(), 0, slot)
    else:
        rf = acq.rf
        ok = L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, slot)
    assert ok, lib.last_error()
    if i % 997 == 0:
        got = lib.get_last_frame(acq.bp)
        assert np.array_equal(got.view(np.uint32), golden[slot].view(np.uint32)), i
        stats = P.ComputeStatsTable()
        assert L.beamformer_compute_timings(C.byref(stats), -1) and stats.shader_count >= 2
L.beamformer_hip_synchronize()
el = time.time() - t0
free1 = torch.cuda.mem_get_info()[0]
print(f"{N} frames in {el:.2f} s ({N / el:.0f} frames/s); device memory delta {(free0 - free1) / 1e6:.1f} MB; host RSS {rss0:.0f} -> {rss_mb():.0f} MB")
assert abs(free0 - free1) < 64e6 and rss_mb() - rss0 < 200
L.beamformer_hip_set_das_path(0)
print("stress ok; das paths of the four blocks:", paths)
