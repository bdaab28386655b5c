"""Frame rate of the two-process arrangement: the reference's own client library (compiled from
/root/reference by `make -C oracle ref`) -> shared memory v33 -> ogl_beamformer_server -> MI355X.
    PYTHONPATH=. python tools/shm_rate.py"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
from ogl_beamforming_amd import configs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
server = subprocess.Popen([os.path.join(ROOT, "ogl_beamforming_amd", "ogl_beamformer_server")], stdout=subprocess.DEVNULL)
time.sleep(1.0)
try:
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libogl_beamformer_lib_ref.so"))
    ref.beamformer_get_last_error_string.restype = C.c_char_p
    ref.beamformer_push_data_with_compute.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ref.beamformer_get_last_frames.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    for cfg, n in ((1, 2000), (2, 100)):
        acq = configs.config(cfg, 1.0)
        for slot, fp in enumerate(acq.filters):
            assert ref.beamformer_create_filter(C.byref(fp), slot, 0), ref.beamformer_get_last_error_string()
        assert ref.beamformer_push_simple_parameters(C.byref(acq.bp)), ref.beamformer_get_last_error_string()
        ref.beamformer_set_global_timeout(20000)
        rf = np.ascontiguousarray(acq.rf)
        push = lambda: ref.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0)
        for _ in range(5):
            assert push(), ref.beamformer_get_last_error_string()
        out = np.zeros(acq.voxels * 2 + 16, np.float32)
        assert ref.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1)
        t0 = time.perf_counter()
        for _ in range(n):
            assert push(), ref.beamformer_get_last_error_string()
        assert ref.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1)
        el = time.perf_counter() - t0
        print(f"config {cfg}: {n} frames through the reference client + shm server in {el:.3f} s = {el / n * 1e6:.1f} us/frame", flush=True)
finally:
    server.terminate()
    server.wait(timeout=10)
