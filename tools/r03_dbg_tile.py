import sys, numpy as np
sys.path.insert(0, '.')
from tests import cases
from ogl_beamforming_amd import lib as bflib, params as P
from oracle import binding as oracle
import ctypes as C
L = bflib.library()
for name in ("tile_near_field",):
    acq = cases.make(name)
    ref, pairs = oracle.beamform(acq.bp, acq.rf, acq.filters)
    L.beamformer_hip_set_das_path(0x114)
    out = np.asarray(bflib.beamform(acq.bp, acq.rf, acq.filters)).copy()
    t = P.HipFrameTimings(); L.beamformer_hip_get_last_frame_timings(C.byref(t))
    scale = np.abs(ref).max()
    print(name, "path", t.das_path, "staged", t.tile_staged_chunks, "gather", t.tile_gather_chunks, "err", float(np.abs(out - ref).max() / scale))
    L.beamformer_hip_set_das_path(0)
