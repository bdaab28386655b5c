import sys, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_random import draw_tile
from ogl_beamforming_amd import lib as bflib, params as P
from oracle import binding as oracle
import ctypes as C
L = bflib.library()
for seed in [int(v) for v in sys.argv[1:]]:
    acq = draw_tile(seed)
    bp = acq.bp
    print("seed", seed, "kind", bp.acquisition_kind, "points", tuple(bp.output_points)[:3], "C", bp.channel_count, "A", bp.acquisition_count, "S", bp.sample_count,
          "f#", bp.f_number, "cw", bp.coherency_weighting if hasattr(bp, "coherency_weighting") else None)
    print(np.array(bp.das_voxel_transform).reshape(4, 4))
    ref, pairs = oracle.beamform(acq.bp, acq.rf, acq.filters)
    ok = ~np.isnan(ref)
    scale = np.abs(ref[ok]).max()
    for mode in (0x110, 0x210, 0x11):
        L.beamformer_hip_set_das_path(mode)
        out = np.asarray(bflib.beamform(acq.bp, acq.rf, acq.filters)).copy()
        t = P.HipFrameTimings(); L.beamformer_hip_get_last_frame_timings(C.byref(t))
        err = np.abs(out - ref); err[~ok] = 0
        w = np.unravel_index(np.argmax(err), err.shape)
        print(hex(mode), "path", t.das_path, "staged", t.tile_staged_chunks, "gather", t.tile_gather_chunks, "err", float(err.max() / scale), "at", w, "ref", ref[w], "gpu", out[w],
              "nan equal", bool(np.array_equal(np.isnan(out), np.isnan(ref))), "voxels over 2e-3:", int((err > 2e-3 * scale).sum()))
    L.beamformer_hip_set_das_path(0)
