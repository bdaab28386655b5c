import ctypes as C, sys, numpy as np
from ogl_beamforming_amd import lib as bflib, params as P
from oracle import binding as oracle
from tests import test_gpu_random as R, cases
L = bflib.library()
gen = R.draw_separable if sys.argv[1] == "separable" else R.draw
for seed in [int(v) for v in sys.argv[2:]]:
    acq = gen(seed); bp = acq.bp
    print("seed", seed, "kind", bp.acquisition_kind, "points", tuple(bp.output_points)[:3], "C", bp.channel_count, "A", bp.acquisition_count, "S", bp.sample_count, "interp", bp.interpolation_mode, "cw", bp.coherency_weighting, "f#", round(bp.f_number, 3))
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    ok = ~np.isnan(ref); scale = np.abs(ref[ok]).max(); tol = cases.tolerance(acq)
    res = {}
    for mode in (0, 2, 4, 1):
        L.beamformer_hip_set_das_path(mode)
        out = np.asarray(bflib.beamform(acq.bp, acq.rf, acq.filters)).copy()
        t = P.HipFrameTimings(); L.beamformer_hip_get_last_frame_timings(C.byref(t))
        res[mode] = out
        err = np.abs(out - ref); err[~ok] = 0
        w = np.unravel_index(np.argmax(err), err.shape)
        print("  mode", mode, "path", t.das_path, "err", float(err.max() / scale), "at", tuple(int(v) for v in w), "voxels over tol:", int((err > tol * scale).sum()), "nan equal", bool(np.array_equal(np.isnan(out), np.isnan(ref))))
    L.beamformer_hip_set_das_path(0)
