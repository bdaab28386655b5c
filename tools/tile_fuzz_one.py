"""per-seed breakdown for tools/tile_fuzz.py: das_tile.hip (0x110), das_factored.hip (0x210) and the general kernel (0x11) against the oracle"""
import ctypes as C, sys, numpy as np
from ogl_beamforming_amd import lib as bflib, params as P
from oracle import binding as oracle
from tests.test_gpu_random import draw_tile
L = bflib.library()
for seed in [int(v) for v in sys.argv[1:]]:
    acq = draw_tile(seed); bp = acq.bp
    print("seed", seed, "kind", bp.acquisition_kind, "points", tuple(bp.output_points)[:3], "C", bp.channel_count, "A", bp.acquisition_count, "S", bp.sample_count, "f#", round(bp.f_number, 3), "cw", bp.coherency_weighting)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    ok = ~np.isnan(ref); scale = np.abs(ref[ok]).max()
    res = {}
    for mode in (0x110, 0x210, 0x11):
        L.beamformer_hip_set_das_path(mode)
        out = np.asarray(bflib.beamform(acq.bp, acq.rf, acq.filters)).copy()
        t = P.HipFrameTimings(); L.beamformer_hip_get_last_frame_timings(C.byref(t))
        res[mode] = out
        err = np.abs(out - ref); err[~ok] = 0
        w = np.unravel_index(np.argmax(err), err.shape)
        print("  ", hex(mode), "path", t.das_path, "staged", t.tile_staged_chunks, "gather", t.tile_gather_chunks, "err", float(err.max() / scale), "at", tuple(int(v) for v in w), "over 2e-3:", int((err > 2e-3 * scale).sum()))
    L.beamformer_hip_set_das_path(0)
    w = np.unravel_index(np.argmax(np.where(ok, np.abs(res[0x110] - ref), 0)), ref.shape)
    print("   at the block-staged kernel's worst voxel:", {hex(m): complex(res[m][w]) for m in res}, "oracle", complex(ref[w]), "frame max", float(scale))
