import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from ogl_beamforming_amd import lib
L = lib.library()
L.beamformer_hip_set_das_path(0x14)
for name in sys.argv[1:]:
    acq = cases.make(name)
    os.environ.pop("BEAMFORMER_HIP_SPAN", None)
    a = np.asarray(lib.beamform(acq.bp, acq.rf, acq.filters)).copy()
    os.environ["BEAMFORMER_HIP_SPAN"] = "1"
    b = np.asarray(lib.beamform(acq.bp, acq.rf, acq.filters)).copy()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    d = np.nan_to_num(np.abs(a - b))
    a = np.nan_to_num(a)
    print(name, a.shape, "C", acq.bp.channel_count, "A", acq.bp.acquisition_count, "max", d.max(), "of", np.abs(a).max(), "differing voxels", int((d > 0).sum()), "of", d.size)
    if d.max() > 0:
        idx = np.argwhere(d > 0)
        print("  first", idx[:8].tolist(), "last", idx[-4:].tolist())
        print("  per-row counts (axis -2):", (d > 0).sum(axis=-1).reshape(-1)[:64].tolist())
