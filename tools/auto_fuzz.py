"""One-off fuzz beyond the fixed seeds of tests/test_gpu_random.py: seeds FIRST..LAST of its two generators (general draws on the automatic path;
separable draws on the automatic path and with every staged term range-checked), same comparison as the tests.
PYTHONPATH=. python tools/auto_fuzz.py 72 400"""
import sys
import traceback

import numpy as np

from ogl_beamforming_amd import lib as bflib
from oracle import binding as oracle
from tests import test_gpu_random as R
from tests.test_gpu_parity import compare, last_timings, reference

first, last = int(sys.argv[1]), int(sys.argv[2])
L = bflib.library()
ran = failed = 0
paths = {}
for gen, name in ((R.draw, "general"), (R.draw_separable, "separable")):
    for seed in range(first, last):
        try:
            acq = gen(seed)
        except Exception as e:                      # a draw the generator itself cannot build
            continue
        try:
            ref, pairs, flags = reference(oracle, acq)
        except Exception:
            continue
        ok = ~np.isnan(ref)
        if not ok.any() or np.max(np.abs(ref[ok])) == 0:
            continue
        L.beamformer_hip_set_das_path(0)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            path = int(last_timings(bflib).das_path)
            paths[path] = paths.get(path, 0) + 1
            ran += 1
            compare(gpu, ref, acq, flags)
            if path == 2:
                bflib.set_hook("STAGED_CHECKED", "1")
                try:
                    checked = bflib.beamform(acq.bp, acq.rf, acq.filters)
                    t = last_timings(bflib)
                    assert int(t.staged_window_violations) == 0, "window violation"
                    compare(checked, ref, acq, flags)
                finally:
                    bflib.set_hook("STAGED_CHECKED", None)
        except AssertionError as e:
            failed += 1
            print(name, "seed", seed, "path", path, "FAIL:", str(e)[:160])
        except Exception:
            failed += 1
            print(name, "seed", seed, "ERROR"); traceback.print_exc(limit=1)
print(f"{ran} draws, {failed} failures, DAS paths taken: {dict(sorted(paths.items()))}")
sys.exit(1 if failed else 0)
