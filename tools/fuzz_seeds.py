"""Named seeds of tests/test_gpu_random.py's generators under chosen das-path modes, against the oracle with the suite's comparison:
PYTHONPATH=. python tools/fuzz_seeds.py separable 96,107,112 --modes 0,1,2,3,4   (modes: beamformer_hip_set_das_path)"""
import argparse
import sys

import numpy as np

from ogl_beamforming_amd import lib as bflib
from oracle import binding as oracle
from tests import test_gpu_random as R
from tests.test_gpu_parity import compare, last_timings, reference

ap = argparse.ArgumentParser()
ap.add_argument("generator", choices=["general", "separable", "tile"])
ap.add_argument("seeds")
ap.add_argument("--modes", default="0")
args = ap.parse_args()
gen = {"general": R.draw, "separable": R.draw_separable, "tile": R.draw_tile}[args.generator]
if "-" in args.seeds:
    a, b = args.seeds.split("-")
    seeds = list(range(int(a), int(b)))
else:
    seeds = [int(v) for v in args.seeds.split(",")]
modes = [int(v, 0) for v in args.modes.split(",")]
L = bflib.library()
ran = failed = 0
for seed in seeds:
    try:
        acq = gen(seed)
        ref, pairs, flags = reference(oracle, acq)
    except Exception as e:
        print(args.generator, seed, "not drawn:", str(e)[:80])
        continue
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        continue
    for mode in modes:
        L.beamformer_hip_set_das_path(mode)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            path = int(last_timings(bflib).das_path)
            ran += 1
            err = compare(gpu, ref, acq, flags)
            print(args.generator, seed, "mode", hex(mode), "path", path, "ok", f"{err:.2e}")
        except AssertionError as e:
            failed += 1
            print(args.generator, seed, "mode", hex(mode), "path", path, "FAIL:", str(e)[:120])
        finally:
            L.beamformer_hip_set_das_path(0)
print(f"{ran} runs, {failed} failures")
sys.exit(1 if failed else 0)
