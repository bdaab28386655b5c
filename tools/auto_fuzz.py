"""Out-of-sample fuzz beyond the fixed seeds of tests/test_gpu_random.py: seeds FIRST..LAST of its three generators, every draw
against the oracle with the SUITE'S comparison (tests/test_gpu_parity.py compare(): the tolerance of the pipeline's arithmetic, and
for a voxel over it the double-precision truth -- never another kernel of the library):
  general    draws on the automatic path
  separable  draws on the automatic path, and where the LDS-staged kernel ran, again with every staged term range-checked
  tile       draws with the block-staged factored kernel asked for (flags 0x10 | 0x100)
  plane      view planes as the reference's harness beamforms them, without the small-frame channel split (0x10: the kernels a full-size plane gets)
  plane_hercules   the HERCULES-family draws of `plane` with the aligned-grid kernel asked for (6)
Writes a JSON summary (draws, failures, kernels taken, planes the row-end rule re-routed).
PYTHONPATH=. python tools/auto_fuzz.py 72 1200 [--json gpurun_out/r04/fuzz.json] [--generators general,separable,tile]"""
import argparse
import json
import sys
import traceback

import numpy as np

from ogl_beamforming_amd import lib as bflib
from oracle import binding as oracle
from tests import test_gpu_random as R
from tests.test_gpu_parity import compare, last_timings, reference

ap = argparse.ArgumentParser()
ap.add_argument("first", type=int)
ap.add_argument("last", type=int)
ap.add_argument("--json", default="")
ap.add_argument("--generators", default="general,separable,tile,plane,plane_hercules")
args = ap.parse_args()
L = bflib.library()
GENERATORS = {"general": (R.draw, 0), "separable": (R.draw_separable, 0), "tile": (R.draw_tile, 0x110), "plane": (R.draw_plane, 0x10), "plane_hercules": (R.draw_plane, 6)}
summary = {"seeds": [args.first, args.last], "comparison": "tests/test_gpu_parity.py compare(): tolerance of the pipeline, second bar against the oracle's double twin",
           "generators": {}, "failures": []}
total = failed_total = 0
for name in args.generators.split(","):
    gen, mode = GENERATORS[name]
    ran = failed = row_end_draws = second_bar = 0
    paths = {}
    for seed in range(args.first, args.last):
        try:
            acq = gen(seed)
            if name == "plane_hercules" and int(acq.bp.acquisition_kind) not in (int(R.K.HERCULES), int(R.K.UHERCULES)):
                continue
            ref, pairs, flags = reference(oracle, acq)
        except Exception:                               # a draw the generator or the oracle's planner cannot build
            continue
        ok = ~np.isnan(ref)
        if not ok.any() or np.max(np.abs(ref[ok])) == 0:
            continue
        path = -1
        try:
            L.beamformer_hip_set_das_path(mode)
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            t = last_timings(bflib)
            path = int(t.das_path)
            paths[path] = paths.get(path, 0) + 1
            row_end_draws += int(t.das_row_end_planes) > 0
            ran += 1
            compare(gpu, ref, acq, flags)
            if path == 2:
                bflib.set_hook("STAGED_CHECKED", "1")
                try:
                    checked = bflib.beamform(acq.bp, acq.rf, acq.filters)
                    assert int(last_timings(bflib).staged_window_violations) == 0, "window violation"
                    compare(checked, ref, acq, flags)
                finally:
                    bflib.set_hook("STAGED_CHECKED", None)
        except AssertionError as e:
            failed += 1
            what = str(e)[:200] or traceback.format_exc(limit=-1).strip().splitlines()[-2].strip()[:200]
            summary["failures"].append({"generator": name, "seed": seed, "path": path, "what": what})
            print(name, "seed", seed, "path", path, "FAIL:", what, flush=True)
        except Exception:
            failed += 1
            summary["failures"].append({"generator": name, "seed": seed, "path": path, "what": "exception"})
            print(name, "seed", seed, "ERROR", flush=True); traceback.print_exc(limit=1)
        finally:
            L.beamformer_hip_set_das_path(0)
        if ran and ran % 100 == 0:
            print(f"{name}: {ran} draws, {failed} failures so far", flush=True)
    summary["generators"][name] = {"draws": ran, "failures": failed, "das_paths_taken": {str(k): v for k, v in sorted(paths.items())},
                                   "draws_with_planes_rerouted_by_the_row_end_rule": row_end_draws}
    total += ran; failed_total += failed
    print(f"{name}: {ran} draws, {failed} failures, DAS paths taken: {dict(sorted(paths.items()))}, row-end re-routed draws: {row_end_draws}", flush=True)
summary["draws"] = total; summary["failed"] = failed_total
if args.json:
    json.dump(summary, open(args.json, "w"), indent=1)
print(f"{total} draws, {failed_total} failures")
sys.exit(1 if failed_total else 0)
