"""Reduce the wave stamps of the timeline variant of das_factored.hip (tools/timeline_probe.sh) on the reference harness's frames:
resident waves over time (whole chip and per XCD), when each XCD runs dry, and how long a wave lives by depth.
OGL_BEAMFORMER_LIB=build/variants/libogl_timeline.so PYTHONPATH=. python tools/timeline_probe.py --json out.json"""
import argparse
import ctypes as C
import json

import numpy as np

from ogl_beamforming_amd import configs, lib as bflib
from tests.test_gpu_parity import last_timings

ap = argparse.ArgumentParser()
ap.add_argument("--json", default="")
ap.add_argument("--kinds", default="tpw,forces")
args = ap.parse_args()
L = bflib.library()
L.bf_debug_timeline.argtypes = [C.c_void_p, C.c_uint]
L.bf_debug_timeline.restype = C.c_int
TICK_NS = 10.0                                     # s_memrealtime: 100 MHz
out = {"what": "das_factored_kernel on the reference harness's 512 x 1024 view plane: per-wave s_memrealtime stamps (variant build, tools/timeline_probe.sh)",
       "tick_ns": TICK_NS, "frames": {}}
for kind in args.kinds.split(","):
    acq = configs.harness(kind)
    for _ in range(3):
        bflib.beamform(acq.bp, acq.rf, acq.filters)
    t = last_timings(bflib)
    waves = int(np.prod([max(1, v) for v in acq.bp.output_points[:3]])) // 64
    buf = np.zeros(4 * 65536, dtype=np.uint64)
    assert L.bf_debug_timeline(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(-1, 4)[:waves]
    t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
    ok = t1 > 0
    t0, t1, st = t0[ok], t1[ok], st[ok]
    xcc = (st[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = st[:, 2].astype(np.int64) & 0xFFFFFFFF
    cu, se, simd = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 4) & 0x3          # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    by = st[:, 3].astype(np.int64) & 0xFFFFFFFF
    start, end = t0.min(), t1.max()
    span = end - start
    grid = np.linspace(start, end, 201)
    def resident(mask):
        return [int(((t0[mask] <= g) & (t1[mask] > g)).sum()) for g in grid]
    whole = resident(np.ones(len(t0), bool))
    per_xcd_end = {int(k): float((t1[xcc == k].max() - start) * TICK_NS * 1e-6) for k in np.unique(xcc)}
    per_xcd_waves = {int(k): int((xcc == k).sum()) for k in np.unique(xcc)}
    dur = (t1 - t0) * TICK_NS * 1e-6
    rows = np.unique(by)
    by_depth = [[int(r), float(dur[by == r].mean())] for r in rows[:: max(1, len(rows) // 32)]]
    wave_ms = float(((t1 - t0).sum()) * TICK_NS * 1e-6)
    frame = {
        "stage_ms_event_timers": [float(t.stage_ms[i]) for i in range(int(t.stage_count))], "das_path": int(t.das_path),
        "waves_stamped": int(len(t0)), "kernel_span_ms": float(span * TICK_NS * 1e-6),
        "mean_resident_waves": wave_ms / (span * TICK_NS * 1e-6), "mean_resident_waves_per_cu": wave_ms / (span * TICK_NS * 1e-6) / 256.0,
        "resident_waves_at_200_points": whole,
        "time_with_fewer_than_half_of_peak_resident_frac": float(np.mean(np.array(whole) < 0.5 * max(whole))),
        "xcd_runs_dry_at_ms": per_xcd_end, "xcd_waves": per_xcd_waves,
        "wave_duration_ms": {"min": float(dur.min()), "median": float(np.median(dur)), "max": float(dur.max())},
        "mean_wave_duration_ms_by_tile_row": by_depth,
        "start_of_first_and_second_generation": {"waves_started_in_first_5_percent": int((t0 < start + 0.05 * span).sum())},
        "cus_seen": int(len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))),
    }
    out["frames"][f"harness:{kind}"] = frame
    print(kind, json.dumps({k: v for k, v in frame.items() if k not in ("resident_waves_at_200_points", "mean_wave_duration_ms_by_tile_row")}))
    print(" resident:", whole[::5])
    print(" duration by row:", [(r, round(d, 2)) for r, d in by_depth])
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
