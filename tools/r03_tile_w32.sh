#!/bin/bash
mkdir -p gpurun_out/r03
for w in 64 32; do
  BEAMFORMER_HIP_TILE_WINDOW=$w timeout -k 10 200 python - <<PY
import ctypes as C, numpy as np, torch
from ogl_beamforming_amd import configs, lib, params as P
L=lib.library(); L.beamformer_set_global_timeout(0xFFFFFFFF)
a=configs.config(2)
for s,fp in enumerate(a.filters): assert L.beamformer_create_filter(C.byref(fp), s, 0)
assert L.beamformer_push_simple_parameters(C.byref(a.bp))
dev=torch.from_numpy(np.ascontiguousarray(a.rf).view(np.uint8).reshape(-1)).cuda()
t=P.HipFrameTimings(); best=1e9
for _ in range(12):
    assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(dev.data_ptr()), dev.numel(), 0, 0)
    assert L.beamformer_hip_get_last_frame_timings(C.byref(t))
    kinds=[int(t.stage_kind[k]) for k in range(int(t.stage_count))]
    best=min(best, float(t.stage_ms[kinds.index(int(P.ShaderKind.DAS))]))
print("window $w", "path", t.das_path, "DAS ms", round(best,3), "staged", t.tile_staged_chunks, "gather", t.tile_gather_chunks)
PY
done
