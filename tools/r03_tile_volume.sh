#!/bin/bash
# config 4's geometry with cubic interpolation at half size, no LDS-table staging (das path 2): the block-staged factored kernel
# (automatic there) against the per-voxel one (flag 0x200)
mkdir -p gpurun_out/r03
for path in 2 514; do
  f=gpurun_out/r03/tile_volume_${path}.json
  timeout -k 10 300 python bench.py --config 4 --scale 0.5 --interpolation cubic --das-path $path --steps 3 --warmup 1 --no-cpu-baseline > $f 2> ${f%.json}.err || { tail -3 ${f%.json}.err; exit 1; }
  python - $f $path <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("das-path", sys.argv[2], round(d["ms_per_step"],2), "ms", d["config"]["das_path"], d["config"]["das_plan"].get("tile_window_samples"))
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "transmit_counts" 2>&1 | tail -3
