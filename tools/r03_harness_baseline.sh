#!/bin/bash
# round 3, first pass: parity of the view-plane cases + the harness frames on the automatic path and on the general kernel
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plane or harness" > gpurun_out/r03/parity_planes.log 2>&1
echo "parity rc $?" >> gpurun_out/r03/parity_planes.log
tail -5 gpurun_out/r03/parity_planes.log
for k in tpw tpw_swapped vls hercules forces; do
  for path in 0 1; do
    python bench.py --config harness:$k --steps 10 --warmup 2 --no-cpu-baseline --das-path $path > gpurun_out/r03/harness_${k}_path${path}.json 2> gpurun_out/r03/harness_${k}_path${path}.err || echo "bench $k $path failed"
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/harness_${k}_path${path}.json").read().strip().splitlines()[-1])
    print("$k path $path:", round(d["ms_per_step"],3), "ms", d["config"]["das_path"], d["config"]["stage_ms"], "pairs", d["roofline"]["pairs_per_launch"])
except Exception as e:
    print("$k $path: no line", e)
PY
  done
done
