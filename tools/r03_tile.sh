#!/bin/bash
# das_tile.hip: parity of the block-staged kernel, then BASELINE config 2 with it (automatic) and without (flag 0x200)
mkdir -p gpurun_out/r03
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block_staging" > gpurun_out/r03/tile_pytest.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/r03/tile_pytest.log
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],3), "ms", d["config"]["das_path"], "DAS", round(d["config"]["stage_ms"]["DAS"],3), d["config"].get("das_plan",{}).get("tile_window_samples"))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for path in 0 512; do
  f=gpurun_out/r03/tile_c2_${path}.json
  timeout -k 10 200 python bench.py --config 2 --das-path $path --steps 20 --warmup 5 --no-cpu-baseline > $f 2> ${f%.json}.err || { echo "bench failed"; tail -5 ${f%.json}.err; exit 1; }
  line $f "config 2 das-path $path"
done
