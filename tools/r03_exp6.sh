#!/bin/bash
mkdir -p gpurun_out/r03
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],3), "ms", d["config"]["das_path"], "DAS", round(d["config"]["stage_ms"]["DAS"],3), "pairs", d["roofline"]["pairs_per_launch"])
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
export BEAMFORMER_HIP_TILE_WALK=plane
for ch in 16 32 64 128 256; do
  for span in 0 1; do
    f=gpurun_out/r03/chan_${ch}_${span}.json
    if [ $span = 1 ]; then export BEAMFORMER_HIP_SPAN=1; else unset BEAMFORMER_HIP_SPAN; fi
    HARNESS_CHANNELS=$ch timeout -k 10 120 python bench.py --config harness:tpw --steps 20 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "tpw channels $ch span $span"
  done
done
