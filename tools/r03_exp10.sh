#!/bin/bash
mkdir -p gpurun_out/r03
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],3), "ms", d["config"]["das_path"], "DAS", round(d["config"]["stage_ms"]["DAS"],3))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for walk in plane depth; do
  for np in 0 1; do
    if [ $walk = plane ]; then export BEAMFORMER_HIP_TILE_WALK=plane; else unset BEAMFORMER_HIP_TILE_WALK; fi
    if [ $np = 1 ]; then export BEAMFORMER_HIP_HERCULES_NOPAIRS=1; else unset BEAMFORMER_HIP_HERCULES_NOPAIRS; fi
    f=gpurun_out/r03/herc_${walk}_${np}.json
    timeout -k 10 120 python bench.py --config harness:hercules --steps 10 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "hercules walk $walk nopairs $np"
  done
done
unset BEAMFORMER_HIP_TILE_WALK
export BEAMFORMER_HIP_HERCULES_NOPAIRS=1
timeout -k 10 300 python3 tools/pmc_das.py --config harness:hercules --das-path 0 --groups 0,1,3,7,9 --timeout 90 --out gpurun_out/r03/pmc_herc_nopairs > gpurun_out/r03/pmc_herc_nopairs.log 2>&1
