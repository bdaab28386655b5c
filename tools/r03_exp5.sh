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
for k in tpw forces hercules; do
  for walk in plane depth; do
   for span in 0 1; do
    [ $k = hercules ] && [ $span = 1 ] && continue
    f=gpurun_out/r03/walk_${k}_${walk}_${span}.json
    if [ $span = 1 ]; then export BEAMFORMER_HIP_SPAN=1; else unset BEAMFORMER_HIP_SPAN; fi
    if [ $walk = plane ]; then export BEAMFORMER_HIP_TILE_WALK=plane; else unset BEAMFORMER_HIP_TILE_WALK; fi
    timeout -k 10 120 python bench.py --config harness:$k --steps 10 --warmup 2 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k walk $walk span $span"
   done
  done
done
export BEAMFORMER_HIP_SPAN=1
export BEAMFORMER_HIP_TILE_WALK=plane
timeout -k 10 300 python3 tools/pmc_das.py --config harness:tpw --das-path 0 --groups 0,1,3,7,9 --timeout 90 --out gpurun_out/r03/pmc_tpw_span3_plane > gpurun_out/r03/pmc_tpw_span3_plane.log 2>&1
unset BEAMFORMER_HIP_TILE_WALK
timeout -k 10 300 python3 tools/pmc_das.py --config harness:tpw --das-path 0 --groups 7,9 --timeout 90 --out gpurun_out/r03/pmc_tpw_span3_depth > gpurun_out/r03/pmc_tpw_span3_depth.log 2>&1
