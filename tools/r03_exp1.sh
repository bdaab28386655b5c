#!/bin/bash
# round 3 experiment: tile shapes for the harness frame on the factored and general kernels + PMC of the three kernels
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
for k in tpw hercules forces; do
 for path in 0 1; do
  for shape in 8,0,0 7,1,0 6,2,0 5,3,0 4,4,0 3,5,0; do
    [ $k = hercules ] && [ $path = 0 ] && [ $shape != 8,0,0 ] && continue
    f=gpurun_out/r03/shape_${k}_p${path}_${shape//,/}.json
    BEAMFORMER_HIP_TILE_SHAPE=$shape timeout -k 10 120 python bench.py --config harness:$k --steps 10 --warmup 2 --no-cpu-baseline --das-path $path > $f 2> ${f%.json}.err
    line $f "$k path $path shape $shape"
  done
 done
done
for spec in "harness:tpw 0" "harness:hercules 0" "harness:hercules 1"; do
  set -- $spec
  timeout -k 10 400 python3 tools/pmc_das.py --config $1 --das-path $2 --groups 0,1,5,7,8,9 --timeout 90 --out gpurun_out/r03/pmc_${1#harness:}_p$2 > gpurun_out/r03/pmc_${1#harness:}_p$2.log 2>&1
  echo "pmc $1 $2 done"
done
