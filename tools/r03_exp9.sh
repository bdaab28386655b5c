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
export BEAMFORMER_HIP_TILE_WALK=plane
export BEAMFORMER_HIP_SPAN=1
for v in NO_DMA NO_CONSUME; do
  export OGL_BEAMFORMER_LIB=$PWD/ogl_beamforming_amd/libogl_$v.so
  for k in tpw; do
    f=gpurun_out/r03/abl_${v}_${k}.json
    timeout -k 10 120 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k span $v"
  done
done
