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
for ch in 2 3 4; do
  if [ $ch = 4 ]; then unset OGL_BEAMFORMER_LIB; else export OGL_BEAMFORMER_LIB=$PWD/ogl_beamforming_amd/libogl_ch$ch.so; fi
  for k in tpw forces; do
    f=gpurun_out/r03/ch${ch}_${k}.json
    timeout -k 10 120 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k span chunk $ch"
  done
done
