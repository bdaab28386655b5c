#!/bin/bash
# A/B on one box: the shipping library against ogl_beamforming_amd/libogl_old.so (the previous das_staged.hip), config 4, alternating
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staged" 2>&1 | tail -2
for rep in 1 2; do
 for v in new old; do
  if [ $v = old ]; then export OGL_BEAMFORMER_LIB=$PWD/ogl_beamforming_amd/libogl_old.so; else unset OGL_BEAMFORMER_LIB; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03/ab_staged_${v}_$rep.json 2>/dev/null
  python - gpurun_out/r03/ab_staged_${v}_$rep.json "$v $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],2), "ms", d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],2), "binding frac", round(d["roofline"]["binding"]["frac"],3))
PY
 done
done
