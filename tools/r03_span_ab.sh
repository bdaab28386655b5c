#!/bin/bash
# wave-span staging (automatic on the harness's coarse grid) against the factored kernel's gather loop (flag 0x80), every harness kind, one box
mkdir -p gpurun_out/r03
for k in tpw tpw_swapped vls forces; do
  for path in 0 128; do
    f=gpurun_out/r03/span_ab_${k}_${path}.json
    timeout -k 10 200 python bench.py --config harness:$k --das-path $path --steps 20 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err || { tail -2 ${f%.json}.err; continue; }
    python - $f $k $path <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d["config"]["das_plan"]
print(sys.argv[2], "das-path", sys.argv[3], "DAS", round(d["config"]["stage_ms"]["DAS"],2), "ms", "span" if p["span_stage"] else "gather", p["tile_shift"], p["blocks"])
PY
  done
done
for path in 512; do
  f=gpurun_out/r03/span_ab_cfg2_${path}.json
  timeout -k 10 200 python bench.py --config 2 --das-path $path --steps 20 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
  python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config 2, per-voxel factored kernel (0x200): DAS", round(d["config"]["stage_ms"]["DAS"],3), "ms")
PY
done
