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
 for t in 4096 16384 32768 65536 131072; do
   for path in 0 1; do
    [ $k != hercules ] && [ $path = 1 ] && continue
    f=gpurun_out/r03/split_${k}_${t}_${path}.json
    BEAMFORMER_HIP_SPLIT_WAVES=$t timeout -k 10 120 python bench.py --config harness:$k --das-path $path --steps 10 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k split target $t path $path"
   done
 done
done
