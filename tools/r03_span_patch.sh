#!/bin/bash
# the two span-staging entries of profiles/r03_harness.json (the rest of the file comes from tools/profile_round.sh bench)
OUT=gpurun_out/r03; mkdir -p $OUT
for k in tpw forces; do
  timeout -k 10 200 python bench.py --config harness:$k --das-path 64 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_harness_${k}_span_staging.json 2> /dev/null
  timeout -k 10 200 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_harness_${k}_again.json 2> /dev/null
  python - $k <<'PY'
import json,sys
k=sys.argv[1]
a=json.loads(open(f"gpurun_out/r03/bench_harness_{k}_span_staging.json").read().strip().splitlines()[-1])
b=json.loads(open(f"gpurun_out/r03/bench_harness_{k}_again.json").read().strip().splitlines()[-1])
print(k, "span", round(a["ms_per_step"],2), a["config"]["das_plan"]["span_stage"], "automatic on the same box", round(b["ms_per_step"],2), b["config"]["das_plan"]["span_stage"])
PY
done
