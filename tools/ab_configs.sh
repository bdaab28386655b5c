#!/bin/bash
# A/B of variant libraries (tools/build_variant.sh) on BASELINE configurations: DAS ms per variant.  usage: bash tools/ab_configs.sh TAG "2 3" base VARIANT...
OUT=gpurun_out/r04/abc_$1; shift
CONFIGS=$1; shift
mkdir -p $OUT
for name in "$@"; do
  if [ $name = base ]; then unset OGL_BEAMFORMER_LIB; else export OGL_BEAMFORMER_LIB=$PWD/build/variants/libogl_$name.so; fi
  for c in $CONFIGS; do
    for r in 1 2; do timeout -k 10 200 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline > $OUT/${name}_cfg${c}_$r.json 2> $OUT/err; done
  done
done
python3 - $OUT "$CONFIGS" "$@" <<'PY'
import json, sys
out, configs = sys.argv[1], sys.argv[2].split()
for name in sys.argv[3:]:
    row = []
    for c in configs:
        v = []
        for r in (1, 2):
            try: v.append(round(json.loads(open(f"{out}/{name}_cfg{c}_{r}.json").read().strip().splitlines()[-1])["config"]["stage_ms"]["DAS"], 3))
            except Exception as e: v.append(str(e)[:30])
        row.append(f"cfg{c} {v}")
    print(name, " | ".join(row))
PY
