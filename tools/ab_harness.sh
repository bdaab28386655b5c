#!/bin/bash
# A/B of variant libraries (tools/build_variant.sh) on the harness frames: DAS ms per variant and kind.  usage: bash tools/ab_harness.sh TAG base VARIANT...
OUT=gpurun_out/r04/ab_$1; shift
mkdir -p $OUT
for name in "$@"; do
  if [ $name = base ]; then unset OGL_BEAMFORMER_LIB; else export OGL_BEAMFORMER_LIB=$PWD/build/variants/libogl_$name.so; fi
  for k in tpw vls forces hercules; do
    timeout -k 10 120 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${name}_$k.json 2> $OUT/${name}_$k.err
  done
done
python3 - $OUT "$@" <<'PY'
import json, sys
out = sys.argv[1]
for name in sys.argv[2:]:
    row = []
    for k in ("tpw", "vls", "forces", "hercules"):
        try:
            d = json.loads(open(f"{out}/{name}_{k}.json").read().strip().splitlines()[-1])
            row.append(f"{k} {d['config']['stage_ms']['DAS']:.2f} (path {d['config']['das_path']})")
        except Exception as e:
            row.append(f"{k} ERR {str(e)[:40]}")
    print(name, " | ".join(row))
PY
