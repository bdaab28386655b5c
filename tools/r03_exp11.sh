#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plane or harness or config1 or config2" 2>&1 | tail -2
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
 for walk in plane column band; do
  for opt in 0 1; do
    unset BEAMFORMER_HIP_TILE_WALK BEAMFORMER_HIP_SPAN BEAMFORMER_HIP_HERCULES_NOPAIRS
    [ $walk = band ] || export BEAMFORMER_HIP_TILE_WALK=$walk
    if [ $opt = 1 ]; then if [ $k = hercules ]; then export BEAMFORMER_HIP_HERCULES_NOPAIRS=1; else export BEAMFORMER_HIP_SPAN=1; fi; fi
    f=gpurun_out/r03/band_${k}_${walk}_${opt}.json
    timeout -k 10 120 python bench.py --config harness:$k --steps 10 --warmup 3 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k walk $walk opt(span/nopairs) $opt"
  done
 done
done
unset BEAMFORMER_HIP_TILE_WALK BEAMFORMER_HIP_SPAN BEAMFORMER_HIP_HERCULES_NOPAIRS
timeout -k 10 120 python bench.py --config harness:hercules --das-path 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/band_herc_general.json 2>/dev/null; line gpurun_out/r03/band_herc_general.json "hercules general band"
timeout -k 10 120 python bench.py --config harness:tpw --das-path 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/band_tpw_general.json 2>/dev/null; line gpurun_out/r03/band_tpw_general.json "tpw general band"
timeout -k 10 120 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/band_cfg2.json 2>/dev/null; line gpurun_out/r03/band_cfg2.json "config2 band"
BEAMFORMER_HIP_TILE_WALK=plane timeout -k 10 120 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/band_cfg2p.json 2>/dev/null; line gpurun_out/r03/band_cfg2p.json "config2 plane"
