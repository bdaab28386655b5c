#!/bin/bash
mkdir -p gpurun_out/r03
python tools/r03_dbg.py rca_a1s2 rca_staged_auto rca_vls_staged_short_rows harness_tpw_small harness_forces_small config2_small rca_staged_cubic uforces_sparse harness_vls_small harness_tpw_yz_small 2>&1 | grep -v "^  " | tail -12
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],3), "ms", d["config"]["das_path"], "DAS", round(d["config"]["stage_ms"]["DAS"],3))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for k in tpw tpw_swapped vls forces hercules; do
  for span in 0 1; do
    f=gpurun_out/r03/span2_${k}_${span}.json
    if [ $span = 1 ]; then export BEAMFORMER_HIP_SPAN=1; else unset BEAMFORMER_HIP_SPAN; fi
    timeout -k 10 120 python bench.py --config harness:$k --steps 10 --warmup 2 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k span $span"
  done
done
unset BEAMFORMER_HIP_SPAN
timeout -k 10 120 python bench.py --config harness:hercules --das-path 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03/span2_herc_general.json 2>/dev/null; line gpurun_out/r03/span2_herc_general.json "hercules general"
export BEAMFORMER_HIP_SPAN=1
timeout -k 10 300 python3 tools/pmc_das.py --config harness:tpw --das-path 0 --groups 0,1,2,3 --timeout 90 --out gpurun_out/r03/pmc_tpw_span2 > gpurun_out/r03/pmc_tpw_span2.log 2>&1
