#!/bin/bash
# round 3 experiment: wave-span staging in the factored kernel -- parity (oracle + bit-equality with the gather loop), then the harness frames
mkdir -p gpurun_out/r03
BEAMFORMER_HIP_SPAN=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_factored_kernel and False" > gpurun_out/r03/span_parity.log 2>&1
echo "span parity rc $?"; tail -3 gpurun_out/r03/span_parity.log
timeout -k 10 300 python - <<'PY' 2>&1 | tail -20
import os, numpy as np
from tests import cases
from ogl_beamforming_amd import lib
L = lib.library()
L.beamformer_hip_set_das_path(0x14)
for name in ["harness_tpw_small", "harness_forces_small", "harness_vls_small", "harness_tpw_swapped_small", "harness_tpw_yz_small", "config2_small", "rca_staged_cubic", "rca_staged_auto", "uforces_sparse", "rca_vls_staged_short_rows"]:
    acq = cases.make(name)
    os.environ.pop("BEAMFORMER_HIP_SPAN", None)
    a = np.asarray(lib.beamform(acq.bp, acq.rf, acq.filters)).copy()
    os.environ["BEAMFORMER_HIP_SPAN"] = "1"
    b = np.asarray(lib.beamform(acq.bp, acq.rf, acq.filters)).copy()
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(name, "bit-identical" if same else f"DIFFER max {np.nanmax(np.abs(a-b)):.3e} of {np.nanmax(np.abs(a)):.3e}")
PY
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],3), "ms", d["config"]["das_path"], "DAS", round(d["config"]["stage_ms"]["DAS"],3))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for k in tpw tpw_swapped vls forces; do
  for span in 0 1; do
    f=gpurun_out/r03/span_${k}_${span}.json
    if [ $span = 1 ]; then export BEAMFORMER_HIP_SPAN=1; else unset BEAMFORMER_HIP_SPAN; fi
    timeout -k 10 120 python bench.py --config harness:$k --steps 10 --warmup 2 --no-cpu-baseline > $f 2> ${f%.json}.err
    line $f "$k span $span"
  done
done
export BEAMFORMER_HIP_SPAN=1
timeout -k 10 120 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03/span_cfg2_1.json 2>/dev/null; line gpurun_out/r03/span_cfg2_1.json "config2 span 1"
unset BEAMFORMER_HIP_SPAN
timeout -k 10 120 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03/span_cfg2_0.json 2>/dev/null; line gpurun_out/r03/span_cfg2_0.json "config2 span 0"
