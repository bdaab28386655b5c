#!/bin/bash
mkdir -p gpurun_out/r03
export BEAMFORMER_HIP_SPAN=1
timeout -k 10 500 python3 tools/pmc_das.py --config harness:tpw --das-path 0 --groups 0,1,2,3,5,9 --timeout 90 --out gpurun_out/r03/pmc_tpw_span > gpurun_out/r03/pmc_tpw_span.log 2>&1
echo done
