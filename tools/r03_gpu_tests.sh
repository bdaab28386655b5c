#!/bin/bash
# the whole GPU suite, log under gpurun_out/r03
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r03/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/r03/pytest_gpu.log
