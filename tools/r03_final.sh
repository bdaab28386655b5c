#!/bin/bash
# the default bench line (with the CPU baseline) and the whole GPU suite, for profiles/r03_bench.json and r03_pytest_gpu.log
mkdir -p gpurun_out/r03/profiles
timeout -k 10 500 python bench.py > gpurun_out/r03/profiles/r03_bench.json 2> gpurun_out/r03/bench.err
echo "bench exit $?"; cut -c1-300 gpurun_out/r03/profiles/r03_bench.json
timeout -k 10 200 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_cfg2_final.json 2> gpurun_out/r03/bench_cfg2_final.err
cut -c1-300 gpurun_out/r03/bench_cfg2_final.json
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r03/profiles/r03_pytest_gpu.log
cat gpurun_out/r03/profiles/r03_pytest_gpu.log
