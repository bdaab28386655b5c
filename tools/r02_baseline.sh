# Round-2 opening measurement: microbenchmarks + PMC passes of the round-1 DAS kernels (baseline).
set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT
timeout -k 10 300 tools/bin/microbench > $OUT/microbench.json 2> $OUT/microbench.err
echo "microbench done"; head -c 600 $OUT/microbench.json; echo
timeout -k 10 500 python3 tools/pmc_das.py --config 4 --planes 64 --out $OUT/pmc_cfg4 > $OUT/pmc_cfg4.log 2>&1
echo "cfg4 pmc done"
timeout -k 10 300 python3 tools/pmc_das.py --config 5 --planes 16 --groups 0,1,3,4 --out $OUT/pmc_cfg5 > $OUT/pmc_cfg5.log 2>&1
echo "cfg5 pmc done"
timeout -k 10 300 python3 tools/pmc_das.py --config 2 --groups 0,1,3,4 --out $OUT/pmc_cfg2 > $OUT/pmc_cfg2.log 2>&1
echo "cfg2 pmc done"
