set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02c; mkdir -p $OUT
timeout -k 10 400 python3 tools/pmc_das.py --config 5 --planes 8 --groups 0,1,2 --timeout 150 --out $OUT/pmc_cfg5_herc > $OUT/pmc_cfg5_herc.log 2>&1
echo sq done
timeout -k 10 200 python3 tools/pmc_das.py --config 5 --planes 8 --groups 4 --timeout 150 --out $OUT/pmc_cfg5_herc_tcp > $OUT/pmc_cfg5_herc_tcp.log 2>&1
echo tcp done
