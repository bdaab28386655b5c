#!/bin/bash
# config 2 after a change to das_tile.hip or its selection: PMC pass -> profiles/r03_pmc_tile_cfg2.json + das_traffic / das_bound entries, bench line, rocprofv3 kernel stats
set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT/profiles
timeout -k 10 300 python3 tools/pmc_das.py --config 2 --groups 0,1,3,5,7,8,9 --timeout 60 --out $OUT/pmc_cfg2 > $OUT/pmc_cfg2.log 2>&1
python3 tools/summarize_profiles.py --round r03 --merge $OUT/pmc_cfg2/summary.json
python3 tools/tile_pmc_summary.py $OUT/pmc_cfg2/summary.json profiles/r03_pmc_tile_cfg2.json
cp profiles/das_traffic.json profiles/r03_das_bound.json profiles/r03_pmc_tile_cfg2.json $OUT/profiles/
timeout -k 10 200 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
cut -c1-260 $OUT/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o cfg2 -- python3 $ROOT/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_rocprof.json 2> $OUT/rocprof2.err
cd $ROOT
f=$(find $OUT/stats2 -name "cfg2_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/profiles/r03_cfg2_kernel_stats.csv
head -3 $OUT/profiles/r03_cfg2_kernel_stats.csv
