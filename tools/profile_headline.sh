# The headline kernel's part of tools/profile_round.sh alone (PMC passes of config 4's default DAS kernel, merged into the tracked
# summaries; default bench; the gather kernel beside it; rocprofv3 kernel stats): for a change that touches that kernel only.
#   gpurun --timeout 600 -- 'bash tools/profile_headline.sh'
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT $OUT/profiles
timeout -k 10 300 python3 tools/pmc_das.py --config 4 --groups 0,1,2,7,8,9 --timeout 120 --out $OUT/pmc_cfg4 > $OUT/pmc_cfg4.log 2>&1
timeout -k 10 120 python3 tools/pmc_das.py --config 4 --planes 16 --groups 3,5 --timeout 90 --out $OUT/pmc_cfg4_ta > $OUT/pmc_cfg4_ta.log 2>&1
echo "pmc done"
python3 tools/summarize_profiles.py --round r02 --merge $OUT/pmc_cfg4/summary.json $OUT/pmc_cfg4_ta/summary.json
cp profiles/das_traffic.json profiles/r02_das_bound.json $OUT/profiles/
timeout -k 10 300 python bench.py > $OUT/profiles/r02_bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/profiles/r02_bench.json
timeout -k 10 120 python bench.py --das-path 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4_gather.json 2> $OUT/bench_cfg4_gather.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fast -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/profiles/r02_bench_under_rocprof.json 2> $OUT/rocprof.err
f=$(find $OUT/stats -name "fast_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/profiles/r02_fast_kernel_stats.csv
