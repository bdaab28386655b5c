# The GPU-box command behind profiles/r01_*: tests, default bench, per-config benches, rocprofv3
# kernel stats and the two PMC passes (one counter per pass).  Run from the repository root:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'   then copy summaries from gpurun_out/ into profiles/.
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/r01n
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
for c in 1 2 3 5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fast -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_rocprof.json 2> $OUT/rocprof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o cfg2 -- python3 $ROOT/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_rocprof.json 2> $OUT/rocprof2.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o cfg1 -- python3 $ROOT/bench.py --config 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cfg1_rocprof.json 2> $OUT/rocprof1.err
cd $ROOT
find gpurun_out/r01n -name "*stats.csv" | head
