set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02e; mkdir -p $OUT
for walk in plane depth; do
  export BEAMFORMER_HIP_TILE_WALK=$walk
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$walk.json 2> $OUT/bench_$walk.err
  timeout -k 10 300 python3 tools/pmc_das.py --config 4 --groups 7,9 --timeout 140 --out $OUT/pmc_$walk > $OUT/pmc_$walk.log 2>&1
  echo "$walk done"
done
unset BEAMFORMER_HIP_TILE_WALK
timeout -k 10 200 python3 tools/pmc_das.py --config 4 --planes 8 --groups 3 --timeout 100 --out $OUT/pmc_ta > $OUT/pmc_ta.log 2>&1
echo ta done
python3 - <<'PY'
import json
for w in ('plane','depth'):
    b=json.loads(open(f'gpurun_out/r02e/bench_{w}.json').read()); p=json.load(open(f'gpurun_out/r02e/pmc_{w}/summary.json'))
    print(w, b['ms_per_step'], b['roofline']['kernel_ms'], {k:v for k,v in p['counters'].items()}, p['dispatches_summed'], p['failed_groups'])
print(json.load(open('gpurun_out/r02e/pmc_ta/summary.json'))['counters'])
PY
