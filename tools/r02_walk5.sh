set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02f; mkdir -p $OUT
for walk in plane depth; do
  export BEAMFORMER_HIP_TILE_WALK=$walk
  timeout -k 10 200 python bench.py --config 5 --planes 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$walk.json 2> $OUT/bench_$walk.err
  timeout -k 10 300 python3 tools/pmc_das.py --config 5 --planes 64 --groups 7,9 --timeout 140 --out $OUT/pmc_$walk > $OUT/pmc_$walk.log 2>&1
  timeout -k 10 200 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench3_$walk.json 2> $OUT/bench3_$walk.err
  echo "$walk done"
done
python3 - <<'PY'
import json
for w in ('plane','depth'):
    b=json.loads(open(f'gpurun_out/r02f/bench_{w}.json').read()); p=json.load(open(f'gpurun_out/r02f/pmc_{w}/summary.json')); b3=json.loads(open(f'gpurun_out/r02f/bench3_{w}.json').read())
    print(w, b['roofline']['kernel_ms'], b3['roofline']['kernel_ms'], {k:v for k,v in p['counters'].items()}, p['dispatches_summed'], p['failed_groups'])
PY
