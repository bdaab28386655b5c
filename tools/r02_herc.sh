set -e
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02b; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hercules or config3 or config5 or frame_parity" 2>&1 | tail -15 > $OUT/pytest_herc.log
cat $OUT/pytest_herc.log
for path in 0 1; do
  timeout -k 10 200 python bench.py --config 5 --planes 16 --steps 2 --warmup 1 --no-cpu-baseline --das-path $path > $OUT/cfg5_p$path.json 2> $OUT/cfg5_p$path.err
  timeout -k 10 200 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --das-path $path > $OUT/cfg3_p$path.json 2> $OUT/cfg3_p$path.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/cfg*_p*.json')):
    d=json.loads(open(f).read())
    print(f, d['ms_per_step'], d['config']['das_path'], d['roofline']['kernel_ms'])
PY
