# The GPU-box commands behind profiles/<round>_* (round = $R, default r04): microbenchmarks, PMC passes (one counter group per rocprofv3 pass, tools/pmc_das.py),
# the default bench, the other configurations and the reference harness's frames, rocprofv3 kernel stats, the GPU test log.
# Two calls (each fits a gpurun limit), from the repository root:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh pmc'
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh bench'
# They write under gpurun_out/$R/ and ALSO refresh the tracked summaries in the box's copy; copy those back from gpurun_out/$R/profiles/.
# A leg that fails leaves its stderr under gpurun_out/$R/ and the script goes on (no `set -e`: one failing leg must not hide the others).
ROOT=$PWD
R=${R:-r04}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT $OUT/profiles
part=${1:-pmc}
if [ $part = pmc ]; then
timeout -k 10 300 tools/bin/microbench > $OUT/profiles/${R}_microbench.json 2> $OUT/microbench.err
cp $OUT/profiles/${R}_microbench.json profiles/${R}_microbench.json
echo "microbench done"
timeout -k 10 600 python3 tools/pmc_das.py --config 4 --groups 0,1,2,7,8,9,10 --timeout 120 --out $OUT/pmc_cfg4 > $OUT/pmc_cfg4.log 2>&1
timeout -k 10 200 python3 tools/pmc_das.py --config 4 --planes 16 --groups 3,5 --timeout 90 --out $OUT/pmc_cfg4_ta > $OUT/pmc_cfg4_ta.log 2>&1
timeout -k 10 300 python3 tools/pmc_das.py --config 2 --groups 0,1,3,5,7,8,9 --timeout 60 --out $OUT/pmc_cfg2 > $OUT/pmc_cfg2.log 2>&1
timeout -k 10 420 python3 tools/pmc_das.py --config 5 --planes 32 --groups 0,1,3,5,9 --timeout 90 --out $OUT/pmc_cfg5 > $OUT/pmc_cfg5.log 2>&1
for k in tpw hercules forces; do
  timeout -k 10 400 python3 tools/pmc_das.py --config harness:$k --groups 0,1,2,3,5,7,8,9,10,11 --timeout 60 --out $OUT/pmc_harness_$k > $OUT/pmc_harness_$k.log 2>&1
done
echo "pmc done"
python3 tools/summarize_profiles.py --round $R $OUT/pmc_cfg4/summary.json $OUT/pmc_cfg4_ta/summary.json $OUT/pmc_cfg2/summary.json $OUT/pmc_cfg5/summary.json \
        $OUT/pmc_harness_tpw/summary.json $OUT/pmc_harness_hercules/summary.json $OUT/pmc_harness_forces/summary.json
python3 tools/tile_pmc_summary.py $OUT/pmc_cfg2/summary.json profiles/${R}_pmc_tile_cfg2.json
cp profiles/das_traffic.json profiles/${R}_das_bound.json profiles/${R}_pmc_tile_cfg2.json $OUT/profiles/
exit 0
fi
# ---- part "bench" (expects profiles/${R}_microbench.json, das_traffic.json, ${R}_das_bound.json and ${R}_pmc_tile_cfg2.json of part "pmc" in the tree)
timeout -k 10 400 python bench.py > $OUT/profiles/${R}_bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/profiles/${R}_bench.json
for c in 1 2 3 5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
timeout -k 10 300 python bench.py --config 5 --interpolation cubic --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5_cubic.json 2> $OUT/bench_cfg5_cubic.err
timeout -k 10 300 python bench.py --das-path 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4_gather.json 2> $OUT/bench_cfg4_gather.err
for k in tpw tpw_swapped vls hercules forces; do
  timeout -k 10 200 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_harness_$k.json 2> $OUT/bench_harness_$k.err
  timeout -k 10 200 python bench.py --config harness:$k --das-path 1 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_harness_${k}_general.json 2> $OUT/last_leg.err
done
PYTHONPATH=. timeout -k 10 300 python tools/tile_threshold.py --json $OUT/profiles/${R}_tile_threshold.json > $OUT/tile_threshold.log 2>&1
timeout -k 10 300 python bench.py --in-process --devices 0,0 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_inprocess_0_0.json 2> $OUT/bench_inprocess.err
python3 - $R <<'PY'
import json, sys
R = sys.argv[1]
def line(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
def brief(d):
    return {"ms_per_step": d["ms_per_step"], "value": d["value"], "das_path": d["config"]["das_path"], "stage_ms": d["config"]["stage_ms"], "workload": d["config"]["workload"],
            "das_plan": d["config"].get("das_plan"),
            "roofline": {k: d["roofline"].get(k) for k in ("bound", "achieved", "frac", "kernel", "kernel_ms", "pairs_per_launch", "binding")}}
out = {}
for c in (1, 2, 3, 5):
    out[f"config{c}"] = brief(line(f"gpurun_out/{R}/bench_cfg{c}.json"))
out["config5_cubic_interpolation_the_harness_setting"] = brief(line("gpurun_out/" + R + "/bench_cfg5_cubic.json"))
out["config4_gather_kernel_das_path_2"] = brief(line("gpurun_out/" + R + "/bench_cfg4_gather.json"))
d = line("gpurun_out/" + R + "/bench_inprocess_0_0.json")
out["config4_in_process_two_contexts_on_one_gpu"] = {"ms_per_step": d["ms_per_step"], "sharding": d["config"]["sharding"], "devices": d["config"].get("devices"),
                                                      "rf_checksum_equal": d["config"]["rf_checksum_equal_on_all_ranks"], "note": "orchestration check only: both device contexts share one GPU"}
json.dump(out, open("gpurun_out/" + R + "/profiles/" + R + "_other_configs.json", "w"), indent=1)
h = {}
for k in ("tpw", "tpw_swapped", "vls", "hercules", "forces"):
    e = {"automatic": brief(line(f"gpurun_out/{R}/bench_harness_{k}.json"))}
    try: e["general_kernel_das_path_1"] = brief(line(f"gpurun_out/{R}/bench_harness_{k}_general.json"))
    except Exception as x: e["general_kernel_das_path_1"] = str(x)[:100]
    h[f"harness:{k}"] = e
json.dump(h, open("gpurun_out/" + R + "/profiles/" + R + "_harness.json", "w"), indent=1)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fast -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/profiles/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err
for k in tpw hercules forces; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$k -o harness_$k -- python3 $ROOT/bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_harness_${k}_rocprof.json 2> $OUT/rocprof_$k.err
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o cfg2 -- python3 $ROOT/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_rocprof.json 2> $OUT/rocprof2.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -o cfg5 -- python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5_rocprof.json 2> $OUT/rocprof5.err
cd $ROOT
for n in fast cfg2 cfg5 harness_tpw harness_hercules harness_forces; do f=$(find $OUT -name "${n}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/profiles/${R}_${n}_kernel_stats.csv; done
python3 tools/kernel_resources.py --json $OUT/profiles/${R}_kernel_resources.json > $OUT/kernel_resources.log 2>&1 || true
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/profiles/${R}_pytest_gpu.log
cat $OUT/profiles/${R}_pytest_gpu.log
