# The GPU-box command behind profiles/r02_*: microbenchmarks, tests, the default bench, per-config benches,
# rocprofv3 kernel stats, and the PMC passes (one counter group per pass, tools/pmc_das.py).  From the repository root:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'
# It writes under gpurun_out/r02/ and ALSO refreshes the tracked summaries (profiles/das_traffic.json,
# profiles/r02_das_bound.json, profiles/r02_*.json|csv) in the box's copy; copy those back from gpurun_out/r02/profiles/.
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT $OUT/profiles
timeout -k 10 300 tools/bin/microbench > $OUT/profiles/r02_microbench.json 2> $OUT/microbench.err
cp $OUT/profiles/r02_microbench.json profiles/r02_microbench.json
echo "microbench done"
# PMC passes first (bench.py reads their summaries): whole frames for the traffic figure, slabs for the rest
timeout -k 10 480 python3 tools/pmc_das.py --config 4 --groups 0,1,2,7,8,9 --timeout 120 --out $OUT/pmc_cfg4 > $OUT/pmc_cfg4.log 2>&1
timeout -k 10 200 python3 tools/pmc_das.py --config 4 --planes 16 --groups 3,5 --timeout 90 --out $OUT/pmc_cfg4_ta > $OUT/pmc_cfg4_ta.log 2>&1
# the gather kernel the LDS-staged kernel replaced as the default (das path 2), for the comparison DESIGN.md quotes
timeout -k 10 420 python3 tools/pmc_das.py --config 4 --das-path 2 --groups 0,1,7,8,9 --timeout 120 --out $OUT/pmc_cfg4_gather > $OUT/pmc_cfg4_gather.log 2>&1
timeout -k 10 200 python3 tools/pmc_das.py --config 4 --das-path 2 --planes 16 --groups 3,5 --timeout 90 --out $OUT/pmc_cfg4_gather_ta > $OUT/pmc_cfg4_gather_ta.log 2>&1
timeout -k 10 300 python3 tools/pmc_das.py --config 2 --groups 0,1,3,5,7,8,9 --timeout 60 --out $OUT/pmc_cfg2 > $OUT/pmc_cfg2.log 2>&1
timeout -k 10 300 python3 tools/pmc_das.py --config 3 --groups 0,1,3,5,7,8,9 --timeout 60 --out $OUT/pmc_cfg3 > $OUT/pmc_cfg3.log 2>&1
timeout -k 10 420 python3 tools/pmc_das.py --config 5 --planes 32 --groups 0,1,3,5,9 --timeout 90 --out $OUT/pmc_cfg5 > $OUT/pmc_cfg5.log 2>&1
echo "pmc done"
# the TA/TCP groups of config 4 come from a 16-plane slab (those passes are slow on whole frames): merged into the whole-frame entry
python3 tools/summarize_profiles.py --round r02 $OUT/pmc_cfg4/summary.json $OUT/pmc_cfg4_ta/summary.json $OUT/pmc_cfg4_gather/summary.json $OUT/pmc_cfg4_gather_ta/summary.json $OUT/pmc_cfg2/summary.json $OUT/pmc_cfg3/summary.json $OUT/pmc_cfg5/summary.json
cp profiles/das_traffic.json profiles/r02_das_bound.json $OUT/profiles/
timeout -k 10 400 python bench.py > $OUT/profiles/r02_bench.json 2> $OUT/bench.err
cut -c1-600 $OUT/profiles/r02_bench.json
for c in 1 2 3 5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
timeout -k 10 300 python bench.py --config 5 --frame-graph --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_graph.json 2> $OUT/bench_cfg5_graph.err
timeout -k 10 300 python bench.py --config 1 --frame-graph --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg1_graph.json 2> $OUT/bench_cfg1_graph.err
timeout -k 10 300 python bench.py --das-path 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4_gather.json 2> $OUT/bench_cfg4_gather.err
PYTHONPATH=$ROOT timeout -k 10 300 python tools/staged_threshold.py --json $OUT/profiles/r02_staged_threshold.json > $OUT/staged_threshold.log 2>&1 || echo "staged threshold sweep failed"
PYTHONPATH=$ROOT timeout -k 10 300 python tools/staged_threshold.py --real --json $OUT/profiles/r02_staged_threshold_real.json > $OUT/staged_threshold_real.log 2>&1 || echo "staged threshold sweep (real) failed"
PYTHONPATH=$ROOT timeout -k 10 300 python tools/staged_threshold.py --cubic --json $OUT/profiles/r02_staged_threshold_cubic.json > $OUT/staged_threshold_cubic.log 2>&1 || echo "staged threshold sweep (cubic) failed"
PYTHONPATH=$ROOT timeout -k 10 200 python tools/staged_uniform.py --json $OUT/profiles/r02_staged_uniform.json > $OUT/staged_uniform.log 2>&1 || echo "staged uniform-tables comparison failed"
PYTHONPATH=$ROOT timeout -k 10 300 python tools/pull_rate.py --json $OUT/profiles/r02_pull_rate.json > $OUT/pull_rate.log 2>&1 || echo "pull rate failed"
timeout -k 10 300 python bench.py --in-process --devices 0,0 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_inprocess_0_0.json 2> $OUT/bench_inprocess.err
python3 - <<'PY'
import json
out = {}
for c in (1, 2, 3, 5):
    d = json.loads(open(f"gpurun_out/r02/bench_cfg{c}.json").read())
    out[f"config{c}"] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "das_path": d["config"]["das_path"], "stage_ms": d["config"]["stage_ms"],
                         "workload": d["config"]["workload"], "roofline": {k: d["roofline"][k] for k in ("achieved", "frac", "kernel", "kernel_ms", "pairs_per_launch", "binding")}}
for c in (1, 5):
    d = json.loads(open(f"gpurun_out/r02/bench_cfg{c}_graph.json").read())
    out[f"config{c}_frame_graph"] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "notes": d["config"]["notes"]}
d = json.loads(open("gpurun_out/r02/bench_cfg4_gather.json").read())
out["config4_gather_kernel_das_path_2"] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "das_path": d["config"]["das_path"],
                                           "roofline": {k: d["roofline"][k] for k in ("achieved", "frac", "kernel", "kernel_ms", "pairs_per_launch", "binding")},
                                           "note": "the kernel that was the default until the LDS-staged kernel replaced it (same box, same run as the other entries)"}
d = json.loads(open("gpurun_out/r02/bench_inprocess_0_0.json").read())
out["config4_in_process_two_contexts_on_one_gpu"] = {"ms_per_step": d["ms_per_step"], "sharding": d["config"]["sharding"], "device_das_ms": d["config"].get("device_das_ms"),
                                                      "note": "orchestration check only: both device contexts share one GPU"}
json.dump(out, open("gpurun_out/r02/profiles/r02_other_configs.json", "w"), indent=1)
PY
for c in 1 2 5; do
  PYTHONPATH=$ROOT timeout -k 10 200 python tools/graph_probe.py --configs $c > $OUT/graph_probe_$c.json 2> $OUT/graph_probe_$c.err || echo "graph probe $c failed"
done
python3 - <<'PY'
import json
out = {}
for c in (1, 2, 5):
    try:
        out.update(json.loads(open(f"gpurun_out/r02/graph_probe_{c}.json").read()))
    except Exception as e:
        out[f"config{c}"] = {"probe_error": str(e)[:200]}
json.dump(out, open("gpurun_out/r02/profiles/r02_graph_probe.json", "w"), indent=1)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fast -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/profiles/r02_bench_under_rocprof.json 2> $OUT/rocprof.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -o cfg5 -- python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5_rocprof.json 2> $OUT/rocprof5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -o cfg3 -- python3 $ROOT/bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg3_rocprof.json 2> $OUT/rocprof3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o cfg2 -- python3 $ROOT/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_rocprof.json 2> $OUT/rocprof2.err
cd $ROOT
for n in fast cfg5 cfg3 cfg2; do f=$(find $OUT -name "${n}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/profiles/r02_${n}_kernel_stats.csv; done
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/profiles/r02_pytest_gpu.log
cat $OUT/profiles/r02_pytest_gpu.log
