#!/bin/bash
# same box: HBM-side traffic of the config-4 DAS launch, round-2 tree (build/r02tree) against this tree
mkdir -p gpurun_out/r03
ROOT=$PWD
for tree in new old new old; do
  if [ $tree = old ]; then cd $ROOT/build/r02tree; else cd $ROOT; fi
  timeout -k 10 300 python3 tools/pmc_das.py --config 4 --groups 7,9 --timeout 120 --out $ROOT/gpurun_out/r03/ab_traffic_$tree > $ROOT/gpurun_out/r03/ab_traffic_$tree.log 2>&1
  python3 - $ROOT/gpurun_out/r03/ab_traffic_$tree/summary.json $tree <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["counters"]; n=d["dispatches_summed"] or 1
print(sys.argv[2], d["kernels"], "FETCH GB", round(c["FETCH_SIZE"]*2*1024/n/1e9,1), "L2 hit", round(c["TCC_HIT_sum"]/(c["TCC_HIT_sum"]+c["TCC_MISS_sum"]),3), "TCC req", "%.3g"%(c["TCC_REQ_sum"]/n))
PY
done
