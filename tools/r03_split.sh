#!/bin/bash
# the channel split of the factored kernel on frames of 8192 voxel waves (the harness planes): target waves per launch swept
for k in tpw forces; do
 for t in 4096 16384 32768 65536; do
  BEAMFORMER_HIP_SPLIT_TARGET=$t timeout -k 10 200 python bench.py --config harness:$k --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'target', $t, 'split', d['config']['das_plan']['split_shift'], 'DAS', round(d['config']['stage_ms']['DAS'],2), d['config']['das_plan']['tile_shift'], d['config']['das_plan']['blocks'])"
 done
done
