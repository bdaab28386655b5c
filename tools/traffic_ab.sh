#!/bin/bash
# Why the headline kernel's HBM-side traffic depends on how its tiles are dealt to the XCDs (round-3 verdict item 3): variant builds of
# das_staged.hip (patched COPIES under build/variants -- the product source is not touched), each timed and profiled (FETCH_SIZE, TCC
# hit / miss / request counts) on BASELINE config 4.
#   part "build" (no GPU; run here):  bash tools/traffic_ab.sh build
#   part "run"   (GPU box):           gpurun --timeout 1200 -- 'bash tools/traffic_ab.sh run'
# Variants:
#   base      the product as it shipped until round 4 (since then: quad_u): depth-major walk, two 1024-thread blocks per CU (64 tiles in flight per XCD = 64 consecutive depths of one column)
#   one_block 84 KB of LDS asked for: ONE block per CU (32 tiles in flight per XCD: half the depth span, half the phase spread)
#   pair_u    the 64 tiles in flight = 32 consecutive depths of TWO columns adjacent along u
#   quad_u    16 consecutive depths of FOUR columns adjacent along u;  oct_u: 8 x 8;  hex_u: 4 depths of a whole row of 16 columns
#   quad_v    16 consecutive depths of four columns adjacent along v;  quad_uv: 16 depths of 2 x 2 columns
# VARIANTS="a b c" limits either part to those names (profiles/r04_traffic.json was put together from two such runs: gpurun_out/ does not travel to the box).
#   plane     plane-major walk (x, y, z): the 64 tiles in flight lie in one plane
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/ogl_beamforming_amd/csrc
V=$ROOT/build/variants
part=${1:-build}
if [ $part = build ]; then
  mkdir -p $V
  make -s -C $SRC
  for name in ${VARIANTS:-one_block pair_u quad_u oct_u hex_u quad_v quad_uv plane}; do
    d=$V/$name; rm -rf $d; mkdir -p $d
    cp $SRC/*.h $d/; cp $SRC/das_staged.hip $d/
    python3 - $name $d/das_staged.hip <<'PY'
import sys
name, path = sys.argv[1], sys.argv[2]
s = open(path).read()
if name == "one_block":
    old = "	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);\n	if (e != hipSuccess) return e;\n	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);"
    new = "	const uint32_t lds_one = q->lds_bytes < 86016u ? 86016u : q->lds_bytes;\n	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_one);\n	if (e != hipSuccess) return e;\n	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), lds_one, s, *a, *q);"
else:
    old = "		zl = tile % q.tiles[2];\n		tu = (tile / q.tiles[2]) % q.tiles[0];\n		tv = tile / (q.tiles[2] * q.tiles[0]);"
    if name == "plane":
        new = "		tu = tile % q.tiles[0];\n		tv = (tile / q.tiles[0]) % q.tiles[1];\n		zl = tile / (q.tiles[0] * q.tiles[1]);"
    else:
        gu, gv = {"pair_u": (2, 1), "quad_u": (4, 1), "oct_u": (8, 1), "hex_u": (16, 1), "quad_v": (1, 4), "quad_uv": (2, 2)}[name]
        new = (f"		const uint32_t gu_ = {gu}u, gv_ = {gv}u, sub_ = tile % (gu_ * gv_), r_ = tile / (gu_ * gv_);      /* (tiles[0], tiles[1] divisible: config 4 has 16 x 16) */\n"
               "		zl = r_ % q.tiles[2];\n		const uint32_t col_ = r_ / q.tiles[2], per_row_ = q.tiles[0] / gu_;\n"
               "		tu = (col_ % per_row_) * gu_ + sub_ % gu_;\n		tv = (col_ / per_row_) * gv_ + sub_ / gu_;")
assert old in s, name
open(path, "w").write(s.replace(old, new))
PY
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter -c $d/das_staged.hip -o $d/das_staged.o
    objs=$(ls $SRC/build/*.o | grep -v "/das_staged.o" | grep -v "_NO_\|_ch[0-9]\|_old\|amdgcn")
    hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libogl_$name.so $objs $d/das_staged.o
    rm -rf $d
    echo built $V/libogl_$name.so
  done
  exit 0
fi
OUT=$ROOT/gpurun_out/r04/traffic
mkdir -p $OUT
cd $ROOT
NAMES=${VARIANTS:-base one_block pair_u quad_u oct_u hex_u quad_v quad_uv plane}
for name in $NAMES; do
  if [ $name = base ]; then unset OGL_BEAMFORMER_LIB; else export OGL_BEAMFORMER_LIB=$V/libogl_$name.so; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  timeout -k 10 400 python3 tools/pmc_das.py --config 4 --groups 7,9 --timeout 150 --out $OUT/pmc_$name > $OUT/pmc_$name.log 2>&1
  echo "$name done"
done
python3 - $NAMES <<'PY'
import json, os, sys
out = {"what": "BASELINE config 4, das_rca_staged_kernel<true,5,5,3,false>: kernel time and HBM-side traffic per launch under several ways of dealing the tiles to the XCDs (tools/traffic_ab.sh)", "variants": {}}
for name in sys.argv[1:]:
    try:
        b = json.loads(open(f"gpurun_out/r04/traffic/bench_{name}.json").read().strip().splitlines()[-1])
        summary = json.load(open(f"gpurun_out/r04/traffic/pmc_{name}/summary.json"))
        c, n = summary["counters"], summary["dispatches_summed"]          # (bench.py's pair-count frame runs the DAS kernel too: two dispatches)
        out["variants"][name] = {"das_ms": b["config"]["stage_ms"]["DAS"], "fetch_GB": 2 * c["FETCH_SIZE"] * 1024 / 1e9 / n,
                                 "tcc_req_G": c["TCC_REQ_sum"] / 1e9 / n, "tcc_miss_G": c["TCC_MISS_sum"] / 1e9 / n,
                                 "l2_hit": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])}
    except Exception as e:
        out["variants"][name] = {"error": str(e)[:200]}
json.dump(out, open("gpurun_out/r04/traffic/r04_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
