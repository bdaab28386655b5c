#!/bin/bash
# When are the waves of the factored kernel resident?  (round-3 verdict item 4: "say what the harness-frame kernels wait for")
# A VARIANT build of das_factored.hip (a patched copy under build/variants -- the product source is not touched) in which every wave
# stores s_memrealtime at its start and end, its XCC / CU / SIMD and its tile row; tools/timeline_probe.py runs the reference harness's
# frames on it and reduces the stamps to: resident waves over time (whole chip, per XCD), when each XCD runs dry, wave duration by depth.
#   bash tools/timeline_probe.sh build                                   (no GPU)
#   gpurun -- 'bash tools/timeline_probe.sh run'                         -> gpurun_out/r04/timeline/r04_timeline.json
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/ogl_beamforming_amd/csrc
V=$ROOT/build/variants
if [ "${1:-build}" = build ]; then
  mkdir -p $V
  make -s -C $SRC
  d=$V/timeline; rm -rf $d; mkdir -p $d
  cp $SRC/*.h $d/; cp $SRC/das_factored.hip $d/
  python3 - $d/das_factored.hip <<'PY'
import sys
path = sys.argv[1]
s = open(path).read()
old = "	uint32_t tid = threadIdx.x;\n	uint32_t lx  = tid"
new = "	const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();\n	uint32_t tid = threadIdx.x;\n	uint32_t lx  = tid"
assert s.count(old) == 1
s = s.replace(old, new)
old = "		reinterpret_cast<sample_t<CPLX> *>(p.out)[out_index] = v;\n	}\n}\n"
new = ("		reinterpret_cast<sample_t<CPLX> *>(p.out)[out_index] = v;\n	}\n"
       "	{\n		const uint32_t w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);\n"
       "		if ((threadIdx.x & 63u) == 0 && w_ < 65536u) {\n"
       "			uint32_t hw_, xcc_;\n"
       "			asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw_));\n"
       "			asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc_));\n"
       "			bf_timeline[4 * w_ + 0] = tl_t0;\n			bf_timeline[4 * w_ + 1] = __builtin_amdgcn_s_memrealtime();\n"
       "			bf_timeline[4 * w_ + 2] = (unsigned long long)hw_ | ((unsigned long long)xcc_ << 32);\n"
       "			bf_timeline[4 * w_ + 3] = (unsigned long long)by | ((unsigned long long)bx << 32);\n		}\n	}\n}\n")
assert s.count(old) == 1
s = s.replace(old, new)
old = "namespace {\n"
assert old in s
s = s.replace(old, "__device__ unsigned long long bf_timeline[4 * 65536];\nextern \"C\" __attribute__((visibility(\"default\"))) int bf_debug_timeline(unsigned long long *out, unsigned count)\n"
              "{\n	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bf_timeline), (size_t)count * 8u);\n}\n" + old, 1)
open(path, "w").write(s)
PY
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter -c $d/das_factored.hip -o $d/das_factored.o
  objs=$(ls $SRC/build/*.o | grep -v "/das_factored.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libogl_timeline.so $objs $d/das_factored.o
  rm -rf $d
  echo built $V/libogl_timeline.so
  exit 0
fi
OUT=$ROOT/gpurun_out/r04/timeline
mkdir -p $OUT
cd $ROOT
export OGL_BEAMFORMER_LIB=$V/libogl_timeline.so PYTHONPATH=$ROOT
timeout -k 10 500 python3 tools/timeline_probe.py --json $OUT/r04_timeline.json > $OUT/timeline.log 2>&1
tail -40 $OUT/timeline.log
