#!/bin/bash
# Variant builds for A/B runs: patched COPIES of csrc files under build/variants/<name>, linked with the product's other objects into
# build/variants/libogl_<name>.so (the product source is not touched; OGL_BEAMFORMER_LIB=<that .so> selects it in lib.py).
# usage: tools/build_variant.sh NAME file:patch.py ...   (patch.py gets the copied file's path; tools/variants/*.py are the ones round 4 measured)
#   tools/build_variant.sh lpt        bf_kernels.h:tools/variants/lpt.py das_factored.hip:tools/variants/noop.py das_hercules.hip:tools/variants/noop.py
#   tools/build_variant.sh lpt_split4 bf_kernels.h:tools/variants/lpt.py das_factored.hip:tools/variants/noop.py das_hercules.hip:tools/variants/noop.py das_select.cpp:tools/variants/split4.py
#   tools/build_variant.sh lpt_w64    bf_kernels.h:tools/variants/lpt.py das_factored.hip:tools/variants/threads64.py das_hercules.hip:tools/variants/noop.py das_select.cpp:tools/variants/tile6.py
#   tools/build_variant.sh walk1      bf_kernels.h:tools/variants/walk1.py das_select.cpp:tools/variants/noop.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/ogl_beamforming_amd/csrc; V=$ROOT/build/variants
name=$1; shift
d=$V/$name; rm -rf $d; mkdir -p $d
cp $SRC/*.h $d/
replaced=""
for spec in "$@"; do
  f=${spec%%:*}; patch=${spec#*:}
  [ -f $d/$f ] || cp $SRC/$f $d/
  python3 $patch $d/$f
done
objs=""
for f in $(ls $d | grep -E "\.(hip|cpp)$"); do
  o=$d/${f%.*}.o
  if [ ${f##*.} = hip ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter -c $d/$f -o $o
  else (cd $d && hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter -I$SRC -c $f -o $o); fi
  replaced="$replaced ${f%.*}.o"
  objs="$objs $o"
done
for o in $SRC/build/*.o; do b=$(basename $o); case " $replaced " in *" $b "*) ;; *) objs="$objs $o";; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libogl_$name.so $objs
rm -rf $d
echo built $V/libogl_$name.so
