#!/bin/bash
# tools/build_variant.sh NAME "FLAGS" file1.hip [file2.hip ...]: a second build of the library under build/var/NAME/ with the
# named sources recompiled with FLAGS (-D switches of an experiment) and every other object taken from the main build;
# A/B on one box through AECF_LIB_PATH=aecf_amd/lib/var/NAME/libaecf_hip.so (tools/gpu_jobs/ab_libs.sh; aecf_amd/lib/ is
# git-ignored but travels to the GPU box, build/ does not)
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/aecf_amd/csrc; obj=$root/build/obj; out=$root/aecf_amd/lib/var/$name
mkdir -p $out/obj
objs=""
for o in $obj/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" = "$b" ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $flags -I$src -I$root/include -c $src/$b.hip -o $out/obj/$b.o
      use=$out/obj/$b.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libaecf_hip.so $objs
echo "built $out/libaecf_hip.so"
