#!/bin/bash
# usage: tools/build_rev.sh <git-rev> <out.so>   - builds librua_hip.so of another revision (A/B runs on ONE box:
# RUA_LIB_PATH=<out.so> python bench.py ...; boxes differ by +-1.5 %, more than most single changes)
set -e
rev=$1; out=$2
tmp=$(mktemp -d)
mkdir -p $tmp/resunet_a_mltsk_keras_amd/csrc $tmp/include
for f in $(git ls-tree --name-only $rev resunet_a_mltsk_keras_amd/csrc/); do git show $rev:$f > $tmp/$f; done
git show $rev:include/rua_hip.h > $tmp/include/rua_hip.h
objs=""
for s in $tmp/resunet_a_mltsk_keras_amd/csrc/*.hip $tmp/resunet_a_mltsk_keras_amd/csrc/*.cpp; do
  ff=""; case $s in *conv_strip.hip) ff="-fno-slp-vectorize";; esac      # build.py::FILE_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $ff -x hip -c $s -o $s.o &
  objs="$objs $s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
rm -rf $tmp
echo built $out from $rev
