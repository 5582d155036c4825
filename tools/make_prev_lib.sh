#!/bin/bash
# tools/probe/ab/libnsp_hip_prev.so = the tree's library with the listed csrc files taken from a git revision
# (default HEAD): the "before" arm of the A/B tools (NSP_LIB_OVERRIDE).  usage: tools/make_prev_lib.sh [rev] file.hip ...
set -e
rev=HEAD
if [[ "$1" != *.hip ]]; then rev=$1; shift; fi
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
cp $root/neural_sp_amd/lib/*.o $tmp/
mkdir -p $tmp/src/neural_sp_amd/csrc $tmp/src/include $root/tools/probe/ab
git -C $root show $rev:include/nsp_hip.h > $tmp/src/include/nsp_hip.h
git -C $root show $rev:neural_sp_amd/csrc/common.h > $tmp/src/neural_sp_amd/csrc/common.h
for f in "$@"; do
  git -C $root show $rev:neural_sp_amd/csrc/$f > $tmp/src/neural_sp_amd/csrc/$f
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -c $tmp/src/neural_sp_amd/csrc/$f -o $tmp/${f%.hip}.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/probe/ab/libnsp_hip_prev.so $tmp/*.o
rm -rf $tmp
echo built $root/tools/probe/ab/libnsp_hip_prev.so from $rev: "$@"
