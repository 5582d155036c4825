#!/bin/bash
# tools/probe/ab/libnsp_hip_<name>.so = the tree's library with ONE csrc file recompiled with extra flags
# usage: tools/make_variant_lib.sh <name> <file.hip> <flags...>
set -e
name=$1; f=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
cp $root/neural_sp_amd/lib/*.o $tmp/
mkdir -p $root/tools/probe/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result "$@" -c $root/neural_sp_amd/csrc/$f -o $tmp/${f%.hip}.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/probe/ab/libnsp_hip_$name.so $tmp/*.o
rm -rf $tmp
echo built tools/probe/ab/libnsp_hip_$name.so: $f "$@"
