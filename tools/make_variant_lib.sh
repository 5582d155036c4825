#!/bin/bash
# tools/variants/libnsp_hip_<name>.so = the tree's library with ONE object rebuilt from another source file and / or extra flags
# (the directory travels to the GPU box: *.so is git-ignored, not gpurun-ignored; select with NSP_LIB_OVERRIDE).
# usage: tools/make_variant_lib.sh <name> <object stem, e.g. flash_attn> <source.hip> [extra flags...]
set -e
name=$1; stem=$2; src=$3; shift 3
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
cp $root/neural_sp_amd/lib/*.o $tmp/
mkdir -p $root/tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result \
  -Xclang -target-feature -Xclang -packed-fp32-ops -I$root/neural_sp_amd/csrc -I$root/include "$@" -c $src -o $tmp/$stem.o 2>&1 | grep -v "not a recognized feature" || true
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/variants/libnsp_hip_$name.so $tmp/*.o
rm -rf $tmp
echo built tools/variants/libnsp_hip_$name.so: $stem from $src "$@"
