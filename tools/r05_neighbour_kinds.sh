#!/bin/bash
# round 5 (diagnosis only): which kind of second tenant triggers the packed-fp32 failure of the first conv layer?
# needs tools/probe/r05/libnsp_hip_packed.so = the tree's library with conv2d.hip compiled WITH packed fp32 ops
# (tools/make_variant_lib.sh packed conv2d.hip; move the result from tools/probe/ab, which gpurun does not ship)
cd "$(dirname "$0")/.."
L=gpurun_out/r05_neighbour_kinds.log
: > $L
for kind in gemm elementwise softmax; do
  echo "=== neighbour: $kind only" >> $L
  timeout 100 python tests/gpu_neighbour.py 22 $kind >> $L 2>&1 &
  LP=$!
  sleep 9
  NSP_LIB_OVERRIDE=tools/probe/r05/libnsp_hip_packed.so timeout 60 python tools/conv_first_kernel_stress.py --raw --probe 8000 --churn 2>&1 | grep -v "wrong elements" >> $L
  wait $LP
done
grep -v "amdgpu.ids" $L
