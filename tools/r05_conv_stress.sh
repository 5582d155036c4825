#!/bin/bash
# round 5: the reproducer of the packed-fp32 failure -- the first conv layer (and a pure-torch GEMM
# as the control) launched thousands of times beside a second process; on the GPU box, from the repo root.
# With the shipped library (built without packed fp32 ops) every line reports 0 wrong launches; to see the failure build
# conv2d.hip without `-Xclang -target-feature -Xclang -packed-fp32-ops` (tools/make_variant_lib.sh) and pass the result
# through NSP_LIB_OVERRIDE.  profiles/r05_packed_fp32_probe_matrix.log / r05_packed_fp32_variant_libs.log hold the runs.
cd "$(dirname "$0")/.."
L=gpurun_out/r05_conv_matrix.log
: > $L
P="python tools/conv_first_kernel_stress.py"
echo "=== neighbour: pure torch (library GEMMs + elementwise)" >> $L
timeout 400 python tests/gpu_neighbour.py 60 >> $L 2>&1 &
LP=$!
sleep 10
timeout 100 $P --raw --probe 20000 --churn --dump 2>&1 | grep -v "wrong elements" >> $L
timeout 100 $P --kind mm --probe 20000 --churn >> $L 2>&1
wait $LP
echo "=== neighbour: XS transducer steps of this repository" >> $L
timeout 400 $P --load-seconds 60 >> $L 2>&1 &
LP=$!
sleep 12
timeout 100 $P --kind mm --probe 20000 --churn >> $L 2>&1
timeout 100 $P --raw --probe 20000 --churn --h2d --dump 2>&1 | grep -v "wrong elements" >> $L
wait $LP
grep -v "amdgpu.ids" $L
