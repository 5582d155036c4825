#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/r05_conv_variant2.log
: > $L
P="python tools/conv_first_kernel_stress.py"
echo "=== neighbour: pure torch (library GEMMs + elementwise)" >> $L
timeout 400 $P --load-seconds 60 --load-kind torch >> $L 2>&1 &
LP=$!
sleep 10
echo "--- tree library" >> $L
timeout 100 $P --raw --probe 3000 --churn --dump 2>&1 | grep -v "wrong elements" >> $L
echo "--- conv2d.hip without packed fp32 instructions" >> $L
NSP_LIB_OVERRIDE=tools/probe/r05/libnsp_hip_nopk.so timeout 100 $P --raw --probe 10000 --churn --dump 2>&1 | grep -v "wrong elements" >> $L
echo "--- conv2d.hip at -O1" >> $L
NSP_LIB_OVERRIDE=tools/probe/r05/libnsp_hip_o1.so timeout 100 $P --raw --probe 10000 --churn --dump 2>&1 | grep -v "wrong elements" >> $L
echo "--- tree library again" >> $L
timeout 100 $P --raw --probe 3000 --churn 2>&1 | grep -v "wrong elements" >> $L
wait $LP
grep -v "amdgpu.ids" $L
