#!/bin/bash
# round 5: after the library was rebuilt without packed fp32 ops -- the reproducers, the device suite, the bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_verify.log
: > $L
P="python tools/conv_first_kernel_stress.py"
echo "=== first conv layer beside a process running library GEMMs (round-4 build: 36 % of the launches wrong)" >> $L
timeout 100 python tests/gpu_neighbour.py 30 >> $L 2>&1 &
LP=$!
sleep 10
timeout 100 $P --raw --probe 20000 --churn --dump 2>&1 | grep -v "wrong elements" >> $L
wait $LP
echo "=== stock-DDP two-rank test, 15 runs x 12 iterations per rank (round-4 build: 4-6 failures in 25)" >> $L
NSP_DDP_DIAG=0 NSP_DDP_ITERS=12 bash tools/r05_ddp_loop.sh 15 >> $L 2>&1
echo "=== device suite" >> $L
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r05_pytest_gpu_full.log 2>&1
echo "rc=$?" >> gpurun_out/r05_pytest_gpu_full.log
tail -5 gpurun_out/r05_pytest_gpu_full.log >> $L
echo "=== bench" >> $L
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_stderr.log
tail -c 3000 gpurun_out/r05_bench_line.json >> $L
grep -v "amdgpu.ids" $L | cut -c1-3000
