#!/bin/bash
# round 6: kernel trace of 3 steps -> idle time between kernels per stream (tools/trace_gaps.py)
root=$(pwd); out=$root/gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
d=$out/prof; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b16 --no-kernel-events > $d.out 2>&1)
f=$(find $d -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f 200 | tee $out/gaps.txt
head -2 $f > $out/trace_head.csv
rm -rf $d
