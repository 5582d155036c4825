#!/bin/bash
# round 5: the stock-DDP two-rank test in a loop (N runs x NSP_DDP_ITERS iterations per rank), diagnostics on
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-30}
L=gpurun_out/r05_ddp_loop.log
: > $L
export NSP_DDP_ITERS=${NSP_DDP_ITERS:-12} NSP_DDP_DIAG=${NSP_DDP_DIAG:-1}
fails=0
for i in $(seq 1 $N); do
  echo "=== run $i" >> $L
  timeout 300 python -m pytest tests/test_ddp_hip_gpu.py -q -x -k "stock" -s >> $L 2>&1
  rc=$?
  echo "rc=$rc" >> $L
  [ $rc -ne 0 ] && fails=$((fails+1))
done
echo "runs $N failures $fails (NSP_DDP_ITERS=$NSP_DDP_ITERS NSP_DDP_DIAG=$NSP_DDP_DIAG)" >> $L
grep -E "^=== run|rc=|worst per-tensor|ddp diag|AssertionError|failures" $L | cut -c1-1500 > gpurun_out/r05_ddp_loop_summary.log
tail -1 $L
