#!/bin/bash
# round-5 diagnosis of the stock-DDP failure: run on the GPU box, writes gpurun_out/r05_diag.log
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_diag.log
: > $L
run() { echo "=== $*" >> $L; timeout 300 "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
run python tools/ddp_repro.py --hooks
run python tools/ddp_repro.py --single-stream --hooks
run python tools/ddp_repro.py --single-stream --poison --hooks
run python tools/ddp_repro.py --single-stream --no-refresh --hooks
run python tools/ddp_repro.py --single-stream --rank 0 --warm-full
run python tools/ddp_repro.py --poison --hooks
NSP_FLASH_DKV_HALVES=0 run python tools/ddp_repro.py --single-stream
NSP_LN_PREP=0 run python tools/ddp_repro.py --single-stream
NSP_LN_SKIP32=0 run python tools/ddp_repro.py --single-stream
NSP_LINEAR_GLU=0 run python tools/ddp_repro.py --single-stream
NSP_LSTM_PERSISTENT=0 run python tools/ddp_repro.py --single-stream
for i in 1 2 3 4; do
  run python -m pytest tests/test_ddp_hip_gpu.py -q -x -k "stock" -s
done
NSP_POISON=1 run python -m pytest tests/test_ddp_hip_gpu.py -q -k "two_rank" -s
grep -E "^===|RESULT|^it |rc=|passed|failed|worst per-tensor|Error" $L > gpurun_out/r05_diag_summary.log
