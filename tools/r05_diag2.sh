#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r05_diag2.log
: > $L
run() { echo "=== $*" >> $L; timeout 600 "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
run python -m pytest tests/test_alignment_e2e_gpu.py tests/test_ddp_hip_gpu.py -x -q -m gpu -s
run python -m pytest tests/test_hostile_neighbour_gpu.py -q -m gpu -s
grep -E "^===|rc=|passed|failed|worst|Error|hostile|neighbour|pattern" $L | cut -c1-600 > gpurun_out/r05_diag2_summary.log
