#!/bin/bash
cd "$(dirname "$0")/.."
N=${1:-25}
S=gpurun_out/r05_ddp_variants5.log
: > $S
variant() {
  label=$1; shift
  env "$@" bash tools/r05_ddp_loop.sh $N > /dev/null
  echo "### $label ($*): $(tail -1 gpurun_out/r05_ddp_loop.log)" >> $S
  grep "ddp diag" gpurun_out/r05_ddp_loop_summary.log | grep -v print | grep -o "<conv#0 wrong[^']*'" | head -12 >> $S
  cp gpurun_out/r05_ddp_loop.log gpurun_out/r05_ddp_loop_$label.log
}
variant base5 X=1
cat $S
