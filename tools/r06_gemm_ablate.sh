#!/bin/bash
# round 6: what bounds the GEMM epilogues -- the per-shape bench under each -DNSP_EPI_ABLATE variant library (tools/variants)
root=$(pwd); out=$root/gpurun_out/$1; mkdir -p $out; shift
for v in "$@"; do
  lib=""; [ "$v" != "tree" ] && lib=$root/tools/variants/libnsp_hip_$v.so
  NSP_LIB_OVERRIDE=$lib ARMS="128x128,8p forced" python tools/gemm_8p_bench.py > $out/shapes_$v.log 2>&1
  echo "== $v"; grep -A11 "M = 102400" $out/shapes_$v.log | tail -10 | cut -c1-100
done
