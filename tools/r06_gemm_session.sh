#!/bin/bash
# round 6: GEMM epilogue work -- device tests of the GEMM paths, per-shape bench of the tree and of variant libraries, whole-step A/B
# usage: r06_gemm_session.sh <tag> <variant> ...   ("tree" = the library in the tree)
root=$(pwd); out=$root/gpurun_out/$1; mkdir -p $out; shift
python -m pytest tests/test_kernels_basic_gpu.py tests/test_variants_gpu.py -q -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for v in "$@"; do
  lib=""; [ "$v" != "tree" ] && lib=$root/tools/variants/libnsp_hip_$v.so
  NSP_LIB_OVERRIDE=$lib ARMS="128x128,8p forced" python tools/gemm_8p_bench.py > $out/shapes_$v.log 2>&1
  echo "== $v"; grep -A11 "M = 102400" $out/shapes_$v.log | tail -10 | cut -c1-100; grep -A11 "M = 25600" $out/shapes_$v.log | tail -10 | cut -c1-100
done
bash tools/r06_ab_bench.sh "$@" "$@" | tee $out/step_ab.log
