#!/bin/bash
# round 6: GEMM epilogue work -- device tests of the GEMM paths, per-shape A/B (tree against tools/variants/libnsp_hip_prev.so), whole-step A/B
root=$(pwd); out=$root/gpurun_out/$1; mkdir -p $out
python -m pytest tests/test_kernels_basic_gpu.py -q -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
ARMS="128x128,8p forced" python tools/gemm_8p_bench.py > $out/shapes_tree.log 2>&1
NSP_LIB_OVERRIDE=$root/tools/variants/libnsp_hip_prev.so ARMS="128x128,8p forced" python tools/gemm_8p_bench.py > $out/shapes_prev.log 2>&1
paste -d'|' <(grep -A12 "M = 102400" $out/shapes_prev.log | cut -c1-100) <(grep -A12 "M = 102400" $out/shapes_tree.log | cut -c61-100)
paste -d'|' <(grep -A12 "M = 25600" $out/shapes_prev.log | cut -c1-100) <(grep -A12 "M = 25600" $out/shapes_tree.log | cut -c61-100)
bash tools/r06_ab_bench.sh prev tree prev tree | tee $out/step_ab.log
