#!/bin/bash
# stream-K in the step: the tree library with NSP_GEMM_8P_STREAMK = 0 / 1, alternating, plus a variant library if given
root=$(pwd)
for arm in "$@"; do
  lib=""; sk=${arm##*:}; v=${arm%%:*}; [ "$v" != "tree" ] && lib=$root/tools/variants/libnsp_hip_$v.so
  NSP_GEMM_8P_STREAMK=$sk NSP_LIB_OVERRIDE=$lib python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-b16 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s ms_per_step %7.2f  frames/s %.0f' % ('$arm', d['ms_per_step'], d['value']))"
done
