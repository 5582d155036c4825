#!/bin/bash
# A/B of whole-step time between library builds in ONE call: usage r06_ab_bench.sh <variant name or "tree"> ...
root=$(pwd)
for v in "$@"; do
  lib=""; [ "$v" != "tree" ] && lib=$root/tools/variants/libnsp_hip_$v.so
  NSP_LIB_OVERRIDE=$lib python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-b16 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s ms_per_step %7.2f  frames/s %.0f' % ('$v', d['ms_per_step'], d['value']))"
done
