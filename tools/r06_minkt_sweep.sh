#!/bin/bash
# weight-gradient split planning: k-tiles per split at least NSP_WGRAD_MIN_KT -- whole-step sweep at 16 / 64 / 128 utterances per GPU
for b in 16 64 128; do
  for kt in 12 8 6 4 12 8 6 4; do
    NSP_WGRAD_MIN_KT=$kt python bench.py --batch $b --steps 12 --warmup 4 --no-cpu-baseline --no-b16 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %3d min_kt %2d  ms_per_step %7.2f  frames/s %.0f' % ($b, $kt, d['ms_per_step'], d['value']))"
  done
done
