#!/bin/bash
# stream-K on / off inside the step: kernel stats rows of the 8-phase fp32-output kernels (tile list and stream-K twins)
root=$(pwd); export TMPDIR=/tmp
for sk in 0 1; do
  d=$root/gpurun_out/stats_sk$sk; rm -rf $d
  (cd /tmp && NSP_GEMM_8P_STREAMK=$sk timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events > $d.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== NSP_GEMM_8P_STREAMK=$sk"
  python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('total %.1f ms' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6))
t = 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n = r['Name']
    if 'kk8p_kernel' in n and ('EpiSpec<0, 0, false, false' in n) and 'true>' not in n.split('EpiSpec')[1][:60].split('>')[0][-5:]:
        print('%-150s calls %5s total %8.2f ms avg %8.2f us' % (n[:150], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
        t += float(r['TotalDurationNs']) / 1e6
print('sum of these: %.2f ms' % t)
P
  rm -rf $d
done
