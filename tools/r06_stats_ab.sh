#!/bin/bash
# kernel stats of the step under two libraries (variant names or "tree"): the rows of the kernels named in $KERNELS
root=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
  lib=""; [ "$v" != "tree" ] && lib=$root/tools/variants/libnsp_hip_$v.so
  d=$root/gpurun_out/stats_$v; rm -rf $d
  (cd /tmp && NSP_LIB_OVERRIDE=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events > $d.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python - "$f" <<'P'
import csv, sys, os
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total %.1f ms' % (tot / 1e6))
want = os.environ.get('KERNELS', 'splitk_reduce,EpiSpec<0, 0, false, false, false, fa,colsum,kk_glds_kernel<0>').split(',')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if any(w in r['Name'] for w in want):
        print('%-100s calls %5s total %8.2f ms avg %8.2f us' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
P
  rm -rf $d
done
