#!/bin/bash
# round 6, GPU session 3: staggered starts of the co-resident dK/dV workgroups
root=$(pwd); export TMPDIR=/tmp
out=$root/gpurun_out/r06s3; mkdir -p $out; rm -f $out/*.log
kstats() {
  d=$out/prof_$1; rm -rf $d
  (cd /tmp && NSP_LIB_OVERRIDE=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/tools/flash_bench.py 128 6 $3 > $d.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $1 T=$3" >> $out/kernel_split.log
  python - "$f" >> $out/kernel_split.log <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'flash' in n:
        print('%-40s calls %4s avg %9.1f us' % (n.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
P
  rm -rf $d
}
for T in 800 400; do
kstats tree "" $T
for a in "$@"; do kstats $a $root/tools/variants/libnsp_hip_$a.so $T; done
kstats tree "" $T
done
