#!/bin/bash
# L2 hit rate and fabric reads of the persistent LSTM kernels (B = $LB, default 64): separate rocprofv3 --pmc passes
root=$(pwd); export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum"; do
  tagc=$(echo $c | tr ' ' '_')
  out=$root/gpurun_out/pmc_lstm_$tagc
  rm -rf $out; mkdir -p $out
  (cd /tmp && LB=${LB:-64} rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python $root/tools/lstm_stack_bench.py > $out/run.out 2>&1)
  f=$(find $out -name '*counter_collection.csv' | head -1)
  echo "== $c"
  python - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'persistent' in r['Kernel_Name']:
        k = ('fwd' if 'fwd' in r['Kernel_Name'] else 'bwd', r['Counter_Name'])
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items()):
    print('%s %-24s launches %3d  mean per launch %.4g' % (k[0], k[1], n, v / n))
P
  rm -rf $out
done
