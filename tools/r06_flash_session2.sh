#!/bin/bash
# round 6, GPU session 2: keep-bits build -- device flash tests, bench, per-kernel split
root=$(pwd); export TMPDIR=/tmp
out=$root/gpurun_out/r06s2; mkdir -p $out
python -m pytest tests/test_flash_attn_gpu.py -x -q > $out/pytest_flash.log 2>&1
python tools/flash_bench.py 128 10 > $out/flash_tree.log 2>&1
d=$out/prof; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/tools/flash_bench.py 128 6 800 > $d.out 2>&1)
f=$(find $d -name '*kernel_stats.csv' | head -1)
python - "$f" > $out/kernel_split.log <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'flash' in n:
        print('%-40s calls %4s avg %9.1f us' % (n.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
P
rm -rf $d
