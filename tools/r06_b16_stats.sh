#!/bin/bash
# kernel stats of the 16-utterance step (SURVEY 8d's per-GPU batch)
root=$(pwd); export TMPDIR=/tmp
out=$root/gpurun_out/r06b16; mkdir -p $out
d=$out/prof; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/bench.py --batch 16 --steps 8 --warmup 4 --no-cpu-baseline --no-b16 --no-kernel-events > $d.out 2>&1)
tail -1 $d.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=16 under rocprof: ms_per_step', d['ms_per_step'])"
f=$(find $d -name '*kernel_stats.csv' | head -1)
python - "$f" > $out/kernel_stats_b16.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = sum(int(r['Calls']) for r in rows)
print('# bench.py --batch 16 --steps 8 --warmup 4 (12 steps): total kernel time %.1f ms = %.2f ms/step over %d launches = %.0f/step' % (tot / 1e6, tot / 1e6 / 12, n, n / 12))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    print('%-100s calls/step %6.1f  ms/step %7.3f avg %8.2f us' % (r['Name'][:100], int(r['Calls']) / 12, float(r['TotalDurationNs']) / 1e6 / 12, float(r['AverageNs']) / 1e3))
P
rm -rf $d; head -32 $out/kernel_stats_b16.txt | cut -c1-60,100-170
