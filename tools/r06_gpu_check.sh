#!/bin/bash
# round 6: one GPU call = selected device tests + a short bench line (+ optional kernel stats); usage: r06_gpu_check.sh <tag> "<pytest args>" [stats]
root=$(pwd); export TMPDIR=/tmp
tag=$1; out=$root/gpurun_out/$tag; mkdir -p $out
if [ -n "$2" ]; then python -m pytest $2 -q -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log; fi
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || tail -5 $out/bench.err
python - $out/bench.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print('no bench line', e); sys.exit(0)
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'mfu', d['encoder_mfu'], 'roof', d['roofline']['frac'] if d.get('roofline') else None)
print('phases', d['config']['phases'])
for c in (d.get('roofline') or {}).get('classes', []):
    print('  %-22s %6.2f ms/step  %s %s frac %s' % (c['class'], c['ms_per_step'], c['achieved'], c['unit'], c['frac']))
print('also', [(a['per_gpu_batch'], a['ms_per_step'], a['value']) for a in d.get('also') or []])
P
if [ "$3" = "stats" ]; then
  d=$out/prof; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events > $d.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" > $out/kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events (6 steps); total %.1f ms' % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:70]:
    print('%-110s calls %6s total %9.3f ms avg %9.2f us %5.2f%%' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
P
  rm -rf $d; head -45 $out/kernel_stats.txt
fi
