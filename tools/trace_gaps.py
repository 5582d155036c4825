#!/usr/bin/env python
"""Idle time between consecutive kernels of each HIP stream in a rocprofv3 --kernel-trace CSV.

usage: trace_gaps.py <kernel_trace.csv> [max_gap_us=200]
For every queue: busy time, the sum of the gaps shorter than max_gap_us (longer ones are step boundaries / host stalls),
the gap histogram, and the kernels that most often sit BEHIND a gap."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
maxgap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
byq = collections.defaultdict(list)
for r in rows:
    q = r.get('Queue_Id') or r.get('Stream_Id') or '0'
    byq[q].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    busy = sum(e - s for s, e, _ in ks) / 1e3
    gaps, long_gaps, neg = [], 0.0, 0
    after = collections.Counter()
    aftert = collections.Counter()
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        g = (s1 - e0) / 1e3
        if g < 0:
            neg += 1
            continue
        if g > maxgap:
            long_gaps += g
            continue
        gaps.append(g)
        after[n1[:70]] += 1
        aftert[n1[:70]] += g
    if not gaps:
        continue
    gaps.sort()
    print('queue %s: %d kernels, busy %.1f ms, short gaps %.1f ms (median %.2f us, p90 %.2f us), long gaps %.1f ms, overlapping starts %d' % (
        q, len(ks), busy / 1e3, sum(gaps) / 1e3, gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], long_gaps / 1e3, neg))
    hist = collections.Counter(min(int(g), 20) for g in gaps)
    print('   gap histogram (us: count): ' + ' '.join('%d:%d' % (k, hist[k]) for k in sorted(hist)))
    for n, t in aftert.most_common(8):
        print('   %8.1f us of gaps in front of %4d x %s' % (t, after[n], n))
