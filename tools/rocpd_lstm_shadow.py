#!/usr/bin/env python
"""How much slower do the main stream's kernels run while a persistent LSTM kernel holds half of the CUs?
For every (kernel name, grid) class of a rocprofv3 rocpd trace: launches, mean duration of the launches that lie
entirely OUTSIDE every persistent-LSTM interval, mean duration of those entirely INSIDE one, and the ratio; then
the time the inside launches would have taken at the outside rate (the cost of sharing the device).
usage: python tools/rocpd_lstm_shadow.py <results.db> [pattern=persistent]"""
import bisect
import sqlite3
import sys
from collections import defaultdict


def main(path, pat='persistent'):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    gridcol = 'grid_x' if 'grid_x' in cols else ('grid_size' if 'grid_size' in cols else None)
    q = "select %s, start, end%s from kernels order by start" % (namecol, (', ' + gridcol) if gridcol else '')
    rows = c.execute(q).fetchall()
    shadow = sorted((r[1], r[2]) for r in rows if pat in r[0])
    if not shadow:
        print('no kernel matches', pat)
        return
    starts = [s for s, _ in shadow]

    def where(a, b):
        """'in' if [a,b] lies inside one shadow interval, 'out' if it touches none, else 'mixed'"""
        i = bisect.bisect_right(starts, a) - 1
        if i >= 0 and shadow[i][0] <= a and b <= shadow[i][1]:
            return 'in'
        for s, e in shadow[max(i, 0):i + 3]:
            if e > a and s < b:
                return 'mixed'
        return 'out'

    acc = defaultdict(lambda: {'in': [], 'out': [], 'mixed': []})
    for r in rows:
        if pat in r[0]:
            continue
        acc[(r[0], r[3] if gridcol else 0)][where(r[1], r[2])].append((r[2] - r[1]) / 1e3)
    tot_in = tot_in_ideal = 0.0
    lines = []
    for (name, grid), d in acc.items():
        if len(d['in']) < 2 or len(d['out']) < 2:
            continue
        mi, mo = sum(d['in']) / len(d['in']), sum(d['out']) / len(d['out'])
        tot_in += sum(d['in'])
        tot_in_ideal += mo * len(d['in'])
        lines.append((sum(d['in']) - mo * len(d['in']), name[:60], grid, len(d['out']), mo, len(d['in']), mi, mi / mo))
    lines.sort(reverse=True)
    print('shadow = %d launches of *%s*, %.1f ms in total' % (len(shadow), pat, sum(e - s for s, e in shadow) / 1e6))
    print('%-60s %10s | %5s %9s | %5s %9s | %5s | %9s' % ('kernel', 'grid', 'n out', 'mean us', 'n in', 'mean us', 'ratio', 'excess ms'))
    for ex, name, grid, no, mo, ni, mi, ratio in lines[:40]:
        print('%-60s %10d | %5d %9.1f | %5d %9.1f | %5.2f | %9.2f' % (name, grid, no, mo, ni, mi, ratio, ex / 1e3))
    print('classes seen both ways: time inside the shadow %.1f ms, at the outside rate %.1f ms -> excess %.1f ms over the whole trace'
          % (tot_in / 1e3, tot_in_ideal / 1e3, (tot_in - tot_in_ideal) / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:])
