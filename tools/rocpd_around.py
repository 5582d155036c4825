#!/usr/bin/env python
"""Timeline around one kernel of a rocprofv3 rocpd trace: every dispatch that overlaps or directly neighbours the
n-th occurrence of <pattern>, with start / end relative to that kernel's start and the queue / stream it ran on.
usage: python tools/rocpd_around.py <results.db> <pattern> [occurrence=2] [before=4] [after=6]"""
import sqlite3
import sys


def main(path, pat, occ=2, before=4, after=6):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    extra = [x for x in ('queue_id', 'stream_id', 'grid_x', 'workgroup_x', 'lds_size', 'grid_size', 'workgroup_size', 'lds_block_size') if x in cols]
    rows = c.execute("select %s, start, end %s from kernels order by start" % (namecol, ''.join(', ' + x for x in extra))).fetchall()
    hits = [i for i, r in enumerate(rows) if pat in r[0]]
    if not hits:
        print('no kernel matches', pat, '| columns:', cols)
        return
    i = hits[min(occ, len(hits) - 1)]
    t0, t1 = rows[i][1], rows[i][2]
    print('columns:', extra)
    for j, r in enumerate(rows):
        near = i - before <= j <= i + after
        overlap = r[2] > t0 and r[1] < t1
        if near or overlap:
            print('%s %-70s start %+10.1f us  end %+10.1f us  dur %9.1f us  %s' % (
                '>>' if j == i else '  ', r[0][:70], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3:]))


if __name__ == '__main__':
    a = sys.argv
    main(a[1], a[2], *[int(x) for x in a[3:]])
