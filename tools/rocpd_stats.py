#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / %.
usage: python tools/rocpd_stats.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name[:110]


def main(path, top=40):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows)
    print('total kernel time: %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))
    print('%-110s %7s %11s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    for r in rows[:top]:
        print('%-110s %7d %11.3f %10.2f %10.2f %10.2f %6.2f' % (short(r[0]), r[1], r[2] / 1e6, r[3] / 1e3,
                                                                r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
