#!/usr/bin/env python
"""Per-training-step stream overlap from a rocprofv3 rocpd trace: wall, busy union, time only the
side stream (prediction network) is running, and the span of the LSTM forward / backward chains.
usage: python tools/trace_overlap.py <results.db>"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, ce = 0, None
    out = []
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def subtract(a, b):
    """time in union a not covered by union b"""
    res = 0
    j = 0
    for s, e in a:
        cur = s
        for bs, be in b:
            if be <= cur or bs >= e:
                continue
            if bs > cur:
                res += bs - cur
            cur = max(cur, be)
            if cur >= e:
                break
        if cur < e:
            res += e - cur
    return res


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select start,end,stream_id,name from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r[3]]
    groups, cur = [], [adam[0]]
    for a, b in zip(adam, adam[1:]):
        if rows[b][0] - rows[a][1] > 5e6:
            groups.append(cur); cur = [b]
        else:
            cur.append(b)
    groups.append(cur)
    main_stream = max(set(r[2] for r in rows), key=lambda s: sum(1 for r in rows if r[2] == s))
    for gi in range(1, len(groups)):
        lo = rows[groups[gi - 1][-1]][1]; hi = rows[groups[gi][-1]][1]
        ks = [r for r in rows if r[0] >= lo and r[1] <= hi]
        um = union([(r[0], r[1]) for r in ks if r[2] == main_stream])
        us = union([(r[0], r[1]) for r in ks if r[2] != main_stream])
        ua = union([(r[0], r[1]) for r in ks])
        lf = [r for r in ks if 'lstm_step_fwd' in r[3]]
        lb = [r for r in ks if 'lstm_step_bwd' in r[3]]
        print('step %d: wall %.2f busy %.2f main %.2f side %.2f side-only %.2f | lstm fwd span %.2f (sum %.2f) bwd span %.2f (sum %.2f) ms'
              % (gi, (hi - lo) / 1e6, length(ua) / 1e6, length(um) / 1e6, length(us) / 1e6, subtract(us, um) / 1e6,
                 (lf[-1][1] - lf[0][0]) / 1e6 if lf else 0, sum(r[1] - r[0] for r in lf) / 1e6,
                 (lb[-1][1] - lb[0][0]) / 1e6 if lb else 0, sum(r[1] - r[0] for r in lb) / 1e6))


if __name__ == '__main__':
    main(sys.argv[1])
