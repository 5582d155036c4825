#!/usr/bin/env python3
"""Per-basic-block instruction census of a gfx950 `.s` file (hipcc -S --cuda-device-only).

usage: isa_blocks.py file.s [kernel-substring] [min_instructions]
Prints, for every basic block of the selected kernels with at least `min_instructions` instructions, the counts of
VALU / transcendental / quarter-rate integer multiplies / MFMA / LDS / VMEM / SALU / waits / branches, so that the hot
block of a kernel (the one with the MFMAs and exponentials) can be read at a glance.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')):
        return 'trans'
    if op.startswith(('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_hi_i32', 'v_mad_u64', 'v_mad_i64')):
        return 'qmul'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    kernel = None
    block = None
    counts = Counter()
    ops = Counter()
    n = 0

    def flush():
        nonlocal counts, n, ops
        if kernel and want in kernel and n >= minins:
            keys = ['valu', 'trans', 'qmul', 'mfma', 'lds', 'vmem', 'salu', 'wait', 'barrier', 'branch', 'nop']
            print('%-28s n=%4d  ' % (block, n) + ' '.join('%s=%d' % (k, counts[k]) for k in keys if counts[k]))
            if '-v' in sys.argv:
                print('      ' + ' '.join('%s:%d' % kv for kv in ops.most_common(24)))
        counts = Counter()
        ops = Counter()
        n = 0

    for line in open(path):
        s = line.strip()
        if not s or s.startswith((';', '//', '.')) and not s.startswith('.LBB'):
            continue
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', s)
        if m:
            flush()
            lab = m.group(1)
            if not lab.startswith('.L'):
                kernel = lab
                if want in kernel:
                    print('== ' + kernel[:100])
            block = lab
            continue
        op = s.split()[0]
        if op.startswith('.'):
            continue
        counts[classify(op)] += 1
        ops[op] += 1
        n += 1
    flush()


if __name__ == '__main__':
    main()
