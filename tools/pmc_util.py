#!/usr/bin/env python
"""Per-kernel MFMA / VALU utilisation and wave-stall breakdown from ONE rocprofv3 PMC pass (csv output).
usage: python tools/pmc_util.py <counter_collection.csv> [n_cu]
Counters expected (any subset): SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE.
Derived with the gfx94x formulas of rocprofiler's derived_counters.xml (ROCm 7.2 ships none for gfx950,
MI355X_MICROARCH.md): MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CU * 4),
VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / (CU * 4) / GRBM_GUI_ACTIVE; stall shares are fractions of
SQ_WAVE_CYCLES; LDS conflict rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import re
import sys


def main(path, ncu):
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
            a = agg.setdefault(name, {'n': collections.Counter(), 'v': collections.Counter(), 'us': 0.0, 'd': set()})
            a['v'][r['Counter_Name']] += float(r['Counter_Value'])
            did = r.get('Dispatch_Id', r.get('Correlation_Id'))
            if did not in a['d']:
                a['d'].add(did)
                a['us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    rows = []
    for k, a in agg.items():
        v, n = a['v'], max(1, len(a['d']))
        gui = v.get('GRBM_GUI_ACTIVE', 0.0)
        wc = v.get('SQ_WAVE_CYCLES', 0.0)
        rows.append((a['us'], k, n,
                     100.0 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * ncu * 4) if gui else float('nan'),
                     100.0 * v.get('SQ_ACTIVE_INST_VALU', 0.0) * 4 / (ncu * 4) / gui if gui else float('nan'),
                     100.0 * v.get('SQ_WAIT_ANY', 0.0) / wc if wc else float('nan'),
                     100.0 * v.get('SQ_WAIT_INST_ANY', 0.0) / wc if wc else float('nan'),
                     100.0 * v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'] if v.get('SQ_LDS_IDX_ACTIVE') else float('nan')))
    rows.sort(reverse=True)
    print('%-52s %6s %10s %9s %9s %10s %12s %9s' % ('kernel', 'calls', 'total ms', 'MfmaUtil%', 'VALUBusy%', 'WAIT_ANY%', 'WAIT_INST%', 'LDSconf%'))
    for us, k, n, mf, vb, wa, wi, lc in rows[:40]:
        print('%-52s %6d %10.2f %9.1f %9.1f %10.1f %12.1f %9.1f' % (k[:52], n, us / 1e3, mf, vb, wa, wi, lc))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256)
