#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output).
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section) -> doubled here.  WRITE_SIZE is calibrated against the
bf16 cast kernel, whose written bytes are known from its grid (see --calib).
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json revision per_gpu_batch]
With out.json: also writes the GEMM-class aggregate (all gemm_bf16_* kernels) that bench.py quotes as roofline.traffic."""
import collections
import csv
import re
import sys


def load(path, counter):
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
            name = name.split('(')[0]
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
            a[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    return agg


def main(fp, wp):
    fe, wr = load(fp, 'FETCH_SIZE'), load(wp, 'WRITE_SIZE')
    rows = []
    for k in fe:
        n, kib, us = fe[k]
        wk = wr.get(k, [0, 0.0, 0.0])
        rd = 2.0 * kib * 1024 / n                 # gfx950 correction: x2
        wrb = wk[1] * 1024 / max(1, wk[0])
        rows.append((rd * n + wrb * n, k, n, rd, wrb, us / n))
    rows.sort(reverse=True)
    print('%-48s %6s %12s %12s %10s %9s' % ('kernel', 'calls', 'read MB/call', 'write MB/call', 'avg us', 'GB/s'))
    for tot, k, n, rd, wrb, us in rows[:40]:
        print('%-48s %6d %12.3f %12.3f %10.1f %9.0f' % (k[:48], n, rd / 1e6, wrb / 1e6, us, (rd + wrb) / us / 1e3))


    return rows


if __name__ == '__main__':
    rows = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:
        import json
        BATCH = int(sys.argv[5]) if len(sys.argv) > 5 else 128
        g = [r for r in rows if r[1].startswith('gemm_bf16') or 'rnnt_joint_rows_kernel' in r[1]]   # (the joint's logit passes: node-stationary kernel)
        n = sum(r[2] for r in g)
        rd = sum(r[3] * r[2] for r in g) / n
        wr = sum(r[4] * r[2] for r in g) / n
        alg = None
        if len(sys.argv) > 6:
            # NSP_GEMM_DEBUG=1 dump of the SAME command: one line per bf16 GEMM launch with its algorithmic HBM bytes
            vals = [int(l.rsplit('algbytes', 1)[1]) for l in open(sys.argv[6]) if 'algbytes' in l]
            alg = {'launches': len(vals), 'bytes_per_launch': sum(vals) / max(1, len(vals))}
        # the other kernel classes bench.py times (roofline.classes): counter bytes per STEP (the trace holds `steps` steps)
        steps = 2
        CLASSES = {'flash_fwd': ('flash_fwd_kernel',), 'flash_bwd': ('flash_bwd_dq_kernel', 'flash_bwd_dkv_kernel'),
                   'layernorm_fwd': ('ln_fwd_kernel', 'ln_pair_fwd_kernel'), 'layernorm_bwd': ('ln_bwd_kernel', 'ln_pair_bwd_kernel'),
                   'conv_frontend_fwd+dgrad': ('conv3x3_c1_kernel', 'conv3x3_c32_b16_kernel', 'conv3x3_c32_kernel')}
        classes = {}
        for cname, pats in CLASSES.items():
            sel = [r for r in rows if any(pt in r[1] for pt in pats)]      # (substring: anonymous-namespace kernels come back mangled)
            if not sel:
                continue
            nl = sum(r[2] for r in sel)
            by = sum((r[3] + r[4]) * r[2] for r in sel)
            us = sum(r[5] * r[2] for r in sel)
            classes[cname] = {'launches_per_step': nl / steps, 'hbm_bytes_per_step': by / steps, 'kernel_us_per_step': round(us / steps, 1),
                              'hbm_tb_per_s_while_running': round(by / us / 1e6, 3) if us > 0 else None,
                              'kernels': sorted(set(r[1] for r in sel))}
        json.dump({'classes': classes,
                   'classes_note': 'counter bytes (FETCH_SIZE x2 + WRITE_SIZE) of every launch of the named kernels in the traced steps / steps; '
                                   'conv3x3 kernels serve the forward AND the data gradient, bench.py times only the forward calls',
                   'algorithmic_bytes_per_launch': alg['bytes_per_launch'] if alg else None,
                   'algorithmic_launches': alg['launches'] if alg else None,
                   'algorithmic_definition': 'both operands once + every output image (split-K slabs included) + every side operand once, '
                                             'summed over all bf16 GEMM launches of the traced steps (NSP_GEMM_DEBUG=1 pass of the same command) / launches',
                   'kernel_class': 'bf16 GEMM class (all gemm_bf16_* kernels of libnsp_hip.so + rnnt_joint_rows_kernel, the RNN-T joint logit passes)',
                   'workload': 'bench.py default (Conformer-L, per-GPU batch %d, bf16), 2 steps in the trace' % BATCH,
                   'per_gpu_batch': BATCH,
                   'launches': n, 'hbm_bytes_per_launch': rd + wr, 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                   'per_kernel': {r[1]: {'launches': r[2], 'read_MB_per_launch': round(r[3] / 1e6, 3),
                                         'write_MB_per_launch': round(r[4] / 1e6, 3), 'avg_us': round(r[5], 1)} for r in g},
                   'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel trace only) over '
                             'bench.py --steps 1 --warmup 1; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section)',
                   'revision': sys.argv[4] if len(sys.argv) > 4 else 'unknown'}, open(sys.argv[3], 'w'), indent=1)
