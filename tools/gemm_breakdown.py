#!/usr/bin/env python
"""Per-shape breakdown of GEMM time in one bench step (HIP events around every nsp_gemm)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from neural_sp_amd import ops  # noqa: E402
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch  # noqa: E402
from neural_sp_amd.speech2text import Speech2Text  # noqa: E402

rec = []
orig = ops._gemm_raw_untimed


def timed(M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, **k)
    e1.record()
    batch = k.get('batch', (1, 1))
    lay = ('K' if a_cs == 1 else 'R') + ('K' if b_ks == 1 else 'R')
    lay += ':' + ''.join(f for f, t in (('b', k.get('bias')), ('p', k.get('pre_out')), ('d', k.get('dact_src')), ('r', k.get('res')),
                                        ('s', k.get('colsum_slabs'))) if t is not None)
    lay += ('o16' if C.dtype == torch.bfloat16 else 'o32') + ('' if torch.cuda.current_stream() == torch.cuda.default_stream() else '*')
    rec.append(((M, N, K, batch[0] * batch[1], lay, k.get('splitk', 1)), e0, e1))


def timed_joint(fn, M, Vp, J, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    rec.append(((M, Vp, J, 1, 'J%d' % args[0], 1), e0, e1))
    return rc


ops.gemm_raw = timed
ops.rnnt_joint_gemm_timed = timed_joint
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ops.set_compute_mode(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
batch = synthetic_batch(B=BATCH, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
for it in range(2):
    rec.clear()
    loss, _ = model(batch, task='all')
    loss.backward()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print('total gemm ms %.2f over %d launches' % (tot, len(rec)))
print('%8s %6s %6s %5s %12s %3s %5s %9s %9s %8s' % ('M', 'N', 'K', 'batch', 'lay', 'sk', 'calls', 'total_ms', 'avg_us', 'TFLOP/s'))
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, b, lay, sk = key
    tf = 2.0 * M * N * K * b * n / (ms * 1e-3) / 1e12
    print('%8d %6d %6d %5d %12s %3d %5d %9.3f %9.1f %8.1f' % (M, N, K, b, lay, sk, n, ms, ms * 1e3 / n, tf))
