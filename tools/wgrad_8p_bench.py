#!/usr/bin/env python
"""Weight gradients dW[N, K] = dY[rows, N]^T X[rows, K] (incl. the split-K slab reduction): the 8-phase RR kernel
(NSP_GEMM_RR8P=1, default) against the round-3 kernels (128 x 128 LDS-DMA ring / register-staged / 256 x 256 two-stage),
arms interleaved in one process; plus a race screen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')


def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('%-34s %22s %22s' % ('dW[N,K] over rows', 'round-3 kernels', '8-phase RR'))
for rows in (25600, 51200, 102400):
    for (N, K) in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 512)):
        dy = torch.randn(rows, N, device=dev).bfloat16(); x = torch.randn(rows, K, device=dev).bfloat16()
        best = {}
        for r in range(3):
            for arm in ('0', '1'):
                os.environ['NSP_GEMM_RR8P'] = arm
                f = lambda: ops.linear_wgrad(dy, x)
                f(); f()
                best[arm] = min(best.get(arm, 1e30), timeit(f))
        fl = 2.0 * rows * N * K
        print('%-34s %12.1f (%6.0f) %12.1f (%6.0f)   x%.2f' % ('[%d,%d] x %d' % (N, K, rows), best['0'], fl / best['0'] / 1e6, best['1'], fl / best['1'] / 1e6, best['0'] / best['1']))
        del dy, x
rows, N, K = 3600000, 1024, 512
dy = torch.randn(rows, N, device=dev).bfloat16(); x = torch.randn(rows, K, device=dev).bfloat16()
best = {}
for r in range(2):
    for arm in ('0', '1'):
        os.environ['NSP_GEMM_RR8P'] = arm
        f = lambda: ops.linear_wgrad(dy, x)
        f()
        best[arm] = min(best.get(arm, 1e30), timeit(f, 3))
fl = 2.0 * rows * N * K
print('%-34s %12.1f (%6.0f) %12.1f (%6.0f)   x%.2f' % ('RNN-T out [%d,%d] x %d' % (N, K, rows), best['0'], fl / best['0'] / 1e6, best['1'], fl / best['1'] / 1e6, best['0'] / best['1']))
del dy, x
# race screen
bad = 0
for (rows, N, K) in ((25600, 2048, 512), (51201 // 8 * 8, 512, 512), (102400, 512, 2048)):
    dy = torch.randn(rows, N, device=dev).bfloat16(); x = torch.randn(rows, K, device=dev).bfloat16()
    os.environ['NSP_GEMM_RR8P'] = '0'
    ref = ops.linear_wgrad(dy, x)
    os.environ['NSP_GEMM_RR8P'] = '1'
    for r in range(100):
        out = ops.linear_wgrad(dy, x)
        if not ((out - ref).abs().max() <= 1e-4 * ref.abs().max()):
            bad += 1
print('race screen: 300 repetitions, %d mismatches' % bad)
