#!/usr/bin/env python
"""A/B of the weight-gradient paths at the training step's shapes, interleaved in one process:
NSP_GEMM_RR256=0 (128 x 128 kernels: LDS-DMA ring / register-staged) vs 1 (gemm_bf16_rr256_kernel), each incl. its slab
reduction.  dW[N, K] = dY[rows, N]^T X[rows, K]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
shapes = []
for rows in (102400, 51200, 25600):
    shapes += [(rows, 2048, 512), (rows, 512, 2048), (rows, 512, 512), (rows, 1536, 512), (rows, 1024, 512)]
shapes += [(3600007, 1000, 512), (102400, 512, 1280)]
only = os.environ.get('WROWS')
for rows, N, K in shapes:
    if only and str(rows) not in only.split(','):
        continue
    dy = torch.randn(rows, N, device='cuda').bfloat16()
    x = torch.randn(rows, K, device='cuda').bfloat16()
    res = {}
    for rnd in range(3):
        for v in ('0', '1'):
            os.environ['NSP_GEMM_RR256'] = v
            ops.linear_wgrad(dy, x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dw = ops.linear_wgrad(dy, x)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 200)
    fl = 2.0 * rows * N * K
    a, b = min(res['0']), min(res['1'])
    print('dW[%4d,%4d] over %7d rows: 128-tiles %8.1f us %6.1f TF/s | 256-tiles %8.1f us %6.1f TF/s | x%.2f'
          % (N, K, rows, a, fl / a / 1e6, b, fl / b / 1e6, a / b), flush=True)
