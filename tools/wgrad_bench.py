#!/usr/bin/env python
"""Weight-gradient GEMMs dW[N,K] = dY[M,N]^T X[M,K] (split-K slabs) at the encoder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
M = int(os.environ.get('WM', '51200'))
for N, K in [(2048, 512), (512, 2048), (512, 512), (1536, 512), (1024, 512)]:
    dy = torch.randn(M, N, device='cuda').bfloat16()
    x = torch.randn(M, K, device='cuda').bfloat16()
    def run():
        return ops.linear_wgrad(dy, x)
    dw = run()
    ref = dy.float().t() @ x.float()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print('dW[%4d,%4d] over M=%d: %7.1f us (incl. slab reduce) %6.1f TFLOP/s relerr %.1e' % (N, K, M, us, 2.0 * M * N * K / us / 1e6, err))
