#!/usr/bin/env python
"""Kernel-only timing of the bf16 KCxKC GEMM (y = x W^T) at the encoder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
shapes = [(3200, 512, 2048), (3200, 2048, 512), (3200, 512, 512), (3200, 1536, 512), (6400, 512, 2048), (6400, 2048, 512),
          (6400, 512, 512), (12800, 512, 2048), (12800, 2048, 512), (12800, 512, 512), (3216, 4096, 512), (3216, 512, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.float32)
    def run():
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    ref = a.float() @ w.float().t()
    err = ((c - ref).abs().max() / ref.abs().max()).item()
    print('M %6d N %5d K %5d: %7.1f us %7.1f TFLOP/s  relerr %.1e' % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
