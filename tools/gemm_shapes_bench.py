#!/usr/bin/env python
"""Kernel-only timing + correctness of the bf16 KCxKC GEMM (y = x W^T) at the step's shapes (batch 64).
usage: python tools/gemm_shapes_bench.py [quick]      (arms: NSP_GEMM_8P = 0 -> 128 x 128 kernels, 1 -> the launcher's rule)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
shapes = [(51200, 2048, 512), (51200, 512, 2048), (51200, 1536, 512), (51200, 512, 512), (51200, 1024, 512),
          (25600, 2048, 512), (25600, 512, 2048), (25600, 1536, 512), (12800, 2048, 512), (12800, 512, 2048),
          (12800, 1536, 512), (51201, 1000, 512), (4096, 4096, 4096), (8192, 8192, 8192),
          (1843200, 1024, 512), (1843200, 512, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    shapes = shapes[:6]
for M, N, K in shapes:
    big = M > 1000000
    a = (torch.randn(M, K, device='cuda') * 0.5).bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.5).bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if big else torch.float32)
    bias = torch.randn(N, device='cuda')
    def run():
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N, bias=bias)
    out = []
    for arm in ('0', '1'):
        os.environ['NSP_GEMM_8P'] = arm
        for _ in range(3): run()
        torch.cuda.synchronize()
        iters = 5 if big else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        rows = slice(0, M) if M <= 60000 else slice(M - 70000, M)
        ref = a[rows].float() @ w.float().t() + bias
        err = ((c[rows].float() - ref).abs().max() / ref.abs().max()).item()
        out.append('%9.1f us %7.1f TFLOP/s err %.0e' % (us, 2.0 * M * N * K / us / 1e6, err))
    print('M %8d N %5d K %5d | 128 x 128 %s | launcher's rule %s' % (M, N, K, out[0], out[1]), flush=True)
