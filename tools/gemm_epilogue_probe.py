#!/usr/bin/env python
"""Where does a short-K GEMM spend its time?  Same (M,N,K), different epilogues / output types."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
for M, N, K in [(51200, 2048, 512), (51200, 512, 2048), (51200, 512, 512), (51200, 2048, 2048), (51200, 2048, 128)]:
    a = (torch.randn(M, K, device='cuda') * 0.5).bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.5).bfloat16()
    c32 = torch.empty(M, N, device='cuda', dtype=torch.float32)
    c16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    p16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    res = torch.randn(M, N, device='cuda')
    bias = torch.randn(N, device='cuda')
    cases = {
        'fp32 out': lambda: ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c32, N),
        'bf16 out': lambda: ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c16, N),
        'bf16 out + bias + swish + pre(bf16)': lambda: ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c16, N, bias=bias, act=2, pre_out=p16),
        'fp32 out + bias + res + dropout': lambda: ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c32, N, bias=bias, res=res, alpha=0.5, dropout_p=0.1, seed=5, offset=0),
    }
    for name, fn in cases.items():
        r = []
        for persist in ('0', '1'):
            os.environ['NSP_GEMM_PERSIST'] = persist
            us = bench(fn)
            r.append('%8.1f us %6.1f TFLOP/s' % (us, 2.0 * M * N * K / us / 1e6))
        print('M %6d N %5d K %5d  %-38s classic %s | persistent %s' % (M, N, K, name, r[0], r[1]), flush=True)
