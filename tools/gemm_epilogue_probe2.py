#!/usr/bin/env python
"""Cost of each epilogue ingredient on the K = 512 FFN GEMM (51200 x 2048 x 512), back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('PROBE_SHAPES', '51200x2048x512').split(',')]
for M, N, K in SHAPES:
    a = (torch.randn(M, K, device='cuda') * 0.5).bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.5).bfloat16()
    c32 = torch.empty(M, N, device='cuda', dtype=torch.float32)
    c16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    p16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    d16 = torch.randn(M, N, device='cuda').bfloat16()
    res = torch.randn(M, N, device='cuda')
    bias = torch.randn(N, device='cuda')
    slabs = torch.zeros(((M + 127) // 128 * 4, N), device='cuda')
    g = lambda C, **k: (lambda: ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, C, N, **k))
    d32 = d16.float()
    cases = [
        ('o16 +dact16(relu)', g(c16, dact_src=d16, dact=1)),
        ('o16 +dact32(relu)', g(c16, dact_src=d32, dact=1)),
        ('o16 +dact32(swish)', g(c16, dact_src=d32, dact=2)),
        ('o16 +res', g(c16, res=res)),
        ('o16 +swish', g(c16, act=2)),
        ('o32', g(c32)),
        ('o16', g(c16)),
        ('o16 +bias', g(c16, bias=bias)),
        ('o16 +bias +pre16', g(c16, bias=bias, pre_out=p16)),
        ('o16 +bias +pre16 +swish', g(c16, bias=bias, pre_out=p16, act=2)),
        ('o16 +bias +pre16 +swish +dropout', g(c16, bias=bias, pre_out=p16, act=2, dropout_p=0.1, seed=5)),
        ('o16 +dact16(swish)', g(c16, dact_src=d16, dact=2)),
        ('o16 +dact16 +dropout', g(c16, dact_src=d16, dact=2, dropout_p=0.1, seed=5)),
        ('o16 +dact16 +dropout +slabs', g(c16, dact_src=d16, dact=2, dropout_p=0.1, seed=5, colsum_slabs=slabs)),
        ('o32 +res', g(c32, res=res)),
        ('o32 +bias +res +dropout', g(c32, bias=bias, res=res, alpha=0.5, dropout_p=0.1, seed=5)),
    ]
    for name, fn in cases:
        us = bench(fn)
        print('M %6d N %5d K %5d  %-36s %8.1f us %6.1f TFLOP/s' % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6), flush=True)
