#!/usr/bin/env python
"""Microbenchmark: cost of epilogue options on the FFN1 shape (M=12800,N=2048,K=512), bf16 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
def bench(name, M, N, K, **kw):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / 20).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=kw.pop('odt', torch.float32))
    args = dict(kw)
    if args.pop('bias', False): args['bias'] = torch.randn(N, device=dev)
    if args.pop('pre', False): args['pre_out'] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if args.pop('res', False): args['res'] = torch.randn(M, N, device=dev)
    for _ in range(3): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-44s %8.1f us  %7.1f TF' % (name, us, 2.0 * M * N * K / us / 1e6))
M, N, K = 12800, 2048, 512
bench('plain fp32 out', M, N, K)
bench('plain bf16 out', M, N, K, odt=torch.bfloat16)
bench('bias', M, N, K, bias=True, odt=torch.bfloat16)
bench('bias+swish', M, N, K, bias=True, act=2, odt=torch.bfloat16)
bench('bias+swish+pre', M, N, K, bias=True, act=2, pre=True, odt=torch.bfloat16)
bench('bias+swish+pre+dropout', M, N, K, bias=True, act=2, pre=True, dropout_p=0.1, seed=1, offset=8, odt=torch.bfloat16)
bench('bias+dropout', M, N, K, bias=True, dropout_p=0.1, seed=1, offset=8, odt=torch.bfloat16)
bench('FFN2: bias+dropout+res fp32 (N=512,K=2048)', M, 512, 2048, bias=True, dropout_p=0.1, seed=1, offset=8, res=True)
bench('FFN2 plain', M, 512, 2048)
bench('square 4096', 4096, 4096, 4096)
bench('square 8192', 8192, 8192, 8192, odt=torch.bfloat16)
