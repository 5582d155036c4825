#!/usr/bin/env python
"""Two GEMM shapes of the training step, 12 launches each, for rocprofv3 --pmc passes (tools/gemm_pmc.sh):
FFN1 forward [M, 2048] x K = 512 (two bf16 outputs: the write-bound kind) and FFN2 forward [M, 512] x K = 2048."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
M = int(os.environ.get('GM', '25600'))


def run(M, N, K, odt, **kw):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=odt)
    a = dict(kw)
    if a.pop('bias', False): a['bias'] = torch.randn(N, device=dev)
    if a.pop('pre', False): a['pre_out'] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if a.pop('res', False): a['res'] = torch.randn(M, N, device=dev)
    for _ in range(12): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **a)
    torch.cuda.synchronize()


run(M, 2048, 512, torch.bfloat16, bias=True, act=2, pre=True, dropout_p=0.1, seed=1, offset=8)
run(M, 512, 2048, torch.float32, bias=True, dropout_p=0.1, seed=1, offset=8, res=True, alpha=0.5)
