#!/usr/bin/env python
"""Kernel-only timing of the LayerNorm kernels at the step's shapes (rows = 102400 / 51200 / 25600, d = 512).
    [NSP_LIB_OVERRIDE=<variant>] python tools/ln_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
dev = torch.device('cuda:0')
ops.set_compute_mode('bf16')
print('lib: %s' % os.environ.get('NSP_LIB_OVERRIDE', 'tree'), flush=True)
d = 512
g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows in (102400, 51200, 25600):
    x = torch.randn(rows, d, device=dev)
    dy = torch.randn(rows, d, device=dev)
    dres = torch.randn(rows, d, device=dev)
    y, mean, rstd, _, y16 = ops.layernorm_fwd_raw(x, g, b, 1e-12, 0, want16=True, want32=False)
    t_f = tm(lambda: ops.layernorm_fwd_raw(x, g, b, 1e-12, 0, want16=True, want32=False))
    t_b = tm(lambda: ops.layernorm_bwd_raw(dy, x, g, mean, rstd, None, 0, dres=dres, prep=(0, 0.5, 0.1, 1, 2, d)))
    ops._PREP.clear()
    nf, nb = rows * d * 6, rows * d * 18
    print('rows %6d: fwd (bf16 image only) %6.1f us = %.2f TB/s | bwd (+ residual gradient, + prepared image) %6.1f us = %.2f TB/s'
          % (rows, t_f, nf / t_f / 1e6, t_b, nb / t_b / 1e6), flush=True)
