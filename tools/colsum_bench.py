#!/usr/bin/env python
"""Column sums (bias gradients, slab rows) at the step's shapes: microseconds per call and agreement with torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
dev = torch.device('cuda:0')
print('lib: %s' % os.environ.get('NSP_LIB_OVERRIDE', 'tree'))
for rows, cols, dt in ((102400, 1024, torch.bfloat16), (25600, 1024, torch.bfloat16), (3200, 2048, torch.float32), (1600, 512, torch.float32), (102400, 512, torch.float32),
                       (1001, 520, torch.float32), (777, 24, torch.bfloat16)):
    x = torch.randn(rows, cols, device=dev).to(dt)
    def fn(): return ops.colsum(x)
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    ref = x.float().sum(0)
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print('%7d x %5d %-9s %7.1f us (incl. the zero fill)  rel err %.1e' % (rows, cols, str(dt).split('.')[-1], us, err))
    assert err < 1e-4
