#!/usr/bin/env python
"""Split-K slab reduction (nsp_splitk_reduce) at the step's weight-gradient shapes: microseconds and TB/s per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, _lib
dev = torch.device('cuda:0')
print('lib: %s' % os.environ.get('NSP_LIB_OVERRIDE', 'tree'))
for splits, n in ((62, 512 * 512), (58, 512 * 512), (29, 512 * 512), (16, 2048 * 512), (20, 1536 * 512), (32, 1024 * 512), (25, 512 * 1280)):
    part = torch.randn(splits, n, device=dev); out = torch.empty(n, device=dev)
    def fn(): ops._check(_lib.lib().nsp_splitk_reduce(ops._p(part), ops._p(out), splits, n, ops._stream()), 'r')
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    assert torch.allclose(out, part.sum(0), rtol=1e-4, atol=1e-4)
    print('%3d splits x %8d: %7.1f us  %5.2f TB/s' % (splits, n, us, (splits + 1) * n * 4 / us / 1e6))
