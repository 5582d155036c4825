#!/usr/bin/env python
"""Time nsp_conv2d3x3_wgrad for the 1->32 front-end conv at bench size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import _lib, ops
B, T, F = 16, 1600, 80
x = torch.randn(B, T, F, 1, device='cuda')
dz = torch.randn(B, T, F, 32, device='cuda')
buf = torch.zeros(32 * 9 + 32, device='cuda')
def run():
    ops._check(_lib.lib().nsp_conv2d3x3_wgrad(x.data_ptr(), dz.data_ptr(), buf.data_ptr(), buf.data_ptr() + 4 * 288,
                                             B, T, F, 1, 32, 0, 0, ops._stream()), 'wgrad')
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print('c1 wgrad: %.1f us (%.2f TB/s on dy)' % (e0.elapsed_time(e1) * 50, dz.numel() * 4 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12))
