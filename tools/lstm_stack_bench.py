#!/usr/bin/env python
"""Time the 2x1024 prediction-network LSTM stack (fwd, fwd+bwd): persistent vs per-stage launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
B, L, I, H, nl = int(os.environ.get("LB", "16")), int(os.environ.get("LL", "200")), 1024, 1024, 2
refs = [torch.nn.LSTM(I if l == 0 else H, H, 1, batch_first=True).to(dev) for l in range(nl)]
layers = [(r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0) for r in refs]
x = torch.randn(B, L, I, device=dev, requires_grad=True)
dy = torch.randn(B, L, H, device=dev)
def fwd():
    with torch.no_grad():
        return ops.lstm_stack(x, layers, 0.1)
def fb():
    y = ops.lstm_stack(x, layers, 0.1)
    y.backward(dy)
def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for flag in ('0', '1'):
    os.environ['NSP_LSTM_PERSISTENT'] = flag
    f, b = timed(fwd), timed(fb)
    y = fwd()
    print('persistent=%s: fwd %.3f ms, fwd+bwd %.3f ms (bwd incl. 6 GEMMs) finite=%s | per stage: fwd %.1f us, bwd %.1f us' % (flag, f, b, bool(torch.isfinite(y).all()), f * 1e3 / (L + 2), (b - f) * 1e3 / (L + 2)))
