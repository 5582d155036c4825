#!/usr/bin/env python
"""Do the LSTM step-kernel chain (side stream) and big GEMMs (main stream) overlap on this GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
B, L, H = 16, 200, 1024
x = torch.randn(B, L, 512, device=dev)
w_ih = torch.nn.Parameter(torch.randn(4 * H, 512, device=dev) * 0.02)
w_hh = torch.nn.Parameter(torch.randn(4 * H, H, device=dev) * 0.02)
b = torch.nn.Parameter(torch.zeros(4 * H, device=dev))
a = torch.randn(3000, 512, device=dev)
w1 = torch.nn.Parameter(torch.randn(2048, 512, device=dev) * 0.02)
b1 = torch.nn.Parameter(torch.zeros(2048, device=dev))

def lstm_chain():
    with torch.no_grad():
        return ops.lstm(x, w_ih, w_hh, b, b)

def gemms(n):
    with torch.no_grad():
        for _ in range(n):
            ops.linear(a, w1, b1)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

dyv = torch.randn(B, L, H, device=dev)
def lstm_fb():
    y = ops.lstm(x, w_ih, w_hh, b, b)
    y.backward(dyv)
print('lstm fwd chain alone %.3f ms, fwd+bwd (incl. wgrad GEMMs) %.3f ms' % (timed(lstm_chain), timed(lstm_fb)))
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for prio in (0, -1):
    side = torch.cuda.Stream(device=dev, priority=prio)
    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            lstm_chain()
        gemms(NG)
        torch.cuda.current_stream().wait_stream(side)
    def both_rev():
        gemms(NG)
        side.wait_stream(torch.cuda.current_stream()) if False else None
        with torch.cuda.stream(side):
            lstm_chain()
        torch.cuda.current_stream().wait_stream(side)
    print('prio %d: lstm alone %.2f ms | %d gemms alone %.2f ms | both (lstm enqueued first) %.2f ms | both (gemms first) %.2f ms'
          % (prio, timed(lstm_chain), NG, timed(lambda: gemms(NG)), timed(both), timed(both_rev)))
