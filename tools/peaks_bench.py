#!/usr/bin/env python
"""Measured peak denominators on this node (SURVEY 8d asks for them beside the vendor peaks):
streaming copy bandwidth and a library bf16 GEMM (hipBLASLt through torch.matmul), plus this
repo's own GEMM kernel at the same square sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
dev = torch.device('cuda:0')
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
x = torch.empty(1 << 30, device=dev, dtype=torch.uint8); y = torch.empty_like(x)
t = timed(lambda: y.copy_(x), 10)
print('streaming copy 1 GiB: %.2f TB/s (read + write)' % (2 * x.numel() / t / 1e12))
ops.set_compute_mode('bf16')
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev).bfloat16(); b = torch.randn(n, n, device=dev).bfloat16()
    t = timed(lambda: torch.matmul(a, b.t()), 10)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    t2 = timed(lambda: ops._gemm_raw_untimed(n, n, n, a, n, 1, b, 1, n, c, n), 10)
    print('bf16 GEMM %d^3: library (torch.matmul) %.0f TFLOP/s | nsp_gemm %.0f TFLOP/s' % (n, 2 * n ** 3 / t / 1e12, 2 * n ** 3 / t2 / 1e12))
