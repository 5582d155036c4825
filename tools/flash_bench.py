#!/usr/bin/env python
"""Kernel-only timing of the fused attention kernels at the step's shapes (H = 8, d_k = 64; T = 800 / 400 / 200).

usage: python tools/flash_bench.py [B] [reps] [T,T,...]        (NSP_LIB_OVERRIDE=<variant .so> selects another build of the library)
Prints forward and backward (dQ kernel + dK/dV kernel) microseconds per call with and without dropout; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, dk, clamp = 8, 64, 10
d = H * dk
R, Rp = clamp + 1, 16
print('lib: %s' % os.environ.get('NSP_LIB_OVERRIDE', 'tree'), flush=True)
Ts = tuple(int(t) for t in sys.argv[3].split(',')) if len(sys.argv) > 3 else (800, 400, 200)
for T in Ts:
    for p_drop in (0.0, 0.1):
        torch.manual_seed(T)
        qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).bfloat16()
        QP = torch.randn(B, T, H, Rp, device=dev)
        klens = torch.randint(int(T * 0.75), T + 1, (B,), device=dev, dtype=torch.int32)
        klens[0] = T
        mp = ops._mask_params(B, H, T, T, R, clamp, 1.0 / math.sqrt(dk), klens, False, 0, 0, 0, dropout_p=p_drop,
                              seed=3, offset=0, r_pitch=Rp)
        dO = torch.randn(B * T, d, device=dev).bfloat16()
        dqkv = torch.empty(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
        fwd_out = ops.flash_attn_fwd_raw(qkv, d, QP, mp)
        def fwd(): ops.flash_attn_fwd_raw(qkv, d, QP, mp)
        def bwd(): ops.flash_attn_bwd_raw(qkv, d, QP, dO, fwd_out[1], fwd_out[2], fwd_out[3], mp, dqkv)
        res = []
        for fn in (fwd, bwd):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1000.0 / reps)
        fl = 4.0 * B * H * T * T * dk
        print('B %3d T %4d dropout %.1f: fwd %7.1f us (%6.1f TFLOP/s)  bwd %7.1f us (%6.1f TFLOP/s, 2.5x fwd flops)' % (
            B, T, p_drop, res[0], fl / res[0] / 1e6, res[1], 2.5 * fl / res[1] / 1e6), flush=True)
