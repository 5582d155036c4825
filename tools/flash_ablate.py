#!/usr/bin/env python
"""Ablation of the fused-attention forward (temporary template<ABL> build): which part of the k-loop costs what.
bits: 1 no cross-lane shuffles, 2 no exp2, 4 no lo-half P.V MFMAs, 8 no P.V MFMAs, 16 no Q.K MFMAs, 32 no K/V
prefetch + LDS restage, 64 no barrier."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
dev = torch.device('cuda:0')
B, H, dk, clamp, T = 64, 8, 64, 10, 800
d = H * dk
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).bfloat16()
QP = torch.randn(B, T, H, 16, device=dev)
klens = torch.full((B,), T, device=dev, dtype=torch.int32)
for p_drop in (0.0, 0.1):
    mp = ops._mask_params(B, H, T, T, clamp + 1, clamp, 1.0 / math.sqrt(dk), klens, False, 0, 0, 0, dropout_p=p_drop, seed=3, offset=0, r_pitch=16)
    for abl in (0, 1, 2, 3, 4, 8, 16, 24, 27, 32, 59, 64, 96, 123):
        os.environ['NSP_FLASH_ABL'] = str(abl)
        for _ in range(2): ops.flash_attn_fwd_raw(qkv, d, QP, mp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.flash_attn_fwd_raw(qkv, d, QP, mp)
        e1.record(); torch.cuda.synchronize()
        print('dropout %.1f ABL %3d: %7.1f us' % (p_drop, abl, e0.elapsed_time(e1) * 100), flush=True)
