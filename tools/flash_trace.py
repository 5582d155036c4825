#!/usr/bin/env python
"""Where a wave of the dK/dV kernel spends its cycles: s_memtime stamps at the phase boundaries of every query-tile
iteration, summed over all waves (variant library built from a stamped copy of flash_attn.hip: see DESIGN section 13).
    NSP_LIB_OVERRIDE=tools/variants/libnsp_hip_trace.so python tools/flash_trace.py"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, _lib
dev = torch.device('cuda:0')
B, H, dk, clamp, T = 128, 8, 64, 10, 800
d = H * dk
R, Rp = clamp + 1, 16
L = ctypes.CDLL(os.environ['NSP_LIB_OVERRIDE'])
for p_drop in (0.0, 0.1):
    torch.manual_seed(T)
    qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).bfloat16()
    QP = torch.randn(B, T, H, Rp, device=dev)
    klens = torch.randint(int(T * 0.75), T + 1, (B,), device=dev, dtype=torch.int32)
    klens[0] = T
    mp = ops._mask_params(B, H, T, T, R, clamp, 1.0 / math.sqrt(dk), klens, False, 0, 0, 0, dropout_p=p_drop, seed=3, offset=0, r_pitch=Rp)
    dO = torch.randn(B * T, d, device=dev).bfloat16()
    dqkv = torch.empty(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    O, O32, LSE, keep = ops.flash_attn_fwd_raw(qkv, d, QP, mp)
    run = (lambda: ops.flash_attn_fwd_raw(qkv, d, QP, mp)) if os.environ.get('FA_TRACE_FWD') else (lambda: ops.flash_attn_bwd_raw(qkv, d, QP, dO, O32, LSE, keep, mp, dqkv))
    run()
    torch.cuda.synchronize()
    L.nsp_debug_flash_trace(None, 1)
    run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 6)()
    L.nsp_debug_flash_trace(out, 0)
    tot = float(sum(out))
    names = os.environ.get('FA_TRACE_NAMES', 'loads + DMA issue|S, dP (LDS reads + MFMA)|soft-max, dS VALU|second products (tr reads + MFMA)|statistics store (dkv) or vmcnt(0) (dq)|barrier').split('|')
    print('dropout %.1f: total %.3g ticks over all waves' % (p_drop, tot))
    for n, v in zip(names, out):
        print('   %-34s %5.1f %%' % (n, 100.0 * v / tot))
