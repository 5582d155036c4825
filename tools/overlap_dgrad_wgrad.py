#!/usr/bin/env python
"""Does running a layer's data gradient (kk kernel, heavy epilogue, HBM-write phases) on a second stream BESIDE its weight
gradient (rr kernel, MFMA-bound, almost no writes) beat running them back to back?  FFN shapes of the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
side = torch.cuda.Stream()
for M in (102400, 51200, 25600):
    d, dff = 512, 2048
    g2 = torch.randn(M, d, device=dev).bfloat16()          # d(out) of FFN2
    h = torch.randn(M, dff, device=dev).bfloat16()         # FFN2 input
    pre = torch.randn(M, dff, device=dev).bfloat16()
    xa = torch.randn(M, d, device=dev).bfloat16()
    w1 = torch.nn.Parameter(torch.randn(dff, d, device=dev) / 22)
    w2 = torch.nn.Parameter(torch.randn(d, dff, device=dev) / 45)
    wt2 = ops._weight_t_shadow(w2, True)
    dpre = torch.empty(M, dff, device=dev, dtype=torch.bfloat16)
    slabs = torch.zeros(((M + 127) // 128 * 4, dff), device=dev)

    def dgrad2():
        ops._gemm_raw_untimed(M, dff, d, g2, g2.stride(0), 1, wt2, 1, wt2.stride(0), dpre, dff, dact_src=pre, dact=2,
                              dropout_p=0.1, seed=1, offset=8, colsum_slabs=slabs)
    def wgrad2():
        return ops.linear_wgrad(g2, h)
    def dgrad1():
        return ops.linear_dgrad(dpre, w1)
    def wgrad1():
        return ops.linear_wgrad(dpre, xa)

    def seq():
        wgrad2(); dgrad2(); wgrad1(); dgrad1()

    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev); dgrad2(); e2 = torch.cuda.Event(); e2.record(side)
        wgrad2()
        torch.cuda.current_stream().wait_event(e2)
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev); dgrad1(); e3 = torch.cuda.Event(); e3.record(side)
        wgrad1()
        torch.cuda.current_stream().wait_event(e3)

    res = {}
    for rnd in range(3):
        for name, fn in (('seq', seq), ('par', par)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 200)
    print('FFN backward GEMMs, M = %6d: sequential %7.1f us, dgrad on a side stream %7.1f us  (x%.3f)' % (
        M, min(res['seq']), min(res['par']), min(res['seq']) / min(res['par'])), flush=True)
