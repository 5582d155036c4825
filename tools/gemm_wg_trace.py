#!/usr/bin/env python
"""Where does a workgroup of gemm_bf16_kk_glds_kernel<0> spend its life?  Needs the development build
tools/probe/ab/libnsp_hip_gemmtrace.so (tools/make_variant_lib.sh gemmtrace gemm_bf16.hip -DNSP_GEMM_TRACE=1), which
records four cycle-counter stamps per workgroup: start of the k-loop, first tile landed, k-loop done, stores drained.
usage: NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_gemmtrace.so python tools/gemm_wg_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from neural_sp_amd import ops, _lib
ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
lib = _lib.lib()
lib.nsp_gemm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]


def run(name, M, N, K, odt, **kw):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=odt)
    a = dict(kw)
    if a.pop('bias', False): a['bias'] = torch.randn(N, device=dev)
    if a.pop('pre', False): a['pre_out'] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if a.pop('res', False): a['res'] = torch.randn(M, N, device=dev)
    for _ in range(3): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **a); e1.record()
    torch.cuda.synchronize()
    nwg = min(16384, ((M + 127) // 128) * ((N + 127) // 128))
    buf = np.zeros((nwg, 4), dtype=np.uint64)
    assert lib.nsp_gemm_trace_read(buf.ctypes.data, nwg) == 0
    t = buf.astype(np.float64)
    xcd = np.arange(nwg) % 8
    ok = t[:, 3] > t[:, 0]
    t, xcd = t[ok], xcd[ok]
    # Stamps of DIFFERENT workgroups cannot be subtracted (measured: the counters of different CUs / XCDs are millions
    # of ticks apart), so only per-workgroup differences are used, and the tick is calibrated on occupancy: the kernel
    # keeps 4 workgroups on each of 256 CUs for the whole launch, so mean lifetime = launch time x 1024 / workgroups
    # (an over-estimate by the emptier last round).
    spans = []
    us_per_tick = e0.elapsed_time(e1) * 1e3 * min(1.0, 1024.0 / len(t)) / float((t[:, 3] - t[:, 0]).mean())
    d = lambda a, b: (t[:, b] - t[:, a]) * us_per_tick
    nkt = K // 64
    print('%s: launch %.1f us (events), %d workgroups' % (name, e0.elapsed_time(e1) * 1e3, len(t)))
    for lab, v in (('first tile landed', d(0, 1)), ('remaining %d k-tiles' % (nkt - 1), d(1, 2)), ('epilogue + store drain', d(2, 3)), ('whole workgroup', d(0, 3))):
        print('   %-26s mean %7.2f us   p10 %7.2f   p50 %7.2f   p90 %7.2f' % (lab, v.mean(), *np.percentile(v, [10, 50, 90])))
    print('   per k-tile after the first: %.2f us' % (d(1, 2).mean() / max(nkt - 1, 1)))


M = int(os.environ.get('GM', '25600'))
run('FFN1 fwd [M,2048]xK512 bias swish drop, 2 bf16 outputs', M, 2048, 512, torch.bfloat16, bias=True, act=2, pre=True, dropout_p=0.1, seed=1, offset=8)
run('QKV [M,1536]xK512 plain -> bf16', M, 1536, 512, torch.bfloat16)
run('pw1 [M,1024]xK512 bias -> fp32', M, 1024, 512, torch.float32, bias=True)
run('FFN2 fwd [M,512]xK2048 bias drop res -> fp32', M, 512, 2048, torch.float32, bias=True, dropout_p=0.1, seed=1, offset=8, res=True, alpha=0.5)
