#!/usr/bin/env python
"""Round 4: the phase-interleaved 256 x 256 GEMM kernel (gemm_bf16_kk8p_kernel) against the 128 x 128 kernels it replaces,
in ONE process, arms interleaved (NSP_GEMM_8P / NSP_GEMM_8P_VAR are read on every call): the training step's ten GEMM
configurations at M = 25600 and 102400 rows, the 8192^3 / 4096^3 squares beside the node's library GEMM, and a race screen
(the same product repeated, every result compared with the first)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops

ops.set_compute_mode('bf16')
dev = torch.device('cuda:0')
ARMS = [('128x128', {'NSP_GEMM_8P': '0'}), ('8p (rule)', {'NSP_GEMM_8P': '1'}), ('8p forced', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '4'})]
if os.environ.get('NSP_8P_AB_BUILD'):      # a -DNSP_GEMM_8P_AB=1 build: the direct-epilogue twins
    ARMS += [('8p-direct', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '0'}), ('8p-direct-l2', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '8'})]
if os.environ.get('ARMS'):
    ARMS = [a for a in ARMS if a[0] in os.environ['ARMS'].split(',')]


def setarm(env):
    for k in ('NSP_GEMM_8P', 'NSP_GEMM_8P_VAR'):
        os.environ.pop(k, None)
    os.environ.update(env)


def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def configs(M):
    bf = torch.bfloat16
    return [
        ('FFN1 fwd  [M,2048,512] bias swish pre16 drop -> bf16', M, 2048, 512, bf, dict(bias=True, act=2, pre=True, dropout_p=0.1, seed=1, offset=8)),
        ('FFN2 fwd  [M,512,2048] bias drop res -> fp32', M, 512, 2048, torch.float32, dict(bias=True, dropout_p=0.1, seed=1, offset=8, res=True, alpha=0.5)),
        ('QKV       [M,1536,512] plain -> bf16', M, 1536, 512, bf, {}),
        ('att out   [M,512,512] drop res -> fp32', M, 512, 512, torch.float32, dict(dropout_p=0.1, seed=1, offset=8, res=True)),
        ('pw1       [M,1024,512] bias -> fp32', M, 1024, 512, torch.float32, dict(bias=True)),
        ('pw2       [M,512,512] bias drop res -> fp32', M, 512, 512, torch.float32, dict(bias=True, dropout_p=0.1, seed=1, offset=8, res=True)),
        ("dgrad FFN2 [M,2048,512] swish' src16 drop slabs -> bf16", M, 2048, 512, bf, dict(dsrc=True, dact=2, dropout_p=0.1, seed=1, offset=8, slabs=True)),
        ('dgrad FFN1 [M,512,2048] plain -> fp32', M, 512, 2048, torch.float32, {}),
        ('dgrad QKV [M,512,1536] plain -> fp32', M, 512, 1536, torch.float32, {}),
        ('dgrad d   [M,512,512] plain -> fp32', M, 512, 512, torch.float32, {}),
    ]


def run_shapes(M, rounds=3):
    print('\n=== step GEMMs at M = %d rows: min us over %d interleaved rounds (TFLOP/s) ===' % (M, rounds))
    print('%-60s ' % '' + ' '.join('%18s' % a for a, _ in ARMS))
    for name, M_, N, K, odt, kw in configs(M):
        x = torch.randn(M_, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        out = torch.empty(M_, N, device=dev, dtype=odt)
        a = dict(kw)
        if a.pop('bias', False): a['bias'] = torch.randn(N, device=dev)
        if a.pop('pre', False): a['pre_out'] = torch.empty(M_, N, device=dev, dtype=torch.bfloat16)
        if a.pop('res', False): a['res'] = torch.randn(M_, N, device=dev)
        if a.pop('dsrc', False): a['dact_src'] = torch.randn(M_, N, device=dev).bfloat16()
        if a.pop('slabs', False): a['colsum_slabs'] = torch.zeros(((M_ + 127) // 128 * 4, N), device=dev)
        fn = lambda: ops._gemm_raw_untimed(M_, N, K, x, K, 1, w, 1, K, out, N, **a)
        best = {n: 1e30 for n, _ in ARMS}
        for r in range(rounds):
            for n, env in ARMS:
                setarm(env)
                fn(); fn()
                best[n] = min(best[n], timeit(fn))
        fl = 2.0 * M_ * N * K
        print('%-60s ' % name + ' '.join('%9.1f (%6.0f)' % (best[n], fl / best[n] / 1e6) for n, _ in ARMS))
        del x, w, out, a


def squares():
    print('\n=== squares: library (torch.matmul) vs arms, TFLOP/s ===')
    for n in (4096, 8192):
        a = torch.randn(n, n, device=dev).bfloat16(); b = torch.randn(n, n, device=dev).bfloat16()
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        lib = min(timeit(lambda: torch.matmul(a, b.t())) for _ in range(3))
        row = ['library %.0f' % (2 * n ** 3 / lib / 1e6)]
        for nm, env in ARMS:
            setarm(env)
            f = lambda: ops._gemm_raw_untimed(n, n, n, a, n, 1, b, 1, n, c, n)
            f(); f()
            t = min(timeit(f) for _ in range(3))
            row.append('%s %.0f' % (nm, 2 * n ** 3 / t / 1e6))
        print('%d^3: ' % n + ' | '.join(row))


def ablate():
    print('\n=== main-loop ablations (plain epilogues; results of the ablated arms are wrong): us (TFLOP/s) ===')
    arms = [('128x128', {'NSP_GEMM_8P': '0'}), ('8p', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '4'}),
            ('8p no waits', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '20'}), ('8p no loads', {'NSP_GEMM_8P': '2', 'NSP_GEMM_8P_VAR': '36'})]
    print('%-36s ' % '' + ' '.join('%18s' % a for a, _ in arms))
    for (M, N, K, odt) in ((8192, 8192, 8192, torch.bfloat16), (4096, 4096, 4096, torch.bfloat16), (102400, 1536, 512, torch.bfloat16),
                           (102400, 512, 2048, torch.float32), (102400, 512, 512, torch.float32), (16384, 2048, 512, torch.bfloat16)):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=odt)
        fn = lambda: ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N)
        best = {n: 1e30 for n, _ in arms}
        for r in range(3):
            for n, env in arms:
                setarm(env); fn(); fn()
                best[n] = min(best[n], timeit(fn))
        fl = 2.0 * M * N * K
        print('%-36s ' % ('%d x %d x %d -> %s' % (M, N, K, 'bf16' if odt == torch.bfloat16 else 'fp32')) + ' '.join('%9.1f (%6.0f)' % (best[n], fl / best[n] / 1e6) for n, _ in arms))


def race_screen(reps=200):
    """the same product again and again beside a result computed once with the 128 x 128 kernel: a too-early LDS read or a
    buffer re-armed too early shows as rare wrong tiles (they come and go with memory load -- hence the repetitions and
    the two shapes: many tiles per workgroup / long reductions)"""
    print('\n=== race screen ===')
    for (M, N, K) in ((25600, 2048, 512), (8192, 8192, 2048), (70000, 512, 384), (3000, 1000, 128)):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        ref = torch.empty(M, N, device=dev); out = torch.empty(M, N, device=dev)
        setarm({'NSP_GEMM_8P': '0'})
        ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, ref, N)
        bad = 0
        for nm, env in ARMS[1:]:
            setarm(dict(env, NSP_GEMM_8P_MIN_TILES='1', NSP_GEMM_8P='2'))
            for r in range(reps):
                out.fill_(float('nan'))
                ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N)
                d = (out - ref).abs().max().item()
                if not (d <= 1e-3 * ref.abs().max().item()):
                    bad += 1
                    if bad < 5:
                        wrong = ((out - ref).abs() > 1e-3 * ref.abs().max()) | torch.isnan(out)
                        rows = wrong.any(1).nonzero().flatten()
                        print('  MISMATCH %s rep %d: max diff %g, %d wrong elements, rows %d..%d' % (nm, r, d, int(wrong.sum()), int(rows.min()), int(rows.max())))
        os.environ.pop('NSP_GEMM_8P_MIN_TILES', None)
        print('  %d x %d x %d: %d repetitions x %d arms, %d mismatches' % (M, N, K, reps, len(ARMS) - 1, bad))


if __name__ == '__main__':
    what = sys.argv[1:] or ['race', 'shapes', 'squares']
    if 'race' in what: race_screen(int(os.environ.get('RACE_REPS', '100')))
    if 'squares' in what: squares()
    if 'ablate' in what: ablate()
    if 'shapes' in what:
        for M in (25600, 51200, 102400): run_shapes(M)
